cd /tmp && export TMPDIR=/tmp
for v in "cfg3" "cfg3 noobs" "cfg4" "cfg4 noobs"; do
  rm -rf /tmp/lt; rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 $GRAFT_REPO_ROOT/profiles/probe_lon_kernel.py $v > /dev/null 2>&1
  echo "== $v"; python3 $GRAFT_REPO_ROOT/profiles/summarize_trace.py $(ls /tmp/lt/*/*kernel_trace.csv | head -1) | grep -E "rp_lon|rp_eval_kernel<16, false" | cut -c1-125
done
