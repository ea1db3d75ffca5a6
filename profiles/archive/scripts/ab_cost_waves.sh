cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in librp_amd.so librp_amd_w4.so; do echo "== $lib"; RP_AMD_LIBRARY=$GRAFT_REPO_ROOT/commonroad-reactive-planner_amd/lib/$lib python profiles/probe_cost_kernel.py cfg4 cfg5 2>&1 | grep "no collision"; done; done
