"""Shared by the counter summaries (pmc_summary.py, fp64_summary.py, sq_summary.py, collect_lanes.sh): reading rocprofv3's
counter_collection CSVs and picking the launches of the kernel that evaluates the batch."""
import collections
import csv
import glob
import os
import sys

ROUND = "r05"
# the kernels that evaluate a batch (one of them per plan: rp_last_kernel) and the ones around it
MAIN_KERNELS = ("rp_eval_kernel", "rp_cost_kernel", "rp_chunk_kernel")
SIDE_KERNELS = ("rp_lon_kernel", "rp_select_kernel", "rp_finalize_kernel")


def source_hash():
    """hash of the sources of the library these counters were measured on (rp_source_hash)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [os.path.join(root, "commonroad-reactive-planner_amd")]
    from commonroad_rp_amd import _capi
    return _capi.source_hash()


def rows_of(pattern):
    out = []
    for f in glob.glob(pattern):
        out += list(csv.DictReader(open(f)))
    return out


def short(kernel_name):
    """template instance without the argument list: 'void rp_eval_kernel<16, true, ...>'"""
    return kernel_name.split("(")[0]


def median(v):
    v = sorted(v)
    return v[len(v) // 2]


def by_kernel(rows, names=MAIN_KERNELS, largest_grid=True):
    """{kernel instance: {counter: [values]}} over the launches of ``names``; with ``largest_grid`` only the launches of the largest
    grid among them (the batch's evaluation launch -- the winner's re-evaluation and the cost-ordered rounds are smaller)."""
    rows = [r for r in rows if any(n in r["Kernel_Name"] for n in names)]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    if not rows:
        return vals
    gmax = max(int(r["Grid_Size"]) for r in rows)
    for r in rows:
        if not largest_grid or int(r["Grid_Size"]) == gmax:
            vals[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return vals


def main_kernel_of(bench_json):
    """the kernel family the bench line says evaluated the batch (roofline.kernel <- rp_last_kernel); both if the line has none"""
    import json
    try:
        k = json.load(open(bench_json))["roofline"]["kernel"]
        return (k,) if k in MAIN_KERNELS else MAIN_KERNELS
    except Exception:
        return MAIN_KERNELS
