"""The launch paths of the library that the GPU tests run on (the environment switches are read per plan call):
  single_launch  small batches: one kernel computes the longitudinal profiles in LDS and evaluates (default)
  two_kernel     rp_lon_kernel + rp_eval_kernel (what large batches take), 16 lanes per candidate
  g32 / g64      two-kernel path with 32 / 64 lanes per candidate (g64: LDS-staged linear copy-out of state rows)
  lazy           two-kernel path with the cost-ordered collision stage forced for every production-mode plan with obstacles
                 (by default only batches beyond 16 384 candidates take it): costs first, collision rounds over the cheapest
  lane_cand      two-kernel path; costs-only launches (production-mode plans without a collision query, the first pass of the
                 cost-ordered stage) through rp_cost_kernel: one lane per candidate, the lane walks the time steps (what such plans
                 take from 131 072 candidates on)
  wave_wg        two-kernel path, 16 lanes per candidate, ONE wavefront per workgroup of the evaluation kernel (what costs-only plans
                 of batches beyond 131 072 candidates take: rp_host.hip eval_block)"""
import contextlib
import os

LAUNCH_PATHS = {
    "single_launch": {},
    "two_kernel": {"RP_AMD_NO_FUSED_LON": "1"},
    "g32": {"RP_AMD_NO_FUSED_LON": "1", "RP_AMD_G": "32"},
    "g64": {"RP_AMD_NO_FUSED_LON": "1", "RP_AMD_G": "64"},
    "lazy": {"RP_AMD_NO_FUSED_LON": "1", "RP_AMD_LAZY": "1", "RP_AMD_NO_AUTO_MATERIALIZE": "1"},
    "wave_wg": {"RP_AMD_NO_FUSED_LON": "1", "RP_AMD_G": "16", "RP_AMD_EVAL_BLOCK": "64"},
    "lane_cand": {"RP_AMD_NO_FUSED_LON": "1", "RP_AMD_COST_KERNEL": "1", "RP_AMD_NO_AUTO_MATERIALIZE": "1"},
}


@contextlib.contextmanager
def launch_path_env(name):
    saved = {k: os.environ.get(k) for k in ("RP_AMD_NO_FUSED_LON", "RP_AMD_G", "RP_AMD_LAZY", "RP_AMD_NO_AUTO_MATERIALIZE", "RP_AMD_EVAL_BLOCK", "RP_AMD_COST_KERNEL")}
    for k in saved:
        os.environ.pop(k, None)
    os.environ.update(LAUNCH_PATHS[name])
    try:
        yield
    finally:
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
