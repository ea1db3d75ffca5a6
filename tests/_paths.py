"""The launch paths of the library that the GPU tests run on (options of the contexts: rp_set_option):
  single_launch  small batches: one kernel computes the longitudinal profiles in LDS and evaluates (default)
  two_kernel     rp_lon_kernel + rp_eval_kernel (what large batches take), 16 lanes per candidate
  g32 / g64      two-kernel path with 32 / 64 lanes per candidate (g64: LDS-staged linear copy-out of state rows)
  lazy           two-kernel path with the cost-ordered collision stage forced for every production-mode plan with obstacles
                 (by default only batches beyond 16 384 candidates take it): costs first, collision rounds over the cheapest
  lane_cand      two-kernel path; costs-only launches (production-mode plans without a collision query, the first pass of the
                 cost-ordered stage) through rp_cost_kernel: one lane per candidate, the lane walks the time steps (what such plans
                 take from 131 072 candidates on)
  wave_wg        two-kernel path, 16 lanes per candidate, ONE wavefront per workgroup of the evaluation kernel (what costs-only plans
                 of batches beyond 131 072 candidates take: rp_host.hip eval_block)
  lane_chunk     two-kernel path; costs-only launches through rp_chunk_kernel: one lane per candidate and step block of 16 steps
                 (what such plans take from ~30 000 to ~200 000 candidates at horizons of 17 .. 64 steps)
  sweep          two-kernel path with the cost-ordered stage forced AND run as a bounded sweep: costs of every candidate, then the eager
                 kernel over the batch for the candidates no cheaper free one rules out (rp_last_path() == 3)"""
import contextlib

from commonroad_rp_amd import _capi

# option sets of rp_set_option (include/rp_amd.h), applied to every context of the process for the duration of a test
LAUNCH_PATHS = {
    "single_launch": {},
    "two_kernel": {"fused_lon": 0},
    "g32": {"fused_lon": 0, "lanes": 32},
    "g64": {"fused_lon": 0, "lanes": 64},
    "lazy": {"fused_lon": 0, "lazy": 1, "auto_materialize": 0},
    "wave_wg": {"fused_lon": 0, "lanes": 16, "eval_block": 64, "chunk_kernel": 0},
    "lane_cand": {"fused_lon": 0, "cost_kernel": 1, "chunk_kernel": 0, "auto_materialize": 0},
    "lane_chunk": {"fused_lon": 0, "chunk_kernel": 1, "auto_materialize": 0},
    "sweep": {"fused_lon": 0, "lazy": 1, "sweep": 1, "auto_materialize": 0},
}


@contextlib.contextmanager
def launch_path_env(name):
    """(the name is from the rounds when these were environment variables read per plan: they are options of the contexts now)"""
    _capi.set_default_options(LAUNCH_PATHS[name])
    try:
        yield
    finally:
        _capi.set_default_options(None)


@contextlib.contextmanager
def options(**opts):
    _capi.set_default_options(opts)
    try:
        yield
    finally:
        _capi.set_default_options(None)
