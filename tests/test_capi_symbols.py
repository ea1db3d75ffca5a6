"""The C-ABI library loads and exports every symbol that include/rp_amd.h declares; the ctypes
structs have the layout of the C structs.  No compute calls here (no GPU in the build container)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from commonroad_rp_amd import _capi

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(REPO, "include", "rp_amd.h")


def _header_source():
    src = open(HEADER).read()
    return re.sub(r"/\*.*?\*/", "", src, flags=re.S)


def _inline_wrappers():
    """the ABI-version-1 plan entries kept as ``static inline`` wrappers (source compatibility; nothing exported)"""
    return sorted(set(re.findall(r"static\s+inline\s+int\s+(rp_[a-z_]+)\s*\(", _header_source())))


def _declared_functions():
    src = _header_source()
    src = re.sub(r"static\s+inline\s+[^{;]*\{.*?\n\}", "", src, flags=re.S)   # (bodies of the wrappers call the exported entries)
    return sorted(set(re.findall(r"\b(rp_[a-z_]+)\s*\(", src)))


def test_header_functions_match_binding_table():
    assert _declared_functions() == sorted(_capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    if not os.path.exists(_capi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(_capi.LIB_PATH)
    for name in _declared_functions():
        assert hasattr(lib, name), name
    lib.rp_abi_version.restype = C.c_int
    assert lib.rp_abi_version() == _capi.ABI_VERSION == 2
    for name in _inline_wrappers():   # folded into rp_plan / rp_plan_levels / rp_plan_coeffs: wrappers in the header, not symbols
        assert not hasattr(lib, name), name
    assert _inline_wrappers() == ["rp_plan_begin", "rp_plan_coeffs_grouped", "rp_plan_levels_begin", "rp_plan_levels_packed", "rp_plan_packed"]


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof from a C translation unit compiled against the header."""
    prog = r'''
#include "rp_amd.h"   /* first: the header must be self-contained */
#include <stdio.h>
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(rp_params), sizeof(rp_cost), sizeof(rp_grids), sizeof(rp_result));
  printf("%zu %zu %zu %zu\n", offsetof(rp_params, x0_lon), offsetof(rp_params, v_delta_max),
         offsetof(rp_result, reason_counts), offsetof(rp_result, kernel_ms));
  printf("%zu %zu %zu %zu %zu\n", offsetof(rp_params, struct_size), offsetof(rp_cost, struct_size), offsetof(rp_grids, struct_size),
         offsetof(rp_result, struct_size), offsetof(rp_params, dt));
  { rp_params p = RP_PARAMS_INIT; rp_cost c = RP_COST_INIT; rp_grids g = RP_GRIDS_INIT; rp_result r = RP_RESULT_INIT;
    printf("%u %u %u %u\n", p.struct_size, c.struct_size, g.struct_size, r.struct_size); }
  return 0; }
'''
    src = tmp_path / "layout.c"
    src.write_text(prog)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-Wall", "-Wno-unused-function", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = [int(v) for v in out]
    # a caller written against ABI version 1 still compiles: its plan entries are inline wrappers in the header (compile only: no library here)
    old = tmp_path / "v1_caller.c"
    old.write_text(r'''
#include "rp_amd.h"
int v1_calls(rp_ctx *c, const rp_params *p, const rp_cost *k, const rp_grids *g, rp_result *r, double *out, const int32_t *dims, int32_t *lvl) {
  int rc = rp_plan_begin(c, p, k, g, 0, -1, 1);
  if (!rc) rc = rp_plan_wait(c, r, out);
  if (!rc) rc = rp_plan_packed(c, p, k, 3, 3, 3, r, out);
  if (!rc) rc = rp_plan_levels_begin(c, p, k, 2, g, 1);
  if (!rc) rc = rp_plan_levels_packed(c, p, k, 2, dims, r, out, lvl);
  if (!rc) rc = rp_plan_coeffs_grouped(c, p, k, 0, 0, 0, 0, 0, 0, 0, 0, r, out);
  return rc; }
''')
    subprocess.check_call(["gcc", "-Wall", "-Werror", "-c", "-I", os.path.join(REPO, "include"), "-o", str(tmp_path / "v1_caller.o"), str(old)])
    assert sizes[:4] == [C.sizeof(_capi.RpParams), C.sizeof(_capi.RpCost), C.sizeof(_capi.RpGrids),
                         C.sizeof(_capi.RpResult)]
    assert sizes[4:8] == [_capi.RpParams.x0_lon.offset, _capi.RpParams.v_delta_max.offset,
                          _capi.RpResult.reason_counts.offset, _capi.RpResult.kernel_ms.offset]
    assert sizes[8:13] == [0, 0, 0, 0, _capi.RpParams.dt.offset]          # every struct starts with its size
    assert sizes[13:17] == sizes[:4]                                       # RP_*_INIT
    assert [_capi.RpParams().struct_size, _capi.RpCost().struct_size, _capi.RpGrids().struct_size, _capi.RpResult().struct_size] == sizes[:4]


def test_no_environment_reads_on_the_plan_path():
    """Launch-policy switches are options of the context (rp_set_option); the environment is read in ONE place, the defaults of
    rp_create (options_from_environment), and once for the mailbox's wait budget -- nowhere between an rp_plan* entry and its return."""
    src = open(os.path.join(REPO, "commonroad-reactive-planner_amd", "csrc", "rp_host.hip")).read()
    lines = [ln for ln in src.splitlines() if "getenv" in ln and not ln.lstrip().startswith("//")]
    assert len(lines) == 2, lines
    assert "d.env ? std::getenv(d.env)" in lines[0] and "RP_AMD_MAILBOX_TIMEOUT_S" in lines[1]
    for other in ("rp_kernels.h", "rp_device.h", "rp_frontend.h", "rp_corridor.h", "rp_math.h"):
        assert "getenv" not in open(os.path.join(REPO, "commonroad-reactive-planner_amd", "csrc", other)).read(), other


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_capi.RpLibraryMissing):
        _capi.load_library(str(tmp_path / "nope.so"))


def test_create_without_gpu_reports_error_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_capi.RpError):
        _capi.RpContext(0)
