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


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
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
    assert lib.rp_abi_version() == 1


def test_struct_layout_matches_c(tmp_path):
    """sizeof/offsetof from a C translation unit compiled against the header."""
    prog = r'''
#include "rp_amd.h"   /* first: the header must be self-contained */
#include <stdio.h>
int main(void) {
  printf("%zu %zu %zu %zu\n", sizeof(rp_params), sizeof(rp_cost), sizeof(rp_grids), sizeof(rp_result));
  printf("%zu %zu %zu %zu\n", offsetof(rp_params, x0_lon), offsetof(rp_params, v_delta_max),
         offsetof(rp_result, reason_counts), offsetof(rp_result, kernel_ms));
  return 0; }
'''
    src = tmp_path / "layout.c"
    src.write_text(prog)
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(REPO, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    sizes = [int(v) for v in out]
    assert sizes[:4] == [C.sizeof(_capi.RpParams), C.sizeof(_capi.RpCost), C.sizeof(_capi.RpGrids),
                         C.sizeof(_capi.RpResult)]
    assert sizes[4:] == [_capi.RpParams.x0_lon.offset, _capi.RpParams.v_delta_max.offset,
                         _capi.RpResult.reason_counts.offset, _capi.RpResult.kernel_ms.offset]


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(_capi.RpLibraryMissing):
        _capi.load_library(str(tmp_path / "nope.so"))


def test_create_without_gpu_reports_error_instead_of_falling_back():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_capi.RpError):
        _capi.RpContext(0)
