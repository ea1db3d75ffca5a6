"""Instruction-cost probe: s_memtime ticks per instruction of one wavefront (see rp_math_test.hip, k_ubench)."""
import ctypes, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(REPO, "commonroad-reactive-planner_amd", "lib", "librp_mathtest.so"))
lib.rpt_ubench.restype = ctypes.c_double
lib.rpt_ubench.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
names = ["v_fma_f64 dependent", "v_fma_f64 8 streams", "v_mul_f64 dep", "v_add_f64 dep", "v_rcp_f64 dep", "v_rsq_f64 dep",
         "v_readlane_b32", "v_mov_b32_dpp dep", "v_cndmask_b32 dep", "v_lshl_add_u64 dep", "ds_read_b64+wait", "v_cmp_lt_f64",
         "v_rcp_f64 8 streams", "v_add_u32 dep", "s_add_u32 dep", "v_mul_lo_u32 dep", "v_writelane_b32", "ds_read_b64 x64 then wait",
         "v_rndne_f64 dep", "v_add_u32 4 streams", "v_fma_f64 / s_add_u32 interleaved"]
print("ticks per instruction per wavefront (64 instructions + 8 compiler-inserted s_nop per 64); workgroup of W wavefronts on one CU (W=4: one per SIMD, W=8: two, W=12: three, W=16: four)")
print(f"{'instruction':34s} {'W=1':>8s} {'W=4':>8s} {'W=8':>8s} {'W=12':>8s} {'W=16':>8s}")
for k, nm in enumerate(names):
    print(f"{nm:34s} " + " ".join(f"{lib.rpt_ubench(k, w, 200):8.2f}" for w in (1, 4, 8, 12, 16)))
