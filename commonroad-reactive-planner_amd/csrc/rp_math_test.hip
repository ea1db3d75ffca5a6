// rp_math_test.hip -- exposes the device math of rp_math.h to tests/test_gpu_math.py
// (TEST HELPER, not part of the product ABI; built as lib/librp_mathtest.so).
#include <hip/hip_runtime.h>
#include "rp_math.h"

__global__ void k_math(int kind, int n, const double *in, double *out, double *out2) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    double x = i < n ? in[i] : 1.0;
    double a = 0.0, b = 0.0;
    if (kind == 0) a = rp_atan(x);
    else if (kind == 1) rp_sincos(x, &a, &b);
    else if (kind == 2) a = rp_rcp(x);
    else if (kind == 3) a = rp_rsqrt(x);
    else if (kind == 4) {   // ext_travel(v, a, m, dt) with in = [v | a | m | dt] packed in four planes of n
        a = ext_travel(x, i < n ? in[n + i] : 0.0, i < n ? (int)in[2 * n + i] : 0, i < n ? in[3 * n + i] : 0.1);
    }
    if (i < n) { out[i] = a; if (out2) out2[i] = b; }
}

extern "C" int rpt_math(int kind, int n, const double *in, double *out, double *out2) {
    double *d_in = nullptr, *d_out = nullptr, *d_out2 = nullptr;
    const int planes = kind == 4 ? 4 : 1;
    if (hipMalloc((void **)&d_in, sizeof(double) * n * planes) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_out, sizeof(double) * n) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_out2, sizeof(double) * n) != hipSuccess) return -1;
    hipMemcpy(d_in, in, sizeof(double) * n * planes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_math, dim3((n + 255) / 256), dim3(256), 0, 0, kind, n, d_in, d_out, d_out2);
    hipMemcpy(out, d_out, sizeof(double) * n, hipMemcpyDeviceToHost);
    if (out2) hipMemcpy(out2, d_out2, sizeof(double) * n, hipMemcpyDeviceToHost);
    hipError_t e = hipDeviceSynchronize();
    hipFree(d_in); hipFree(d_out); hipFree(d_out2);
    return e == hipSuccess ? 0 : -2;
}

// ---- store-bandwidth ceiling probe (profiles/store_ceiling.py): plain coalesced streaming stores of `bytes` bytes,
//      16 B per lane (wide) or 8 B per lane, grid-stride; returns the average kernel time in ms over `reps` launches
__global__ void k_fill16(double2 *p, size_t n2) {
    const double2 v = make_double2(1.0, 2.0);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void k_fill8(double *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
extern "C" double rpt_store_ceiling(size_t bytes, int wide, int reps) {
    void *d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return -1.0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 8;
    for (int r = 0; r < 2; ++r) {
        if (wide) hipLaunchKernelGGL(k_fill16, dim3(grid), dim3(256), 0, 0, (double2 *)d, bytes / 16);
        else hipLaunchKernelGGL(k_fill8, dim3(grid), dim3(256), 0, 0, (double *)d, bytes / 8);
    }
    hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) {
        if (wide) hipLaunchKernelGGL(k_fill16, dim3(grid), dim3(256), 0, 0, (double2 *)d, bytes / 16);
        else hipLaunchKernelGGL(k_fill8, dim3(grid), dim3(256), 0, 0, (double *)d, bytes / 8);
    }
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return (double)ms / reps;
}

// ---- instruction-cost probe (diagnostic): cycles per instruction of one wavefront for the instruction kinds
//      the evaluation kernel is made of, as a dependent chain and as independent streams, with 1 or 2
//      wavefronts per SIMD.  s_memtime ticks around 64 x REP instructions.
#define UB_REP8(x) x x x x x x x x
#define UB_REP64(x) UB_REP8(UB_REP8(x))
__global__ void k_ubench(int which, int iters, unsigned long long *out, double *sink) {
    double x0 = 1.0 + threadIdx.x * 1e-9, x1 = 1.1, x2 = 1.2, x3 = 1.3, x4 = 1.4, x5 = 1.5, x6 = 1.6, x7 = 1.7;
    const double ca = 0.999999, cb = 1e-7;
    unsigned u0 = threadIdx.x, u1 = 7, u2 = 1, u3 = 2, u4 = 3;
    unsigned long long p0 = threadIdx.x;
    __shared__ double lds[64];
    lds[threadIdx.x & 63] = 1.0;
    __syncthreads();
    unsigned ldsaddr = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; ++it) {
        switch (which) {
        // every case: 8 asm blocks of 8 instructions (the compiler puts one s_nop behind each asm block: 64 + 8 issue slots)
#define UB8(one) one "\n\t" one "\n\t" one "\n\t" one "\n\t" one "\n\t" one "\n\t" one "\n\t" one
        case 0: UB_REP8(asm volatile(UB8("v_fma_f64 %0, %0, %1, %2") : "+v"(x0) : "v"(ca), "v"(cb));) break;
        case 1: UB_REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n\tv_fma_f64 %1, %1, %8, %9\n\tv_fma_f64 %2, %2, %8, %9\n\tv_fma_f64 %3, %3, %8, %9\n\t"
                                     "v_fma_f64 %4, %4, %8, %9\n\tv_fma_f64 %5, %5, %8, %9\n\tv_fma_f64 %6, %6, %8, %9\n\tv_fma_f64 %7, %7, %8, %9"
                                     : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(ca), "v"(cb));) break;
        case 2: UB_REP8(asm volatile(UB8("v_mul_f64 %0, %0, %1") : "+v"(x0) : "v"(ca));) break;
        case 3: UB_REP8(asm volatile(UB8("v_add_f64 %0, %0, %1") : "+v"(x0) : "v"(cb));) break;
        case 4: UB_REP8(asm volatile(UB8("v_rcp_f64 %0, %0") : "+v"(x0));) break;
        case 5: UB_REP8(asm volatile(UB8("v_rsq_f64 %0, %0") : "+v"(x0));) break;
        case 6: UB_REP8(asm volatile(UB8("v_readlane_b32 s20, %0, 5") ::"v"(u0) : "s20");) break;
        case 7: UB_REP8(asm volatile(UB8("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf") : "+v"(u0));) break;
        case 8: UB_REP8(asm volatile(UB8("v_cndmask_b32 %0, %0, %1, vcc") : "+v"(u0) : "v"(u1) : );) break;
        case 9: UB_REP8(asm volatile(UB8("v_lshl_add_u64 %0, %0, 3, %1") : "+v"(p0) : "v"(p0));) break;
        case 10: UB_REP8(asm volatile(UB8("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)") : "=v"(x0) : "v"(ldsaddr));) break;
        case 11: UB_REP8(asm volatile(UB8("v_cmp_lt_f64 vcc, %0, %1") ::"v"(x0), "v"(x1) : "vcc");) break;
        case 12: UB_REP8(asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t"
                                      "v_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7"
                                      : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));) break;
        case 13: UB_REP8(asm volatile(UB8("v_add_u32 %0, %0, %1") : "+v"(u0) : "v"(u1));) break;
        case 14: UB_REP8(asm volatile(UB8("s_add_u32 s20, s20, 1") ::: "s20", "scc");) break;
        case 15: UB_REP8(asm volatile(UB8("v_mul_lo_u32 %0, %0, %1") : "+v"(u0) : "v"(u1));) break;
        case 16: UB_REP8(asm volatile(UB8("v_writelane_b32 %0, s20, 3") : "+v"(u0) :: );) break;
        case 17: UB_REP8(asm volatile(UB8("ds_read_b64 %0, %1") : "=v"(x0) : "v"(ldsaddr));) asm volatile("s_waitcnt lgkmcnt(0)"); break;
        case 18: UB_REP8(asm volatile(UB8("v_rndne_f64 %0, %0") : "+v"(x0));) break;
        case 19: UB_REP8(asm volatile("v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4\n\t"
                                      "v_add_u32 %0, %0, %4\n\tv_add_u32 %1, %1, %4\n\tv_add_u32 %2, %2, %4\n\tv_add_u32 %3, %3, %4"
                                      : "+v"(u0), "+v"(u2), "+v"(u3), "+v"(u4) : "v"(u1));) break;
        case 20: UB_REP8(asm volatile("v_fma_f64 %0, %0, %2, %3\n\ts_add_u32 s20, s20, 1\n\tv_fma_f64 %1, %1, %2, %3\n\ts_add_u32 s21, s21, 1\n\t"
                                      "v_fma_f64 %0, %0, %2, %3\n\ts_add_u32 s20, s20, 1\n\tv_fma_f64 %1, %1, %2, %3\n\ts_add_u32 s21, s21, 1"
                                      : "+v"(x0), "+v"(x1) : "v"(ca), "v"(cb) : "s20", "s21", "scc");) break;
        default: break;
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
    sink[threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + (double)u0 + (double)p0 + (double)(u2 + u3 + u4);
}

// returns ticks per instruction (mean over the wavefronts of one workgroup of `waves` wavefronts)
extern "C" double rpt_ubench(int which, int waves, int iters) {
    unsigned long long *d_out = nullptr;
    double *d_sink = nullptr;
    if (hipMalloc(&d_out, sizeof(unsigned long long) * 64) != hipSuccess) return -1.0;
    if (hipMalloc(&d_sink, sizeof(double) * 1024) != hipSuccess) return -1.0;
    hipLaunchKernelGGL(k_ubench, dim3(1), dim3(64 * waves), 0, 0, which, 4, d_out, d_sink);
    hipLaunchKernelGGL(k_ubench, dim3(1), dim3(64 * waves), 0, 0, which, iters, d_out, d_sink);
    unsigned long long h[64];
    if (hipMemcpy(h, d_out, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost) != hipSuccess) return -1.0;
    hipFree(d_out); hipFree(d_sink);
    double sum = 0.0;
    for (int w = 0; w < waves; ++w) sum += (double)h[w];
    return sum / waves / (64.0 * iters);
}

// What a hipEvent bracket adds to a short kernel (diagnostic for bench.py's roofline.kernel_ms): median elapsed time, in
// microseconds, between two events recorded on one stream with (a) nothing, (b) one kernel that spins for `spin_us` on the
// 100 MHz constant clock between them.  out[0] = (a), out[1] = (b).
__global__ void k_spin(unsigned long long ticks, unsigned long long *sink) {
    unsigned long long t0, t1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    do {
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    } while (t1 - t0 < ticks);
    if (threadIdx.x == 0 && blockIdx.x == 0) *sink = t1 - t0;
}
extern "C" int rpt_event_bracket(double spin_us, int reps, double *out) {
    hipStream_t st;
    hipEvent_t e0, e1;
    unsigned long long *d_sink = nullptr;
    if (hipStreamCreate(&st) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return -1;
    if (hipMalloc(&d_sink, 8) != hipSuccess) return -1;
    const unsigned long long ticks = (unsigned long long)(spin_us * 100.0);
    for (int mode = 0; mode < 2; ++mode) {
        double acc[64];
        const int n = reps < 64 ? reps : 64;
        for (int r = 0; r < n + 3; ++r) {
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 100ull, d_sink);   // some earlier work on the stream, as in a step
            hipEventRecord(e0, st);
            if (mode == 1) hipLaunchKernelGGL(k_spin, dim3(465), dim3(256), 0, st, ticks, d_sink);
            hipEventRecord(e1, st);
            hipStreamSynchronize(st);
            float ms = 0.f;
            hipEventElapsedTime(&ms, e0, e1);
            if (r >= 3) acc[r - 3] = ms * 1e3;
        }
        for (int i = 1; i < n; ++i) { double v = acc[i]; int j = i - 1; while (j >= 0 && acc[j] > v) { acc[j + 1] = acc[j]; --j; } acc[j + 1] = v; }
        out[mode] = acc[n / 2];
    }
    hipFree(d_sink); hipEventDestroy(e0); hipEventDestroy(e1); hipStreamDestroy(st);
    return 0;
}
