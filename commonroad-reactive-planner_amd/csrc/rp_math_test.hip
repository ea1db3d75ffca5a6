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
    if (i < n) { out[i] = a; if (out2) out2[i] = b; }
}

extern "C" int rpt_math(int kind, int n, const double *in, double *out, double *out2) {
    double *d_in = nullptr, *d_out = nullptr, *d_out2 = nullptr;
    if (hipMalloc((void **)&d_in, sizeof(double) * n) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_out, sizeof(double) * n) != hipSuccess) return -1;
    if (hipMalloc((void **)&d_out2, sizeof(double) * n) != hipSuccess) return -1;
    hipMemcpy(d_in, in, sizeof(double) * n, hipMemcpyHostToDevice);
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
