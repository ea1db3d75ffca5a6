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
