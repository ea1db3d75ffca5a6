// rp_math.h -- lean FP64 math for the trajectory kernel (gfx950).
//
// The hot loop is FP64-VALU bound, so the transcendental budget matters.  ocml's generic
// double-precision tan / atan2 / sincos carry huge-argument paths (Payne-Hanek) that cost registers
// even when never taken.  The arguments here are bounded (|angles| <~ 4*pi, |d'| small), so:
//   rp_rcp / rp_rsqrt  : hardware estimate + Newton steps  (<= ~1.5 ulp)
//   rp_atan            : fdlibm-style 4-interval reduction + odd minimax polynomial (< 1 ulp)
//   rp_sincos          : Cody-Waite reduction by pi/2 (3 parts) + fdlibm kernels (< 1 ulp for |x| < 1e5)
// Accuracy is verified against NumPy on the device in tests/test_gpu_math.py.
#pragma once

#include <hip/hip_runtime.h>

// a * b + c as a three-address v_fma_f64.  For Horner steps whose addend is a loop-invariant coefficient held in a register:
// the compiler selects the two-address v_fmac_f64 there and puts a v_mov_b64 copy of the coefficient in front of every step
// (a wavefront of a small batch pays ~10 cycles per instruction whatever its kind: profiles/r01_instruction_costs.txt).
#ifndef RP_FMA3_ASM
#define RP_FMA3_ASM 1
#endif
__device__ __forceinline__ double rp_fma3(double a, double b, double c) {
#if RP_FMA3_ASM
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
#else
    return __builtin_fma(a, b, c);
#endif
}
// first Horner step: both coefficients are constants; the multiplier may sit in scalar registers (one scalar source per
// VOP3 instruction), which keeps it out of the vector registers the other coefficients occupy across the step loop
__device__ __forceinline__ double rp_fma3_first(double x, double c1, double c0) {
#if RP_FMA3_ASM
    double r;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "s"(c1), "v"(x), "v"(c0));
    return r;
#else
    return __builtin_fma(x, c1, c0);
#endif
}

// 1/x: v_rcp_f64 estimate + two Newton-Raphson steps
__device__ __forceinline__ double rp_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// 1/sqrt(x): v_rsq_f64 estimate + one third-order step  y*(1 + e/2 + 3e^2/8), e = 1 - x*y^2
__device__ __forceinline__ double rp_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    double r = x * y;
    double e = __builtin_fma(-r, y, 1.0);
    double t = __builtin_fma(0.375, e, 0.5);
    return __builtin_fma(y * e, t, y);
}

// atan(x), all x.  fdlibm s_atan.c scheme.  Wavefronts whose lanes all lie in the first interval (|x| < 7/16: the slope d' of
// a lateral offset over the arc length almost always does) skip the interval selection and the division altogether; the
// general path gives the same bits for such lanes (hi = lo = 0: 0 - ((p - 0) - t) == t - p).
__device__ __forceinline__ double rp_atan_poly(double t) {   // t * (odd minimax polynomial part), fdlibm aT[]
    const double z = t * t, w = z * z;
    double s1 = rp_fma3_first(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02);
    s1 = rp_fma3(w, s1, 6.66107313738753120669e-02);
    s1 = rp_fma3(w, s1, 9.09088713343650656196e-02);
    s1 = rp_fma3(w, s1, 1.42857142725034663711e-01);
    s1 = rp_fma3(w, s1, 3.33333333333329318027e-01);
    s1 = z * s1;
    double s2 = rp_fma3_first(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02);
    s2 = rp_fma3(w, s2, -7.69187620504482999495e-02);
    s2 = rp_fma3(w, s2, -1.11111104054623557880e-01);
    s2 = rp_fma3(w, s2, -1.99999999998764832476e-01);
    s2 = w * s2;
    return t * (s1 + s2);
}

__device__ __forceinline__ double rp_atan(double x) {
    const double ax = fabs(x);
    const bool r0 = ax < 0.4375;
    double t = ax, hi = 0.0, lo = 0.0;
    if (!__all(r0)) {   // wave-uniform
        // interval selection, reduced argument t = num / den
        const bool r1 = ax < 0.6875, r2 = ax < 1.1875, r3 = ax < 2.4375;
        const double num = r0 ? ax : (r1 ? __builtin_fma(2.0, ax, -1.0) : (r2 ? ax - 1.0 : (r3 ? ax - 1.5 : -1.0)));
        const double den = r0 ? 1.0 : (r1 ? 2.0 + ax : (r2 ? ax + 1.0 : (r3 ? __builtin_fma(1.5, ax, 1.0) : ax)));
        t = r0 ? num : num * rp_rcp(den);
        hi = r0 ? 0.0 : (r1 ? 4.63647609000806093515e-01 : (r2 ? 7.85398163397448278999e-01
                : (r3 ? 9.82793723247329054082e-01 : 1.57079632679489655800e+00)));
        lo = r0 ? 0.0 : (r1 ? 2.26987774529616870924e-17 : (r2 ? 3.06161699786838301793e-17
                : (r3 ? 1.39033110312309984516e-17 : 6.12323399573676603587e-17)));
        if (ax > 7.3786976294838206464e19) t = 0.0;   // |x| >= 2^66 (and inf, whose reciprocal iteration is NaN): pi/2
    }
    const double p = rp_atan_poly(t);
    const double res = hi - ((p - lo) - t);     // first interval: 0 - ((p - 0) - t) = t - p;  NaN stays NaN
    return copysign(res, x);
}

// sin and cos of x for |x| < ~1e5 (angles of this kernel are bounded by a few pi).
__device__ __forceinline__ void rp_sincos(double x, double *s, double *c) {
    const double n = rint(x * 6.36619772367581382433e-01);   // 2/pi
    double r = __builtin_fma(-n, 1.57079632673412561417e+00, x);   // pio2_1 (33 bits)
    const double y = __builtin_fma(-n, 6.07710050650619224932e-11, r);   // pio2_1t (remaining error ~ n * 7e-27)
    const double z = y * y;
    // __kernel_sin
    double ps = rp_fma3_first(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = rp_fma3(z, ps, 2.75573137070700676789e-06);
    ps = rp_fma3(z, ps, -1.98412698298579493134e-04);
    ps = rp_fma3(z, ps, 8.33333333332248946124e-03);
    ps = rp_fma3(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(y * z, ps, y);
    // __kernel_cos
    double pc = rp_fma3_first(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = rp_fma3(z, pc, -2.75573143513906633035e-07);
    pc = rp_fma3(z, pc, 2.48015872894767294178e-05);
    pc = rp_fma3(z, pc, -1.38888888888741095749e-03);
    pc = rp_fma3(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double cs = w1 + (((1.0 - w1) - hz) + z * z * pc);
    const int q = (int)n & 3;
    const double ss = (q & 1) ? cs : sn;
    const double cc = (q & 1) ? sn : cs;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}

// dt * sum_{t=1..m} max(0, v + t dt a): distance covered by the extension after m steps (trajectories.py:182-196),
// in closed form (the terms are an arithmetic progression clipped at zero); callers add a relative slack
__device__ __forceinline__ double ext_travel(double v, double acc, int m, double dt) {
    const double h = dt * acc;   // increment of the progression v + t h
    double t0 = 1.0, t1 = (double)m;   // the range of t whose terms are positive
    // (quotients through the Newton-refined reciprocal: a quotient off by an ulp moves the clipping point by one term only
    //  where that term is ~0 -- |v + t h| <= ulp(v) -- far inside the relative slack the callers add)
    if (h < 0.0) {
        if (!(v > 0.0)) return 0.0;
        t1 = fmin(t1, floor(v * rp_rcp(-h)));     // v + t h >= 0  <=>  t <= v / -h
    } else if (v < 0.0) {
        if (!(h > 0.0)) return 0.0;
        t0 = fmax(t0, floor(-v * rp_rcp(h)) + 1.0);   // v + t h > 0   <=>  t > -v / h
    }
    if (t1 < t0) return 0.0;
    const double cnt = t1 - t0 + 1.0;
    return dt * (cnt * v + h * 0.5 * (t0 + t1) * cnt);
}

