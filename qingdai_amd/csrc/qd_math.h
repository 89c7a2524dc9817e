// qd_math.h -- tanh for x >= 0 without the extended-precision exponential of the device library.
//
// The cloud-source terms (run_simulation.py:1880-1903: np.tanh of a precipitation ratio, a temperature excess, a relative vorticity
// and a thermal advection, each clipped to [0, 1]) made k_cloud_fromp_source VALU-bound: ocml's tanh builds on a double-double
// exponential, ~140 f64 instructions a call, four calls a cell (381 v_add_f64 in the kernel).  Every argument there is >= 0 (or the
// term is dropped), so:
//     x <  0.25 : x + x^3 Q(x^2), the Taylor series through x^23 (next term < 2e-17 x)
//     x >= 0.25 : E / (E + 2), E = expm1(2x) by Cody-Waite reduction and the degree-13 Taylor polynomial on |r| <= ln2 / 2
//     x >  19.1 : 1
// Error <= 2 ulp against the correctly rounded result over [0, 40] (tests/test_host_math_cpu.py compiles this header with g++ and
// checks it against numpy's tanh, which is itself a 1-2 ulp SIMD routine): inside the 1e-12 tolerance the cloud fields are compared
// with.  Plain C++ so that the host test and the device build share the text; fma() is spelled out because the library is built with
// -ffp-contract=off.
#pragma once
#include <math.h>
#ifdef __HIPCC__
#define QD_MATH_FN __host__ __device__ __forceinline__
#else
#define QD_MATH_FN static inline
#endif

QD_MATH_FN double qd_expm1_pos(double y) {                    // exp(y) - 1, 0 <= y <= 700
    const double k = rint(y * 1.4426950408889634);           // y / ln2
    double r = fma(k, -6.93147180369123816490e-01, y);       // ln2 hi (fdlibm's split: the product k * hi is exact for |k| < 2^21)
    r = fma(k, -1.90821492927058770002e-10, r);              // ln2 lo
    double p = 1.0 / 6227020800.0;                           // 1/13!
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    const double pm1 = p * r;                                // exp(r) - 1 without the cancellation of "exp(r), then - 1"
    const double twok = ldexp(1.0, (int)k);
    return fma(twok, pm1, twok - 1.0);
}

QD_MATH_FN double qd_tanh_nonneg(double x) {                  // tanh(x) for x >= 0 (NaN stays NaN)
    const double s = x * x;
    double q = 18888466084.0 / 194896477400625.0;            // x^21
    q = fma(q, s, -443861162.0 / 1856156927625.0);           // x^19
    q = fma(q, s, 6404582.0 / 10854718875.0);                // x^17
    q = fma(q, s, -929569.0 / 638512875.0);                  // x^15
    q = fma(q, s, 21844.0 / 6081075.0);                      // x^13
    q = fma(q, s, -1382.0 / 155925.0);                       // x^11
    q = fma(q, s, 62.0 / 2835.0);                            // x^9
    q = fma(q, s, -17.0 / 315.0);                            // x^7
    q = fma(q, s, 2.0 / 15.0);                               // x^5
    q = fma(q, s, -1.0 / 3.0);                               // x^3
    const double small = fma(x * s, q, x);
    const double xe = x < 19.1 ? x : 19.1;                   // tanh(19.1) rounds to 1; keeps exp's argument small
    const double em1 = qd_expm1_pos(xe + xe);
    const double big = em1 / (em1 + 2.0);
    const double t = x < 0.25 ? small : big;
    return x != x ? x : t;
}

QD_MATH_FN double qd_tanh(double x) { return copysign(qd_tanh_nonneg(fabs(x)), x); }
