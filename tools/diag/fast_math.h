// Diagnostic (never part of the product; used by tools/diag/math_probe.hip only).
// Bounded-range fp64 sin / cos / tanh tried against the two library calls on the critical path of the knot program's first phase
// (the joint transforms: sincos of 23 joint angles; the planar complementarity rows: tanh of gain x height):
//   fast_sincos: |x| < 1e5: Cody-Waite reduction by pi/2 in three pieces with a tail, the classical degree-13 / degree-14 kernels;
//   fast_tanh:   -expm1(-2|x|) / (expm1(-2|x|) + 2), expm1 by ln2 reduction and a degree-14 polynomial without the leading 1.
// Measured (profiles/r02_math_probe.txt, one call per lane, two waves per SIMD): sincos 660 cycles (library) vs 620, tanh 1176 vs 640
// (2.5 ulp against 0.85).  In the kernel the tanh version shortened its task by 200 of 2 600 cycles and the launch by nothing —
// the phase is set by three waves within 50 cycles of each other — so the product keeps the library calls.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#ifndef HD
#if defined(__HIPCC__)
#define HD __host__ __device__ inline
#else
#define HD inline
#endif
#endif

namespace hipnlp {

HD double fm_fma(double a, double b, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_fma(a, b, c);
#else
    return std::fma(a, b, c);
#endif
}

// 2^k for -1022 <= k <= 1023
HD double fm_pow2i(int k) {
    const uint64_t bits = uint64_t(k + 1023) << 52;
    double r;
#if defined(__HIP_DEVICE_COMPILE__)
    r = __longlong_as_double((long long)bits);
#else
    std::memcpy(&r, &bits, sizeof r);
#endif
    return r;
}

HD void fast_sincos(double x, double* sn, double* cs) {
    if (!(std::fabs(x) < 1.0e5)) {   // (also NaN and the infinities) the library's general reduction
#if defined(__HIP_DEVICE_COMPILE__)
        ::sincos(x, sn, cs);
#else
        *sn = std::sin(x); *cs = std::cos(x);
#endif
        return;
    }
    // x = fn pi/2 + (y0 + y1), |y0| <= pi/4: pi/2 = P1 + P2 + P3 (33 bits each: fn P exact for |fn| < 2^20) + tail
    constexpr double INVPIO2 = 6.36619772367581382433e-01;
    constexpr double P1 = 1.57079632673412561417e+00, P2 = 6.07710050630396597660e-11, P2T = 2.02226624879595063154e-21;
    const double fn = std::rint(x * INVPIO2);
    const int n = int(fn);
    double t = fm_fma(-fn, P1, x);                    // exact
    double w = fn * P2;
    double r = t - w;
    w = fm_fma(fn, P2T, -((t - r) - w));              // what the subtraction lost + the next piece
    const double y0 = r - w;
    const double y1 = (r - y0) - w;
    // kernels on [-pi/4, pi/4] with the tail y1
    constexpr double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
                     S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    constexpr double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
                     C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = y0 * y0;
    const double v = z * y0;
    const double rs = fm_fma(z, fm_fma(z, fm_fma(z, fm_fma(z, S6, S5), S4), S3), S2);
    const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
    const double rc = z * fm_fma(z, fm_fma(z, fm_fma(z, fm_fma(z, fm_fma(z, C6, C5), C4), C3), C2), C1);
    const double hz = 0.5 * z;
    const double wc = 1.0 - hz;
    const double kc = wc + (((1.0 - wc) - hz) + (z * rc - y0 * y1));
    // quadrant n mod 4: (s, c), (c, -s), (-s, -c), (-c, s)
    const bool swap = n & 1;
    const double a = swap ? kc : ks, b = swap ? ks : kc;
    *sn = (n & 2) ? -a : a;
    *cs = ((n + 1) & 2) ? -b : b;
}

// expm1(y) for y <= 0 (any magnitude): y = k ln2 + r, |r| <= ln2 / 2; expm1(r) = r + r^2 / 2 + ... (no leading 1);
// expm1(y) = 2^k expm1(r) + (2^k - 1)
HD double fm_expm1_neg(double y) {
    constexpr double INVLN2 = 1.44269504088896338700e+00, LN2_HI = 6.93147180369123816490e-01, LN2_LO = 1.90821492927058770002e-10;
    const double yy = y < -80.0 ? -80.0 : y;          // e^-80 is below half an ulp of 1
    const double fk = std::rint(yy * INVLN2);
    const int k = int(fk);
    const double hi = fm_fma(-fk, LN2_HI, yy);        // exact (LN2_HI has 32 bits)
    const double lo = fk * LN2_LO;
    const double r = hi - lo;
    const double rt = (hi - r) - lo;                  // r + rt = the reduced argument to ~2^-100
    // Taylor coefficients 1/n!, n = 3..14, Horner in r
    double q = 1.0 / 87178291200.0;
    q = fm_fma(q, r, 1.0 / 6227020800.0);
    q = fm_fma(q, r, 1.0 / 479001600.0);
    q = fm_fma(q, r, 1.0 / 39916800.0);
    q = fm_fma(q, r, 1.0 / 3628800.0);
    q = fm_fma(q, r, 1.0 / 362880.0);
    q = fm_fma(q, r, 1.0 / 40320.0);
    q = fm_fma(q, r, 1.0 / 5040.0);
    q = fm_fma(q, r, 1.0 / 720.0);
    q = fm_fma(q, r, 1.0 / 120.0);
    q = fm_fma(q, r, 1.0 / 24.0);
    q = fm_fma(q, r, 1.0 / 6.0);
    const double r2 = r * r;
    // e^(r + rt) - 1 = (r + rt) + r^2 / 2 + r^3 q + rt (e^r - 1 ...) ~ r + (rt + rt r + r^2 / 2 + r^3 q)
    const double p = r + (fm_fma(rt, r, rt) + fm_fma(r2 * r, q, 0.5 * r2));
    const double s = fm_pow2i(k);
    return fm_fma(s, p, s - 1.0);
}

HD double fast_tanh(double x) {
    const double ax = std::fabs(x);
    if (!(ax == ax)) return x;                        // NaN
    const double t = fm_expm1_neg(-2.0 * ax);         // in (-1, 0]
    const double r = -t / (t + 2.0);
    return x < 0.0 ? -r : r;
}

}  // namespace hipnlp
