// knot_tanh.h — tanh for the planar / smooth-terrain complementarity rows (tau = tanh(k_t h), knot_body.h) with a shorter dependent chain
// than the device library's: -expm1(-2|x|) / (expm1(-2|x|) + 2), expm1 by ln2 reduction and a degree-14 polynomial without the leading
// 1.  640 against 1176 cycles per call with two waves per SIMD; 2.5 ulp against 0.85 ulp (profiles/r02_math_probe.txt) — far inside
// the 1e-11 parity tolerance; saturates exactly (0 ulp at |x| = 1e9), NaN in -> NaN out.  One source for device and host: the host
// emulation and the layout recorder evaluate the same operations (std::fma), so the two stay bitwise comparable.
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

// q = q r + c, c a literal.  Device: ONE v_fma_f64 with the coefficient in a scalar register pair (two s_mov_b32 on the scalar unit).  Left
// to the compiler, a Horner step with a 64-bit literal addend becomes two v_mov_b32 — the literal into the destination of a two-address
// v_fmac — in front of the multiply-add: 24 of the 161 vector instructions of the contact-row task, on the unit the batch launches are
// bound by.  (A table in constant memory does not help: the compiler loads it into SGPRs and then copies them to VGPRs for the same
// v_fmac.)  The same operation on the same values either way.
#if defined(__HIP_DEVICE_COMPILE__)
#define FM_HORNER(q, r, c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(q) : "v"(q), "v"(r), "s"(double(c)))
#else
#define FM_HORNER(q, r, c) q = fm_fma(q, r, c)
#endif

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
    FM_HORNER(q, r, 1.0 / 6227020800.0);
    FM_HORNER(q, r, 1.0 / 479001600.0);
    FM_HORNER(q, r, 1.0 / 39916800.0);
    FM_HORNER(q, r, 1.0 / 3628800.0);
    FM_HORNER(q, r, 1.0 / 362880.0);
    FM_HORNER(q, r, 1.0 / 40320.0);
    FM_HORNER(q, r, 1.0 / 5040.0);
    FM_HORNER(q, r, 1.0 / 720.0);
    FM_HORNER(q, r, 1.0 / 120.0);
    FM_HORNER(q, r, 1.0 / 24.0);
    FM_HORNER(q, r, 1.0 / 6.0);
    const double r2 = r * r;
    // e^(r + rt) - 1 = (r + rt) + r^2 / 2 + r^3 q + rt (e^r - 1 ...) ~ r + (rt + rt r + r^2 / 2 + r^3 q)
    const double p = r + (fm_fma(rt, r, rt) + fm_fma(r2 * r, q, 0.5 * r2));
    const double s = fm_pow2i(k);
    return fm_fma(s, p, s - 1.0);
}

HD double knot_tanh(double x) {
    const double ax = std::fabs(x);
    if (!(ax == ax)) return x;                        // NaN
    const double t = fm_expm1_neg(-2.0 * ax);         // in (-1, 0]
    const double r = -t / (t + 2.0);
    return x < 0.0 ? -r : r;
}

}  // namespace hipnlp
