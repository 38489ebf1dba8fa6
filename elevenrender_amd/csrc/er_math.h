// er_math.h -- deterministic transcendental functions for the path-tracing hot path.
//
// The reference calls sycl::{sin,cos,acos,atan2,pow,log} (reference src/sycl.h:8,
// kernel.cpp:152-153,416-426; Texture.cpp:244-245,285-287; Sampling.h:35-52;
// Disney.cpp:57; HDRI.cpp:103).  Their precision is backend-defined, so no two
// reference targets agree bit for bit.  This header fixes ONE implementation that
// is compiled unchanged by hipcc (device) and g++ (host): every function evaluates
// a fixed sequence of IEEE-754 binary64 +,-,*,/,sqrt (no FMA contraction: build with
// -ffp-contract=off) and rounds once to binary32.  Internal error is < 1e-12
// relative, i.e. the float result is the correctly rounded one except when the
// exact value lies within ~1e-5 ulp of a rounding boundary -- which is also what
// glibc's sinf/cosf/logf/powf deliver, so the "libm" mode of the oracle agrees
// with this header on all but a vanishing fraction of arguments (tests/test_math.py).
//
// Polynomial coefficients: sin/cos use the classic fdlibm kernel coefficients
// (public domain, Sun Microsystems 1993); log/atan/exp use plain Taylor /
// atanh series whose coefficients are exact rationals.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ER_HD __host__ __device__ inline
#else
#define ER_HD inline
#endif

namespace ermath {

ER_HD uint64_t d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
ER_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
ER_HD uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
ER_HD float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }

ER_HD bool isnan_f(float x) { return x != x; }
ER_HD double nan_d() { return u2d(0x7ff8000000000000ull); }
ER_HD double inf_d() { return u2d(0x7ff0000000000000ull); }

// float -> int with the x86 cvttss2si convention the reference's CPU path gets
// (out of range or NaN -> INT_MIN); used for every (int)(u * width) on the path.
ER_HD int f2i(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : (int)0x80000000;
}

// ---- sin / cos ------------------------------------------------------------
// k = rint(x*2/pi); r = x - k*pi/2 in two pieces (33-bit head: exact product for
// |k| < 2^20, i.e. |x| < 1.6e6 rad).
ER_HD double sin_kernel(double r) {
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    double z = r * r;
    double p = S5 + z * S6;
    p = S4 + z * p;
    p = S3 + z * p;
    p = S2 + z * p;
    p = S1 + z * p;
    return r + (r * z) * p;
}
ER_HD double cos_kernel(double r) {
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    double z = r * r;
    double p = C5 + z * C6;
    p = C4 + z * p;
    p = C3 + z * p;
    p = C2 + z * p;
    p = C1 + z * p;
    return (1.0 - 0.5 * z) + (z * z) * p;
}
ER_HD double sincos_quadrant(double x, int add) {
    const double TWO_OVER_PI = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;   // pi/2 - PIO2_HI
    double k = __builtin_rint(x * TWO_OVER_PI);
    double r = (x - k * PIO2_HI) - k * PIO2_LO;
    int q = ((int)k + add) & 3;
    double v = (q & 1) ? cos_kernel(r) : sin_kernel(r);
    return (q & 2) ? -v : v;
}
ER_HD float er_sin(float x) {
    if (!(x > -1.0e6f && x < 1.0e6f)) return x - x;  // NaN for NaN/inf; 0 handled below by range
    return (float)sincos_quadrant((double)x, 0);
}
ER_HD float er_cos(float x) {
    if (!(x > -1.0e6f && x < 1.0e6f)) return (x - x) + ((x == x && x - x == 0.0f) ? 1.0f : 0.0f);
    return (float)sincos_quadrant((double)x, 1);
}

// ---- log ------------------------------------------------------------------
// x = 2^e * m, m in [sqrt(1/2), sqrt(2)); log m = 2 atanh((m-1)/(m+1)).
ER_HD double log_pos(double x, double* e_out) {
    uint64_t u = d2u(x);
    int e = (int)((u >> 52) & 0x7ff) - 1023;
    uint64_t mant = (u & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m = u2d(mant);
    if (m > 1.41421356237309514547) { m = m * 0.5; e = e + 1; }
    double s = (m - 1.0) / (m + 1.0);
    double z = s * s;
    double p = 1.0 / 19.0;
    p = 1.0 / 17.0 + z * p;
    p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = 1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = 1.0 / 3.0 + z * p;
    double lm = 2.0 * (s + (s * z) * p);
    *e_out = (double)e;
    return lm;  // natural log of the mantissa part
}
ER_HD float er_log(float x) {
    if (x != x) return x;
    if (x < 0.0f) return (float)nan_d();
    if (x == 0.0f) return (float)(-inf_d());
    if (x == (float)inf_d()) return x;
    const double LN2 = 6.93147180559945286227e-01;
    double e;
    double lm = log_pos((double)x, &e);   // float denormals are normal doubles
    return (float)(e * LN2 + lm);
}

// ---- pow ------------------------------------------------------------------
ER_HD double exp2_d(double t) {
    if (t > 1024.0) return inf_d();
    if (t < -1100.0) return 0.0;
    double n = __builtin_rint(t);
    double f = (t - n) * 6.93147180559945286227e-01;  // |f| <= 0.3466
    double p = 1.0 / 6227020800.0;                     // 1/13!
    p = 1.0 / 479001600.0 + f * p;
    p = 1.0 / 39916800.0 + f * p;
    p = 1.0 / 3628800.0 + f * p;
    p = 1.0 / 362880.0 + f * p;
    p = 1.0 / 40320.0 + f * p;
    p = 1.0 / 5040.0 + f * p;
    p = 1.0 / 720.0 + f * p;
    p = 1.0 / 120.0 + f * p;
    p = 1.0 / 24.0 + f * p;
    p = 1.0 / 6.0 + f * p;
    p = 0.5 + f * p;
    p = 1.0 + f * p;
    p = 1.0 + f * p;
    int ni = (int)n;
    // scale by 2^ni in two steps so results that are float-denormal stay exact enough
    int n1 = ni / 2, n2 = ni - n1;
    double s1 = u2d((uint64_t)(n1 + 1023) << 52);
    double s2 = u2d((uint64_t)(n2 + 1023) << 52);
    return (p * s1) * s2;
}
ER_HD float er_pow(float x, float y) {
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : (float)inf_d();
    if (x < 0.0f) return (float)nan_d();          // only non-integer exponents occur on the path
    if (x == (float)inf_d()) return (y > 0.0f) ? x : 0.0f;
    const double LOG2E = 1.44269504088896338700e+00;
    double e;
    double lm = log_pos((double)x, &e);
    double t = (double)y * (e + lm * LOG2E);
    return (float)exp2_d(t);
}

// ---- atan2 / acos ----------------------------------------------------------
ER_HD double atan_unit(double t) {   // t in [0,1]
    const double PIO4 = 7.85398163397448278999e-01;
    double base = 0.0;
    if (t > 0.41421356237309503) { base = PIO4; t = (t - 1.0) / (t + 1.0); }
    double z = t * t;                // z <= 0.1716
    double p = -1.0 / 31.0;
    p = 1.0 / 29.0 + z * p;
    p = -1.0 / 27.0 + z * p;
    p = 1.0 / 25.0 + z * p;
    p = -1.0 / 23.0 + z * p;
    p = 1.0 / 21.0 + z * p;
    p = -1.0 / 19.0 + z * p;
    p = 1.0 / 17.0 + z * p;
    p = -1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;
    p = -1.0 / 11.0 + z * p;
    p = 1.0 / 9.0 + z * p;
    p = -1.0 / 7.0 + z * p;
    p = 1.0 / 5.0 + z * p;
    p = -1.0 / 3.0 + z * p;
    return base + (t + (t * z) * p);
}
ER_HD double atan2_d(double y, double x) {
    const double PI = 3.14159265358979311600e+00, PIO2 = 1.57079632679489655800e+00;
    double ax = x < 0.0 ? -x : x, ay = y < 0.0 ? -y : y;
    double a;
    if (ax >= ay) a = (ax == 0.0) ? 0.0 : atan_unit(ay / ax);
    else a = PIO2 - atan_unit(ax / ay);
    if ((d2u(x) >> 63) != 0) a = PI - a;
    if ((d2u(y) >> 63) != 0) a = -a;
    return a;
}
ER_HD float er_atan2(float y, float x) {
    if (x != x || y != y) return x + y;
    return (float)atan2_d((double)y, (double)x);
}
ER_HD float er_acos(float x) {
    if (!(x >= -1.0f && x <= 1.0f)) return (float)nan_d();
    double xd = (double)x;
    double s = __builtin_sqrt((1.0 - xd) * (1.0 + xd));
    return (float)atan2_d(s, xd);
}

}  // namespace ermath
