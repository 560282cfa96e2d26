/*
 * mcpt_fmath.h -- the transcendental functions of the hot path, written out in plain IEEE arithmetic.
 *
 * The reference calls the platform's libm at five places on the path: std::cos/std::sin of the GGX azimuth
 * (Material.hpp:114-119), of the lens sample (Renderer.cpp:58-60) and of Sphere::Sample (Sphere.hpp:66-67), and
 * std::atan2/std::acos of Scene::sampleEnv (Scene.hpp:66-67); and once after it: std::pow of the tone map (Renderer.cpp:99-101).  Their last bit is platform-dependent there (glibc,
 * Apple libm and the GPU's ocml all differ), so any implementation within an ulp is equally faithful -- but a CPU
 * checker and the GPU kernels only follow the SAME paths if both use the SAME one.  This header is that one
 * implementation: C99 and HIP C++, only + - * / sqrt, floor and int<->double conversions on doubles (each correctly rounded
 * on x86-64 and on gfx950), fixed operation order, no FMA (every includer compiles with -ffp-contract=off).
 * Float in, float out; the double result (error < 1e-13) is rounded once, so the float is the correctly rounded value except
 * within ~1e-6 ulp of a rounding boundary (tests/test_fmath.py measures <= 1 ulp against glibc everywhere).
 */
#ifndef MCPT_FMATH_H
#define MCPT_FMATH_H

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define MCPT_FM static __host__ __device__ __forceinline__
#define MCPT_FM_FLOOR(x) floor(x)
#define MCPT_FM_SQRT(x) sqrt(x)
#else
#include <math.h>
#define MCPT_FM static inline
#define MCPT_FM_FLOOR(x) floor(x)
#define MCPT_FM_SQRT(x) sqrt(x)
#endif

#define MCPT_FM_PI 3.14159265358979323846264338327950288
#define MCPT_FM_PIO2 1.57079632679489661923132169163975144
#define MCPT_FM_PIO4 0.785398163397448309615660845819875721

/* sin and cos of one argument.  Meant for the |x| <= 2 pi (1 + ulp) the path produces; accurate to float precision for
 * |x| < 1e5 (two-part pi/2: the first part has 33 significant bits, so k * part1 is exact for k < 2^20). */
MCPT_FM void mcpt_sincosf(float x, float *s_out, float *c_out) {
    const double xd = (double)x;
    const double kd = MCPT_FM_FLOOR(xd * 0.636619772367581343075535053490057448 + 0.5); /* nearest multiple of pi/2 */
    const double r = (xd - kd * 1.57079632673412561417e+00) - kd * 6.07710050650619224932e-11;
    const double z = r * r;
    /* Taylor polynomials on |r| <= pi/4 (+ a hair): truncation < 2e-14 (sin, through r^13) and < 1e-15 (cos, through r^14) */
    const double ps = -1.66666666666666666667e-01 +
                      z * (8.33333333333333333333e-03 +
                           z * (-1.98412698412698412698e-04 +
                                z * (2.75573192239858906526e-06 + z * (-2.50521083854417187751e-08 + z * 1.60590438368216145994e-10))));
    const double sn = r + r * (z * ps);
    const double pc = -5.00000000000000000000e-01 +
                      z * (4.16666666666666666667e-02 +
                           z * (-1.38888888888888888889e-03 +
                                z * (2.48015873015873015873e-05 +
                                     z * (-2.75573192239858906526e-07 + z * (2.08767569878680989792e-09 + z * -1.14707455977297247139e-11)))));
    const double cs = 1.0 + z * pc;
    const long long k = (long long)kd;
    const int q = (int)(k & 3);
    double s, c;
    if (q == 0) {
        s = sn;
        c = cs;
    } else if (q == 1) {
        s = cs;
        c = -sn;
    } else if (q == 2) {
        s = -sn;
        c = -cs;
    } else {
        s = -cs;
        c = sn;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

/* atan of 0 <= w <= 1 in double */
MCPT_FM double mcpt_fm_atan01(double w) {
    double base = 0.0;
    if (w > 0.414213562373095048801688724209698079) { /* tan(pi/8): atan(w) = pi/4 + atan((w-1)/(w+1)) */
        base = MCPT_FM_PIO4;
        w = (w - 1.0) / (w + 1.0);
    }
    /* Taylor series on |w| <= tan(pi/8), through w^29: truncation < 3e-13 */
    const double z = w * w;
    double p = 3.44827586206896551724e-02; /* 1/29 */
    p = -3.70370370370370370370e-02 + z * p; /* 1/27 */
    p = 4.00000000000000000000e-02 + z * p;  /* 1/25 */
    p = -4.34782608695652173913e-02 + z * p; /* 1/23 */
    p = 4.76190476190476190476e-02 + z * p;  /* 1/21 */
    p = -5.26315789473684210526e-02 + z * p; /* 1/19 */
    p = 5.88235294117647058824e-02 + z * p;  /* 1/17 */
    p = -6.66666666666666666667e-02 + z * p; /* 1/15 */
    p = 7.69230769230769230769e-02 + z * p;  /* 1/13 */
    p = -9.09090909090909090909e-02 + z * p; /* 1/11 */
    p = 1.11111111111111111111e-01 + z * p;  /* 1/9 */
    p = -1.42857142857142857143e-01 + z * p; /* 1/7 */
    p = 2.00000000000000000000e-01 + z * p;  /* 1/5 */
    p = -3.33333333333333333333e-01 + z * p; /* 1/3 */
    return base + (w + w * (z * p));
}

MCPT_FM double mcpt_fm_atan2(double y, double x) {
    const double ay = y < 0 ? -y : y, ax = x < 0 ? -x : x;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax) a = mcpt_fm_atan01(ay / ax);
    else a = MCPT_FM_PIO2 - mcpt_fm_atan01(ax / ay);
    /* quadrant from the sign BITS, as atan2 does (x = -0 counts as negative) */
    const int xneg = (x < 0) || (x == 0.0 && (1.0 / x) < 0);
    const int yneg = (y < 0) || (y == 0.0 && (1.0 / y) < 0);
    if (xneg) a = MCPT_FM_PI - a;
    return yneg ? -a : a;
}

MCPT_FM float mcpt_atan2f(float y, float x) { return (float)mcpt_fm_atan2((double)y, (double)x); }

/* acos(x) = 2 atan2(sqrt(1 - x), sqrt(1 + x)); NaN outside [-1, 1] (the square root of a negative number) */
MCPT_FM float mcpt_acosf(float x) {
    const double xd = (double)x;
    return (float)(2.0 * mcpt_fm_atan2(MCPT_FM_SQRT(1.0 - xd), MCPT_FM_SQRT(1.0 + xd)));
}

/* pow(x, y) for the tone map's std::pow(c, 0.45f) (Renderer.cpp:99-101): exp(y log x) in double.
 *   log:  x = m 2^e with m in [sqrt(1/2), sqrt(2));  log m = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716, series through s^23
 *   exp:  t = y log x = k ln2 + r, |r| <= ln2/2, Taylor through r^13; 2^k by writing the exponent field
 * Special values as std::pow for y > 0 non-integer: +-0 -> 0, x < 0 -> NaN, NaN -> NaN, +inf -> +inf. */
MCPT_FM double mcpt_fm_log(double x) { /* x positive, finite, normal or subnormal */
    union { double d; unsigned long long u; } v;
    v.d = x;
    int e = (int)((v.u >> 52) & 0x7ff);
    if (e == 0) { /* subnormal: scale into the normal range */
        v.d = x * 18014398509481984.0; /* 2^54 */
        e = (int)((v.u >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    v.u = (v.u & 0x000fffffffffffffull) | 0x3ff0000000000000ull; /* m in [1, 2) */
    double m = v.d;
    if (m > 1.41421356237309504880) {
        m = m * 0.5;
        e += 1;
    }
    const double s = (m - 1.0) / (m + 1.0), z = s * s;
    double p = 4.34782608695652173913e-02;   /* 1/23 */
    p = 4.76190476190476190476e-02 + z * p;  /* 1/21 */
    p = 5.26315789473684210526e-02 + z * p;  /* 1/19 */
    p = 5.88235294117647058824e-02 + z * p;  /* 1/17 */
    p = 6.66666666666666666667e-02 + z * p;  /* 1/15 */
    p = 7.69230769230769230769e-02 + z * p;  /* 1/13 */
    p = 9.09090909090909090909e-02 + z * p;  /* 1/11 */
    p = 1.11111111111111111111e-01 + z * p;  /* 1/9 */
    p = 1.42857142857142857143e-01 + z * p;  /* 1/7 */
    p = 2.00000000000000000000e-01 + z * p;  /* 1/5 */
    p = 3.33333333333333333333e-01 + z * p;  /* 1/3 */
    const double lm = 2.0 * (s + s * (z * p));
    return (double)e * 6.93147180369123816490e-01 + ((double)e * 1.90821492927058770002e-10 + lm); /* ln2 in two parts */
}

MCPT_FM double mcpt_fm_exp(double t) { /* |t| < 700 */
    const double kd = MCPT_FM_FLOOR(t * 1.44269504088896340736 + 0.5);
    const double r = (t - kd * 6.93147180369123816490e-01) - kd * 1.90821492927058770002e-10;
    double p = 1.60590438368216145994e-10;    /* 1/13! */
    p = 2.08767569878680989792e-09 + r * p;   /* 1/12! */
    p = 2.50521083854417187751e-08 + r * p;   /* 1/11! */
    p = 2.75573192239858906526e-07 + r * p;   /* 1/10! */
    p = 2.75573192239858906526e-06 + r * p;   /* 1/9! */
    p = 2.48015873015873015873e-05 + r * p;   /* 1/8! */
    p = 1.98412698412698412698e-04 + r * p;   /* 1/7! */
    p = 1.38888888888888888889e-03 + r * p;   /* 1/6! */
    p = 8.33333333333333333333e-03 + r * p;   /* 1/5! */
    p = 4.16666666666666666667e-02 + r * p;   /* 1/4! */
    p = 1.66666666666666666667e-01 + r * p;   /* 1/3! */
    p = 5.00000000000000000000e-01 + r * p;   /* 1/2! */
    const double er = 1.0 + (r + r * (r * p));
    union { double d; unsigned long long u; } s;
    s.u = (unsigned long long)((long long)kd + 1023) << 52; /* 2^k, k within the normal exponent range for |t| < 700 */
    return er * s.d;
}

MCPT_FM float mcpt_powf(float x, float y) {
    if (x != x || y != y) return x + y;                  /* NaN */
    if (x == 0.0f) return 0.0f;                          /* (y > 0 on this path) */
    if (x < 0.0f) return (x - x) / (x - x);              /* NaN, as std::pow for a non-integer exponent */
    if (x > 3.402823466e+38f) return x;                  /* +inf */
    double t = (double)y * mcpt_fm_log((double)x);
    if (t > 700.0) t = 700.0;                            /* the float conversion below overflows to +inf / underflows to 0 */
    if (t < -700.0) t = -700.0;
    return (float)mcpt_fm_exp(t);
}

/* Renderer.cpp:95-103: raw = (unsigned char) clamp(0, 255, 255 * pow(c, 0.45f)); clamp with std::min/max semantics (NaN -> 255) */
MCPT_FM unsigned char mcpt_tonemap_byte(float c) {
    const float v = 255 * mcpt_powf(c, 0.45f);
    const float a = (255.f < v) ? 255.f : ((v < 255.f) ? v : 255.f); /* std::min(hi, v): hi when v is NaN */
    const float b = (0.f < a) ? a : 0.f;
    return (unsigned char)b;
}

#endif /* MCPT_FMATH_H */
