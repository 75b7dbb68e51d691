/*
 * fovpt_detmath.h -- deterministic transcendental functions (part of libfovpt's contract).
 *
 * The reference calls sinf/cosf/acosf/atan2f/logf/powf from the CUDA math library
 * (built with --use_fast_math, CMakeLists.txt:181), whose results are not reproducible
 * off an NVIDIA device.  libfovpt defines its results in terms of the functions below:
 * every one is evaluated in IEEE binary64 with +,-,*,/ and sqrt only (no fused
 * multiply-add: build with -ffp-contract=off) and rounded once to binary32, so the same
 * source gives the same bits under g++ on the host and under hipcc on gfx950.  The double
 * intermediate keeps each result within 0.5 ulp + 1e-9 ulp of the true value, i.e. it
 * agrees with a correctly rounded libm in all but a vanishing fraction of arguments.
 *
 * Call sites in the reference: Probe.cuh:40-41,53-55,161 (acosf, atan2, sinf, cosf),
 * Disney.cuh:63 (logf), :216-217,293-294 (sinf, cosf), maths.h:250-251,261 (cosf, sinf),
 * cuda/helpers.h:38 (powf).
 */
#ifndef FOVPT_DETMATH_H
#define FOVPT_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define FOVPT_HD __host__ __device__ inline
#else
#define FOVPT_HD static inline
#endif

FOVPT_HD double fovpt_dm_bits2d(uint64_t u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
FOVPT_HD uint64_t fovpt_dm_d2bits(double d) { uint64_t u; __builtin_memcpy(&u, &d, 8); return u; }
FOVPT_HD uint32_t fovpt_dm_f2bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

/* floor for |x| < 2^51 without libm */
FOVPT_HD double fovpt_dm_floor(double x)
{
    double t = (double)(long long)x;      /* truncation toward zero */
    return (t > x) ? t - 1.0 : t;
}

/* sin and cos of a binary32 argument, |x| < 1e6 */
FOVPT_HD void fovpt_dm_sincos(float xf, float* s_out, float* c_out)
{
    const double x = (double)xf;
    const double kd = fovpt_dm_floor(x * 0.63661977236758138 + 0.5);
    const long long k = (long long)kd;
    /* x - k*pi/2 with pi/2 split in two doubles */
    const double r = (x - kd * 1.5707963267948966) - kd * 6.123233995736766e-17;
    const double z = r * r;
    /* Taylor, |r| <= pi/4: truncation < 3e-14 relative */
    double ps = -1.0 / 1307674368000.0;            /* -1/15! */
    ps = ps * z + 1.0 / 6227020800.0;              /*  1/13! */
    ps = ps * z - 1.0 / 39916800.0;                /* -1/11! */
    ps = ps * z + 1.0 / 362880.0;                  /*  1/9!  */
    ps = ps * z - 1.0 / 5040.0;                    /* -1/7!  */
    ps = ps * z + 1.0 / 120.0;                     /*  1/5!  */
    ps = ps * z - 1.0 / 6.0;                       /* -1/3!  */
    const double sr = r + r * (z * ps);
    double pc = 1.0 / 20922789888000.0;            /*  1/16! */
    pc = pc * z - 1.0 / 87178291200.0;             /* -1/14! */
    pc = pc * z + 1.0 / 479001600.0;               /*  1/12! */
    pc = pc * z - 1.0 / 3628800.0;                 /* -1/10! */
    pc = pc * z + 1.0 / 40320.0;                   /*  1/8!  */
    pc = pc * z - 1.0 / 720.0;                     /* -1/6!  */
    pc = pc * z + 1.0 / 24.0;                      /*  1/4!  */
    pc = pc * z - 0.5;                             /* -1/2!  */
    const double cr = 1.0 + z * pc;
    double s, c;
    switch ((int)(k & 3)) {
    case 0:  s = sr;  c = cr;  break;
    case 1:  s = cr;  c = -sr; break;
    case 2:  s = -sr; c = -cr; break;
    default: s = -cr; c = sr;  break;
    }
    *s_out = (float)s;
    *c_out = (float)c;
}

FOVPT_HD float fovpt_dm_sinf(float x) { float s, c; fovpt_dm_sincos(x, &s, &c); return s; }
FOVPT_HD float fovpt_dm_cosf(float x) { float s, c; fovpt_dm_sincos(x, &s, &c); return c; }

/* atan of t in [0,1] (double in, double out) */
FOVPT_HD double fovpt_dm_atan01(double t)
{
    double base = 0.0;
    if (t > 0.41421356237309503) {       /* tan(pi/8): atan t = pi/4 + atan((t-1)/(t+1)) */
        t = (t - 1.0) / (t + 1.0);
        base = 0.78539816339744831;
    }
    const double z = t * t;
    /* odd Taylor series to t^29, |t| <= 0.4143: truncation < 1e-13 */
    double p = 1.0 / 29.0;
    p = -1.0 / 27.0 + z * p;  p = 1.0 / 25.0 + z * p;
    p = -1.0 / 23.0 + z * p;  p = 1.0 / 21.0 + z * p;
    p = -1.0 / 19.0 + z * p;  p = 1.0 / 17.0 + z * p;
    p = -1.0 / 15.0 + z * p;  p = 1.0 / 13.0 + z * p;
    p = -1.0 / 11.0 + z * p;  p = 1.0 / 9.0 + z * p;
    p = -1.0 / 7.0 + z * p;   p = 1.0 / 5.0 + z * p;
    p = -1.0 / 3.0 + z * p;
    return base + (t + t * (z * p));
}

/* atan2 on doubles, full quadrant handling (signed zeros as C99) */
FOVPT_HD double fovpt_dm_atan2d(double y, double x)
{
    const int yneg = (int)(fovpt_dm_d2bits(y) >> 63);
    const int xneg = (int)(fovpt_dm_d2bits(x) >> 63);
    const double ay = yneg ? -y : y;
    const double ax = xneg ? -x : x;
    double a;
    if (ax == 0.0 && ay == 0.0) a = 0.0;
    else if (ay <= ax)          a = fovpt_dm_atan01(ay / ax);
    else                        a = 1.5707963267948966 - fovpt_dm_atan01(ax / ay);
    if (xneg) a = 3.1415926535897931 - a;
    return yneg ? -a : a;
}

FOVPT_HD float fovpt_dm_atan2f(float y, float x) { return (float)fovpt_dm_atan2d((double)y, (double)x); }

/* acos of x in [-1,1] */
FOVPT_HD float fovpt_dm_acosf(float xf)
{
    const double x = (double)xf;
    const double s2 = (1.0 - x) * (1.0 + x);
    return (float)fovpt_dm_atan2d(__builtin_sqrt(s2), x);
}

/* natural log of a positive finite double */
FOVPT_HD double fovpt_dm_logd(double x)
{
    uint64_t b = fovpt_dm_d2bits(x);
    int e = (int)((b >> 52) & 0x7ff);
    if (e == 0) {                          /* subnormal double: rescale */
        x = x * 18014398509481984.0;       /* 2^54 */
        b = fovpt_dm_d2bits(x);
        e = (int)((b >> 52) & 0x7ff) - 54;
    }
    e -= 1023;
    double m = fovpt_dm_bits2d((b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL);   /* [1,2) */
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0);   /* |s| <= 0.1716 */
    const double z = s * s;
    double p = 1.0 / 21.0;
    p = 1.0 / 19.0 + z * p;  p = 1.0 / 17.0 + z * p;  p = 1.0 / 15.0 + z * p;
    p = 1.0 / 13.0 + z * p;  p = 1.0 / 11.0 + z * p;  p = 1.0 / 9.0 + z * p;
    p = 1.0 / 7.0 + z * p;   p = 1.0 / 5.0 + z * p;   p = 1.0 / 3.0 + z * p;
    const double lm = 2.0 * (s + s * (z * p));
    return (double)e * 0.69314718055994529 + lm;
}

FOVPT_HD float fovpt_dm_logf(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return (float)fovpt_dm_bits2d(0x7ff8000000000000ULL);
    if (x == 0.0f) return (float)fovpt_dm_bits2d(0xfff0000000000000ULL);
    if (fovpt_dm_f2bits(x) == 0x7f800000u) return x;
    return (float)fovpt_dm_logd((double)x);
}

/* exp of a double in [-700, 700] */
FOVPT_HD double fovpt_dm_expd(double d)
{
    const double kd = fovpt_dm_floor(d * 1.4426950408889634 + 0.5);
    const double r = (d - kd * 0.69314718055994529) - kd * 2.3190468138462996e-17;
    /* Taylor to r^13, |r| <= 0.3466: truncation < 1e-17 */
    double p = 1.0 / 6227020800.0;
    p = 1.0 / 479001600.0 + r * p;  p = 1.0 / 39916800.0 + r * p;
    p = 1.0 / 3628800.0 + r * p;    p = 1.0 / 362880.0 + r * p;
    p = 1.0 / 40320.0 + r * p;      p = 1.0 / 5040.0 + r * p;
    p = 1.0 / 720.0 + r * p;        p = 1.0 / 120.0 + r * p;
    p = 1.0 / 24.0 + r * p;         p = 1.0 / 6.0 + r * p;
    p = 0.5 + r * p;
    const double er = 1.0 + (r + r * (r * p));
    const long long k = (long long)kd;
    return er * fovpt_dm_bits2d((uint64_t)(k + 1023) << 52);
}

/* powf for x >= 0 (the sRGB curve, cuda/helpers.h:37-38) */
FOVPT_HD float fovpt_dm_powf(float x, float y)
{
    if (x != x || y != y) return x + y;
    if (y == 0.0f) return 1.0f;
    if (x == 0.0f) return (y > 0.0f) ? 0.0f : (float)fovpt_dm_bits2d(0x7ff0000000000000ULL);
    if (x < 0.0f) return (float)fovpt_dm_bits2d(0x7ff8000000000000ULL);
    if (x == 1.0f) return 1.0f;
    double d = (double)y * fovpt_dm_logd((double)x);
    if (d > 700.0) d = 700.0;
    if (d < -700.0) d = -700.0;
    return (float)fovpt_dm_expd(d);
}

#endif /* FOVPT_DETMATH_H */
