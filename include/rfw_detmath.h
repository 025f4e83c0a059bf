/*
 * rfw_detmath.h — bit-reproducible elementary functions for the shading path.
 *
 * The reference's kernels call GLSL sin/cos/log/exp/acos/atan, whose results are
 * implementation-defined (Vulkan allows several ulp and differs per driver), so
 * "the value the reference computes" does not exist for them.  Radiance parity
 * at 1e-4 per pixel between a CPU and a GPU however needs every branch of a
 * path to be taken identically on both sides, i.e. identical bits.  This header
 * therefore fixes ONE definition of those functions, built only from IEEE-754
 * binary32 + - * / (each correctly rounded on x86-64 and on gfx950), integer
 * bit operations and float<->int conversions of in-range values.  It is part of
 * the boundary contract: the oracle (oracle/) and the HIP kernels
 * (rfw-rs_amd/csrc/) both include it; tests/test_detmath.py checks it against
 * libm (<= 2 ulp on the argument ranges the path uses).
 *
 * Polynomials are the classic single-precision minimax forms (Cephes-style
 * sinf/cosf/logf/expf/asinf/atanf).  Translation units including this header
 * MUST be compiled with -ffp-contract=off (no FMA fusion).
 */
#ifndef RFW_DETMATH_H
#define RFW_DETMATH_H

#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define RFW_HD __host__ __device__ static inline
#else
#define RFW_HD static inline
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif

RFW_HD uint32_t rfw_f2u(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
RFW_HD float rfw_u2f(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }
RFW_HD float rfw_absf(float x) { return rfw_u2f(rfw_f2u(x) & 0x7fffffffu); }

/* sin and cos of x for |x| <= 8192 (the path uses [0, 2*pi] and small multiples of pi/4.5). */
RFW_HD void rfw_sincosf(float x, float* s, float* c)
{
    const float FOPI = 1.27323954473516f; /* 4/pi */
    const float DP1 = 0.78515625f, DP2 = 2.4187564849853515625e-4f, DP3 = 3.77489497744594108e-8f;
    float ax = rfw_absf(x);
    int sign_s = (rfw_f2u(x) >> 31) != 0u;
    int32_t j = (int32_t)(FOPI * ax);
    if (j & 1) j += 1;
    float y = (float)j;
    j &= 7;
    int sign_c = 0;
    if (j > 3) { sign_s = !sign_s; sign_c = 1; j -= 4; }
    if (j > 1) sign_c = !sign_c;
    float r = ((ax - y * DP1) - y * DP2) - y * DP3;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float sv, cv;
    if (j == 1 || j == 2) { sv = pc; cv = ps; } else { sv = ps; cv = pc; }
    *s = sign_s ? -sv : sv;
    *c = sign_c ? -cv : cv;
}
RFW_HD float rfw_sinf(float x) { float s, c; rfw_sincosf(x, &s, &c); return s; }
RFW_HD float rfw_cosf(float x) { float s, c; rfw_sincosf(x, &s, &c); return c; }

/* natural log of a positive normal float; x <= 0 -> -inf / NaN-free sentinel is NOT needed by
 * the path (arguments are alpha^2 in [1e-6, 1]); denormals are treated as their scaled value. */
RFW_HD float rfw_logf(float x)
{
    uint32_t u = rfw_f2u(x);
    int32_t e = (int32_t)((u >> 23) & 0xffu) - 126;       /* frexp exponent: x = m * 2^e, m in [0.5, 1) */
    float m = rfw_u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                    + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    float fe = (float)e;
    y = y + -2.12194440e-4f * fe;
    y = y + -0.5f * z;
    float r = m + y;
    r = r + 0.693359375f * fe;
    return r;
}
RFW_HD float rfw_log2f(float x) { return rfw_logf(x) * 1.44269504088896341f; }

/* e^x, clamped to [0, +inf) outside the finite range. */
RFW_HD float rfw_expf(float x)
{
    if (!(x < 88.72283905206835f)) return rfw_u2f(0x7f800000u);
    if (x < -87.0f) return 0.0f;
    float fz = 1.44269504088896341f * x + 0.5f;
    int32_t n = (int32_t)fz;
    if ((float)n > fz) n -= 1; /* floor */
    float z = (float)n;
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    float zz = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
                + 1.6666665459e-1f) * x + 5.0000001201e-1f) * zz + x + 1.0f;
    /* ldexp: n in [-126, 128]; split so both factors are normal */
    int32_t n1 = n / 2, n2 = n - n1;
    return p * rfw_u2f((uint32_t)(n1 + 127) << 23) * rfw_u2f((uint32_t)(n2 + 127) << 23);
}

RFW_HD float rfw_sqrtf_(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_sqrtf(x);
#else
    return __builtin_sqrtf(x);
#endif
}

RFW_HD float rfw_asinf(float x)
{
    float a = rfw_absf(x);
    int neg = (rfw_f2u(x) >> 31) != 0u;
    if (a > 1.0f) a = 1.0f;
    float z, r;
    int flag = 0;
    if (a < 1.0e-4f) return x;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); r = rfw_sqrtf_(z); flag = 1; } else { r = a; z = r * r; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z
               + 1.6666752422e-1f) * z * r + r;
    if (flag) { p = p + p; p = 1.5707963267948966192f - p; }
    return neg ? -p : p;
}

RFW_HD float rfw_acosf(float x)
{
    if (x < -1.0f) x = -1.0f;
    if (x > 1.0f) x = 1.0f;
    if (x < -0.5f) return 3.14159265358979323846f - 2.0f * rfw_asinf(rfw_sqrtf_(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * rfw_asinf(rfw_sqrtf_(0.5f * (1.0f - x)));
    return 1.5707963267948966192f - rfw_asinf(x);
}

RFW_HD float rfw_atanf(float x)
{
    float a = rfw_absf(x);
    int neg = (rfw_f2u(x) >> 31) != 0u;
    float y;
    if (a > 2.414213562373095f) { y = 1.5707963267948966192f; a = -(1.0f / a); }
    else if (a > 0.4142135623730950f) { y = 0.7853981633974483096f; a = (a - 1.0f) / (a + 1.0f); }
    else y = 0.0f;
    float z = a * a;
    y = y + (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * a + a;
    return neg ? -y : y;
}

/* GLSL atan(y, x) */
RFW_HD float rfw_atan2f(float y, float x)
{
    const float PI = 3.14159265358979323846f, PIO2 = 1.5707963267948966192f;
    if (x == 0.0f) {
        if (y > 0.0f) return PIO2;
        if (y < 0.0f) return -PIO2;
        return 0.0f;
    }
    float a = rfw_atanf(y / x);
    if (x < 0.0f) return (rfw_f2u(y) >> 31) ? a - PI : a + PI;
    return a;
}

#endif /* RFW_DETMATH_H */
