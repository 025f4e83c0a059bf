// device_math.h — float3 helpers for the kernels, written against the arithmetic contract of
// every operation is ONE IEEE-754 binary32 operation, evaluated left to right, no FMA
// contraction (the translation unit is compiled with -ffp-contract=off and correctly rounded
// divide/sqrt).  GLSL built-ins used by the reference shaders get the same fixed meaning the
// oracle gives them (dot = x*x' + y*y' + z*z', normalize(v) = v * (1/sqrt(dot(v,v))), ...).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/rfw_detmath.h"

namespace rfwhip {

struct f3 {
    float x, y, z;
};
struct f2 {
    float x, y;
};

#define RFW_DI __device__ __forceinline__

RFW_DI f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RFW_DI f3 mk3(float s) { return f3{s, s, s}; }
RFW_DI f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RFW_DI f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RFW_DI f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RFW_DI f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RFW_DI f3 operator*(float s, f3 a) { return f3{s * a.x, s * a.y, s * a.z}; }
RFW_DI f3 operator/(f3 a, float s) { return f3{a.x / s, a.y / s, a.z / s}; }
RFW_DI f3 operator+(f3 a, float s) { return f3{a.x + s, a.y + s, a.z + s}; }
RFW_DI f3 operator+(float s, f3 a) { return f3{s + a.x, s + a.y, s + a.z}; }
RFW_DI f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
RFW_DI f2 operator+(f2 a, f2 b) { return f2{a.x + b.x, a.y + b.y}; }
RFW_DI f2 operator*(f2 a, float s) { return f2{a.x * s, a.y * s}; }

RFW_DI float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RFW_DI f3 cross(f3 a, f3 b) { return f3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RFW_DI float length(f3 a) { return __builtin_sqrtf(dot(a, a)); }
RFW_DI f3 normalize(f3 a) { return a * (1.0f / __builtin_sqrtf(dot(a, a))); }

RFW_DI float gl_max(float a, float b) { if (a != a) return b; if (b != b) return a; return a < b ? b : a; }
RFW_DI float gl_min(float a, float b) { if (a != a) return b; if (b != b) return a; return b < a ? b : a; }
RFW_DI f3 gl_max(f3 a, f3 b) { return f3{gl_max(a.x, b.x), gl_max(a.y, b.y), gl_max(a.z, b.z)}; }
RFW_DI float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
RFW_DI float gl_abs(float x) { return rfw_absf(x); }
RFW_DI float gl_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
RFW_DI f3 gl_mix(f3 a, f3 b, float t) { return a * (1.0f - t) + b * t; }
RFW_DI f3 gl_reflect(f3 I, f3 N) { return I - (2.0f * dot(N, I)) * N; }
RFW_DI bool gl_isnan(float x) { return x != x; }

// float <-> int with saturation, NaN -> 0 (the oracle's f2i/f2u)
RFW_DI int32_t f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)x;
}
RFW_DI uint32_t f2u(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}
RFW_DI uint32_t fbits(float f) { return __float_as_uint(f); }
RFW_DI float bitsf(uint32_t u) { return __uint_as_float(u); }

// row-form affine transform: rows r0..r2 = (m[i], m[4+i], m[8+i], m[12+i]); w = 1 (point) or 0 (vector).
// component i = ((m[i]*x + m[4+i]*y) + m[8+i]*z) + m[12+i]*w  — the column-sum order of mat4 * vec4.
RFW_DI f3 xform_rows(const float4 r0, const float4 r1, const float4 r2, f3 v, float w)
{
    return f3{((r0.x * v.x + r0.y * v.y) + r0.z * v.z) + r0.w * w, ((r1.x * v.x + r1.y * v.y) + r1.z * v.z) + r1.w * w,
              ((r2.x * v.x + r2.y * v.y) + r2.z * v.z) + r2.w * w};
}

// random.glsl:5-23
RFW_DI uint32_t wang_hash(uint32_t s)
{
    s = (s ^ 61u) ^ (s >> 16u);
    s *= 9u;
    s = s ^ (s >> 4u);
    s *= 0x27d4eb2du;
    s = s ^ (s >> 15u);
    return s;
}
RFW_DI uint32_t randi(uint32_t& s)
{
    s ^= s << 13;
    s ^= s >> 17;
    s ^= s << 5;
    return s;
}
RFW_DI float randf(uint32_t& s) { return (float)randi(s) * 2.3283064365387e-10f; }

} // namespace rfwhip
