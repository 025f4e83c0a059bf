/*
 * oracle/glsl.h — TEST INFRASTRUCTURE (CPU oracle).  Not part of the product.
 *
 * Scalar restatement of the GLSL built-ins the reference kernels use
 * (the .glsl and .comp files under backends/gpu-rt/shaders).  GLSL leaves evaluation order and
 * contraction of these built-ins to the implementation; for parity at the bit
 * level the oracle pins ONE meaning for each, written out below, each operation
 * a single IEEE-754 binary32 op evaluated strictly left to right.  The HIP
 * kernels are written against the same definitions (DESIGN.md "arithmetic
 * contract").  Compile with -ffp-contract=off.
 */
#ifndef ORACLE_GLSL_H
#define ORACLE_GLSL_H

#include <cmath>
#include <cstdint>
#include <cstring>

#include "../include/rfw_detmath.h"

namespace orc {

struct vec2 { float x, y; };
struct vec3 { float x, y, z; };
struct vec4 { float x, y, z, w; };
struct mat4 { vec4 c[4]; }; // column-major, as glam / GLSL

inline vec3 V3(float x, float y, float z) { return vec3{x, y, z}; }
inline vec3 V3(float s) { return vec3{s, s, s}; }
inline vec4 V4(vec3 v, float w) { return vec4{v.x, v.y, v.z, w}; }
inline vec3 xyz(vec4 v) { return vec3{v.x, v.y, v.z}; }

inline vec3 operator+(vec3 a, vec3 b) { return vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, vec3 b) { return vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 operator*(vec3 a, float s) { return vec3{a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(float s, vec3 a) { return vec3{s * a.x, s * a.y, s * a.z}; }
inline vec3 operator/(vec3 a, float s) { return vec3{a.x / s, a.y / s, a.z / s}; }
inline vec3 operator+(vec3 a, float s) { return vec3{a.x + s, a.y + s, a.z + s}; }
inline vec3 operator+(float s, vec3 a) { return vec3{s + a.x, s + a.y, s + a.z}; }
inline vec3 operator-(vec3 a) { return vec3{-a.x, -a.y, -a.z}; }
inline vec4 operator+(vec4 a, vec4 b) { return vec4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline vec4 operator*(vec4 a, float s) { return vec4{a.x * s, a.y * s, a.z * s, a.w * s}; }
inline vec4 operator*(float s, vec4 a) { return vec4{s * a.x, s * a.y, s * a.z, s * a.w}; }
inline vec3 operator-(vec3 a, float s) { return vec3{a.x - s, a.y - s, a.z - s}; }
inline vec2 operator+(vec2 a, vec2 b) { return vec2{a.x + b.x, a.y + b.y}; }
inline vec2 operator*(vec2 a, float s) { return vec2{a.x * s, a.y * s}; }

// dot/cross: sum left to right, no FMA
inline float dot(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(vec3 a, vec3 b) { return vec3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float length(vec3 a) { return std::sqrt(dot(a, a)); }
// normalize(v) = v * (1 / sqrt(dot(v, v)))  (glam: v * length_recip)
inline vec3 normalize(vec3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }

// GLSL min/max on non-NaN inputs; a NaN operand yields the other one (what SPIR-V FMin/FMax do on GCN/CDNA).
inline float gl_max(float a, float b) { if (a != a) return b; if (b != b) return a; return a < b ? b : a; }
inline float gl_min(float a, float b) { if (a != a) return b; if (b != b) return a; return b < a ? b : a; }
inline vec3 gl_max(vec3 a, vec3 b) { return vec3{gl_max(a.x, b.x), gl_max(a.y, b.y), gl_max(a.z, b.z)}; }
inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
inline float gl_abs(float x) { return rfw_absf(x); }
inline float gl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
inline float gl_mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 gl_mix(vec3 a, vec3 b, float t) { return a * (1.0f - t) + b * t; }
// reflect(I, N) = I - 2 * dot(N, I) * N
inline vec3 gl_reflect(vec3 I, vec3 N) { return I - (2.0f * dot(N, I)) * N; }
inline bool gl_isnan(float x) { return x != x; }

// mat4 * vec4 = ((c0*x + c1*y) + c2*z) + c3*w
inline vec4 mul(const mat4& m, vec4 v) { return ((m.c[0] * v.x + m.c[1] * v.y) + m.c[2] * v.z) + m.c[3] * v.w; }

// float <-> int conversions with saturation (v_cvt_i32_f32 / v_cvt_u32_f32 semantics; NaN -> 0)
inline int32_t f2i(float x)
{
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
inline uint32_t f2u(float x)
{
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)x;
}
inline uint32_t fbits(float f) { return rfw_f2u(f); }
inline float bitsf(uint32_t u) { return rfw_u2f(u); }

} // namespace orc
#endif
