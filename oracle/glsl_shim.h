/*
 * oracle/glsl_shim.h — TEST INFRASTRUCTURE.  Not part of the product, not part of the oracle's own arithmetic.
 *
 * Just enough of the GLSL type system for the reference's pure shader functions (backends/gpu-rt/shaders/{utils,random,structs,disney,
 * intersection}.glsl) to compile as C++ where they lie: oracle/make_glsl_ref.py reads those files IN PLACE, applies a handful of textual
 * rewrites (qualifiers, float literal suffixes, `.xyz`, conversion constructors) and includes the result behind this header.  The library
 * that comes out (oracle/_ref/libglsl_ref.so, never committed, never shipped) is held against the oracle's hand-written twins function by
 * function, bit for bit, by tests/test_glsl_differential.py: a transliteration slip that sits in BOTH textual twins of the reference
 * (oracle/oracle.cpp and csrc/shade_device.h) cannot hide there, because this third build is made from the reference's own text.
 *
 * It is a stand-in RUNTIME (the rules say so: it pins nothing about the reference's results on a GPU), so DESIGN.md keeps "parity unpinned".
 * Every built-in whose evaluation GLSL leaves open gets the ONE meaning oracle/glsl.h pins (dot / normalize / mix / min / max / reflect, the
 * elementary functions of include/rfw_detmath.h, saturating float -> int conversions); vector operators are component-wise single IEEE
 * binary32 operations in source order.  Compile with -ffp-contract=off.
 */
#ifndef ORACLE_GLSL_SHIM_H
#define ORACLE_GLSL_SHIM_H

#include <cmath>
#include <cstdint>

#include "glsl.h"

namespace glslref {

typedef uint32_t uint;

struct vec2 {
    float x, y;
    vec2() : x(0), y(0) {}
    vec2(float a, float b) : x(a), y(b) {}
};
struct vec4;
struct vec3 {
    float x, y, z;
    vec3() : x(0), y(0), z(0) {}
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    explicit vec3(const vec4& v);
    vec3& xyz_() { return *this; } // `v.xyz` on a vec3, also as an l-value (utils.glsl: contribution.xyz = ...)
    const vec3& xyz_() const { return *this; }
};
struct vec4 {
    float x, y, z, w;
    vec4() : x(0), y(0), z(0), w(0) {}
    explicit vec4(float s) : x(s), y(s), z(s), w(s) {}
    vec4(float a, float b, float c, float d) : x(a), y(b), z(c), w(d) {}
    vec4(const vec3& v, float d) : x(v.x), y(v.y), z(v.z), w(d) {}
    vec3 xyz_() const { return vec3(x, y, z); }
    float& operator[](int i) { return (&x)[i]; }
    float operator[](int i) const { return (&x)[i]; }
};
inline vec3::vec3(const vec4& v) : x(v.x), y(v.y), z(v.z) {}
struct ivec3 {
    int x, y, z;
    ivec3() : x(0), y(0), z(0) {}
    explicit ivec3(const vec3& v) : x(orc::f2i(v.x)), y(orc::f2i(v.y)), z(orc::f2i(v.z)) {} // float -> int: truncation, saturating (v_cvt_i32_f32)
};
struct ivec4 { int x, y, z, w; };
struct uvec4 { uint x, y, z, w; };
struct bvec4 {
    bool v[4];
    bvec4() : v{false, false, false, false} {}
    bool& operator[](int i) { return v[i]; }
    bool operator[](int i) const { return v[i]; }
};
struct mat4 { vec4 c[4]; };

// ---- component-wise operators: one IEEE operation per component, in source order
#define GLSLREF_OP3(op)                                                                                                   \
    inline vec3 operator op(const vec3& a, const vec3& b) { return vec3(a.x op b.x, a.y op b.y, a.z op b.z); }            \
    inline vec3 operator op(const vec3& a, float s) { return vec3(a.x op s, a.y op s, a.z op s); }                         \
    inline vec3 operator op(float s, const vec3& a) { return vec3(s op a.x, s op a.y, s op a.z); }                         \
    inline vec4 operator op(const vec4& a, const vec4& b) { return vec4(a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w); } \
    inline vec4 operator op(const vec4& a, float s) { return vec4(a.x op s, a.y op s, a.z op s, a.w op s); }               \
    inline vec4 operator op(float s, const vec4& a) { return vec4(s op a.x, s op a.y, s op a.z, s op a.w); }
GLSLREF_OP3(+)
GLSLREF_OP3(-)
GLSLREF_OP3(*)
GLSLREF_OP3(/)
#undef GLSLREF_OP3
inline vec3 operator-(const vec3& a) { return vec3(-a.x, -a.y, -a.z); }
inline vec4 operator-(const vec4& a) { return vec4(-a.x, -a.y, -a.z, -a.w); }
inline vec4& operator+=(vec4& a, const vec4& b) { a = a + b; return a; }
inline vec3& operator+=(vec3& a, const vec3& b) { a = a + b; return a; }
inline vec3& operator*=(vec3& a, const vec3& b) { a = a * b; return a; }
inline vec3& operator*=(vec3& a, float s) { a = a * s; return a; }

// ---- built-ins, with the meanings oracle/glsl.h pins
inline float dot(const vec3& a, const vec3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline vec3 cross(const vec3& a, const vec3& b) { return vec3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float sqrt(float x) { return std::sqrt(x); }
inline vec3 normalize(const vec3& a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline float length(const vec3& a) { return std::sqrt(dot(a, a)); }
inline float max(float a, float b) { return orc::gl_max(a, b); }
inline float min(float a, float b) { return orc::gl_min(a, b); }
inline vec4 max(const vec4& a, const vec4& b) { return vec4(max(a.x, b.x), max(a.y, b.y), max(a.z, b.z), max(a.w, b.w)); }
inline vec4 min(const vec4& a, const vec4& b) { return vec4(min(a.x, b.x), min(a.y, b.y), min(a.z, b.z), min(a.w, b.w)); }
inline float clamp(float x, float lo, float hi) { return orc::gl_clamp(x, lo, hi); }
inline uint clamp(uint x, uint lo, uint hi) { return x < lo ? lo : (x > hi ? hi : x); }
inline float abs(float x) { return orc::gl_abs(x); }
inline float gl_sign_f(float x) { return orc::gl_sign(x); } // (`sign(` is rewritten: utils.glsl also names a variable `sign`)
inline float mix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 mix(const vec3& a, const vec3& b, float t) { return a * (1.0f - t) + b * t; }
inline vec3 reflect(const vec3& I, const vec3& N) { return I - (2.0f * dot(N, I)) * N; }
inline float sin(float x) { return rfw_sinf(x); }
inline float cos(float x) { return rfw_cosf(x); }
inline float log(float x) { return rfw_logf(x); }
inline float exp(float x) { return rfw_expf(x); }
inline vec3 exp(const vec3& v) { return vec3(rfw_expf(v.x), rfw_expf(v.y), rfw_expf(v.z)); }
inline int floatBitsToInt(float f) { return (int)orc::fbits(f); }
inline float intBitsToFloat(int i) { return orc::bitsf((uint32_t)i); }
inline float intBitsToFloat(uint i) { return orc::bitsf(i); }
inline float intBitsToFloat(long long i) { return orc::bitsf((uint32_t)i); } // (int & 0xFFFFFFFC promotes in C++)
inline bvec4 greaterThanEqual(const vec4& a, const vec4& b) { bvec4 r; for (int i = 0; i < 4; i++) r[i] = a[i] >= b[i]; return r; }
inline bvec4 lessThan(const vec4& a, const vec4& b) { bvec4 r; for (int i = 0; i < 4; i++) r[i] = a[i] < b[i]; return r; }
inline bool any(const bvec4& b) { return b[0] || b[1] || b[2] || b[3]; }
// conversion "constructors" (`uint(x)`, `int(x)`, `float(x)` are rewritten to these): saturating like the hardware conversions
inline uint to_uint(float x) { return orc::f2u(x); }
inline uint to_uint(uint x) { return x; }
inline uint to_uint(int x) { return (uint)x; }
inline int to_int(float x) { return orc::f2i(x); }
inline int to_int(int x) { return x; }
inline int to_int(uint x) { return (int)x; }
inline float to_float(float x) { return x; }
inline float to_float(int x) { return (float)x; }
inline float to_float(uint x) { return (float)x; }

} // namespace glslref
#endif
