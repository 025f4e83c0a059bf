// shade_device.h — material evaluation, BSDF sampling and light sampling for the shade kernel.
// Device restatement of backends/gpu-rt/shaders/{disney.glsl, utils.glsl, structs.glsl:217-270} and
// shade.comp:283-528 under the arithmetic contract of device_math.h.  Function names follow the GLSL.
#pragma once
#include "device_math.h"
#include "device_types.h"

namespace rfwhip {

#define RFW_PI 3.14159265359f
#define RFW_TWOPI (2.0f * 3.14159265359f)
#define RFW_INVPI (1.0f / 3.14159265359f)
#define RFW_INV2PI (1.0f / (2.0f * 3.14159265359f))

// utils.glsl:22-26
RFW_DI uint32_t PackNormal(f3 N)
{
    const float f = 65535.0f / __builtin_sqrtf(8.0f * N.z + 8.0f);
    return f2u(N.x * f + 32767.0f) + (f2u(N.y * f + 32767.0f) << 16);
}
// utils.glsl:28-35
RFW_DI f3 UnpackNormal(uint32_t p)
{
    float nx = (float)(p & 65535u) * (2.0f / 65535.0f);
    float ny = (float)(p >> 16) * (2.0f / 65535.0f);
    float nz = 0.0f;
    nx += -1.0f; ny += -1.0f; nz += 1.0f;
    float l = nx * -nx + ny * -ny + nz * -nz;
    nz = l;
    l = __builtin_sqrtf(l);
    nx *= l;
    ny *= l;
    return mk3(nx, ny, nz) * 2.0f + mk3(0.0f, 0.0f, -1.0f);
}
// utils.glsl:55-70
RFW_DI f3 DiffuseReflectionUniform(float r0, float r1)
{
    const float term1 = RFW_TWOPI * r0, term2 = __builtin_sqrtf(1.0f - r1 * r1);
    float s, c;
    rfw_sincosf(term1, &s, &c);
    return mk3(c * term2, s * term2, r1);
}
RFW_DI f3 DiffuseReflectionCosWeighted(float r0, float r1)
{
    const float term1 = RFW_TWOPI * r0;
    const float term2 = __builtin_sqrtf(1.0f - r1);
    float s, c;
    rfw_sincosf(term1, &s, &c);
    return mk3(c * term2, s * term2, __builtin_sqrtf(r1));
}
// utils.glsl:72-80
RFW_DI void CLAMPINTENSITY(f3& contribution, float clampValue)
{
    const float v = gl_max(contribution.x, gl_max(contribution.y, contribution.z));
    if (v > clampValue) {
        const float m = clampValue / v;
        contribution = contribution * m;
    }
}
// utils.glsl:83-92
RFW_DI float safe_origin_1(float o, float n)
{
    const int32_t of_i = f2i(256.0f * n);
    const float p_i = bitsf((uint32_t)((int32_t)fbits(o) + ((o < 0.0f) ? -of_i : of_i)));
    return gl_abs(o) < (1.0f / 32.0f) ? o + (1.0f / 65536.0f) * n : p_i;
}
RFW_DI f3 safe_origin(f3 O, f3 R, f3 N)
{
    const f3 _N = dot(N, R) > 0.0f ? N : -N;
    return mk3(safe_origin_1(O.x, _N.x), safe_origin_1(O.y, _N.y), safe_origin_1(O.z, _N.z));
}

// structs.glsl:177-270
struct ShadingData {
    f3 color, absorption, specular;
    float metallic, subsurface, specular_f, roughness, specular_tint, anisotropic, sheen, sheen_tint;
    float clearcoat, clearcoat_gloss, transmission, eta;
    // Material-only terms of the two BSDF evaluations and pdfs of a hit, made once by extractParameters (round 6: inside BSDFEval / Fr they sit
    // in lane-divergent branches, so the compiler evaluated them once per CALL — the logarithm of GTR1 and a division, twice per hit).  The
    // same expressions in the same order: GTR1(NDotH, a) = (a^2 - 1) / ((PI * log(a^2)) * t) with a = mix(.1, .001, clearcoat_gloss); 1 / eta.
    float cc_a, cc_a2m1, cc_pilog, inv_eta;
    f3 cspec0; // prepare_tint(): Cspec0 of BSDFEval from the (textured) colour
    uint32_t flags;
    int32_t diffuse_map, normal_map;
};
RFW_DI float CHAR2FLT(uint32_t x, int s) { return (float)((x >> s) & 255u) * (1.0f / 255.0f); }
RFW_DI ShadingData extractParameters(const rfw_device_material* m)
{
    // 96-B material = 6 dwordx4 loads
    const float4* mp = reinterpret_cast<const float4*>(m);
    const float4 c = mp[0], a = mp[1], s = mp[2];
    const uint4 p = *reinterpret_cast<const uint4*>(mp + 3);
    ShadingData d;
    const uint4 q = *reinterpret_cast<const uint4*>(mp + 4);
    d.flags = q.x;
    d.diffuse_map = (int32_t)q.y;
    d.normal_map = (int32_t)q.z;
    d.color = mk3(c.x, c.y, c.z);
    d.absorption = mk3(a.x, a.y, a.z);
    d.specular = mk3(s.x, s.y, s.z);
    d.metallic = CHAR2FLT(p.x, 0);
    d.subsurface = CHAR2FLT(p.x, 8);
    d.specular_f = CHAR2FLT(p.x, 16);
    d.roughness = gl_max(0.01f, CHAR2FLT(p.x, 24));
    d.specular_tint = CHAR2FLT(p.y, 0);
    d.anisotropic = CHAR2FLT(p.y, 8);
    d.sheen = CHAR2FLT(p.y, 16);
    d.sheen_tint = CHAR2FLT(p.y, 24);
    d.clearcoat = CHAR2FLT(p.z, 0);
    d.clearcoat_gloss = CHAR2FLT(p.z, 8);
    d.transmission = CHAR2FLT(p.z, 16);
    d.eta = CHAR2FLT(p.z, 24);
    d.cc_a = gl_mix(.1f, .001f, d.clearcoat_gloss);
    {
        const float a2 = d.cc_a * d.cc_a;
        d.cc_a2m1 = a2 - 1.0f;
        d.cc_pilog = RFW_PI * rfw_logf(a2);
    }
    d.inv_eta = 1.0f / d.eta;
    return d;
}

enum { BSDF_TYPE_REFLECTED = 0, BSDF_TYPE_TRANSMITTED = 1, BSDF_TYPE_SPECULAR = 2 };
RFW_DI float sqr(float x) { return x * x; }

// disney.glsl:13-25
RFW_DI bool Refract(f3 wi, f3 n, float eta, f3& wt)
{
    const float cosThetaI = dot(n, wi);
    const float sin2ThetaI = gl_max(0.0f, 1.0f - cosThetaI * cosThetaI);
    const float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1.0f) return false;
    const float cosThetaT = __builtin_sqrtf(1.0f - sin2ThetaT);
    wt = eta * (wi * -1.0f) + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
// disney.glsl:27-31
RFW_DI float SchlickFresnel(float u)
{
    const float m = gl_clamp(1.0f - u, 0.0f, 1.0f);
    return (m * m) * (m * m) * m;
}
// disney.glsl:54-59
RFW_DI float GTR2(float NDotH, float a)
{
    const float a2 = a * a;
    const float t = 1.0f + (a2 - 1.0f) * NDotH * NDotH;
    return a2 / (RFW_PI * t * t);
}
// disney.glsl:61-66
RFW_DI float SmithGGX(float NDotv, float alphaG)
{
    const float a = alphaG * alphaG;
    const float b = NDotv * NDotv;
    return 1.0f / (NDotv + __builtin_sqrtf(a + b - a * b));
}
// disney.glsl:45-52 (GTR1, for the clearcoat lobe's a = mix(.1, .001, clearcoat_gloss)) and disney.glsl:68-78 (Fr at the material's eta),
// through the terms extractParameters made once (same expressions, same order)
RFW_DI float GTR1_of(const ShadingData& sd, float NDotH)
{
    if (sd.cc_a >= 1.0f) return RFW_INVPI;
    const float t = 1.0f + sd.cc_a2m1 * NDotH * NDotH;
    return sd.cc_a2m1 / (sd.cc_pilog * t);
}
RFW_DI float Fr_of(const ShadingData& sd, float VDotN)
{
    const float SinThetaT2 = sqr(sd.eta) * (1.0f - VDotN * VDotN);
    if (SinThetaT2 > 1.0f) return 1.0f;
    const float LDotN = __builtin_sqrtf(1.0f - SinThetaT2);
    const float eta = sd.inv_eta;
    const float r1 = (VDotN - eta * LDotN) / (VDotN + eta * LDotN);
    const float r2 = (LDotN - eta * VDotN) / (LDotN + eta * VDotN);
    return 0.5f * (sqr(r1) + sqr(r2));
}
// disney.glsl:80-87
RFW_DI f3 SafeNormalize(f3 a)
{
    const float ls = dot(a, a);
    if (ls > 0.0f) return a * (1.0f / __builtin_sqrtf(ls));
    return mk3(0.0f);
}
// The half vector of a (wi, wo) pair, as far as BOTH the pdf (SafeNormalize, disney.glsl:80-87) and the evaluation (normalize) need it: the sum,
// its squared length and 1 / its length — one square root and one division per pair instead of one per function (round 6)
struct HalfVec { f3 v; float ls, inv; };
RFW_DI HalfVec half_of(f3 wi, f3 wo)
{
    HalfVec h;
    h.v = wi + wo;
    h.ls = dot(h.v, h.v);
    h.inv = 1.0f / __builtin_sqrtf(h.ls);
    return h;
}
// disney.glsl:89-108
// `F` = Fr_of(sd, dot(N, wo)): it does not depend on wi, and a hit evaluates this pdf for two directions over the same N and wo
RFW_DI float BSDFPdf(const ShadingData& sd, f3 N, f3 wo, f3 wi, const HalfVec& hh, const float F)
{
    float bsdfPdf = 0.0f, brdfPdf;
    if (dot(wi, N) <= 0.0f) {
        brdfPdf = RFW_INV2PI * sd.subsurface * 0.5f;
    } else {
        const f3 halfway = hh.ls > 0.0f ? hh.v * hh.inv : mk3(0.0f); // SafeNormalize(wi + wo)
        const float cosThetaHalf = gl_abs(dot(halfway, N));
        const float pdfHalf = GTR2(cosThetaHalf, sd.roughness) * cosThetaHalf;
        const float pdfSpec = 0.25f * pdfHalf / gl_max(1.e-6f, dot(wi, halfway));
        const float pdfDiff = gl_abs(dot(wi, N)) * RFW_INVPI * (1.0f - sd.subsurface);
        bsdfPdf = pdfSpec * F;
        brdfPdf = gl_mix(pdfDiff, pdfSpec, 0.5f);
    }
    return gl_mix(brdfPdf, bsdfPdf, sd.transmission);
}
RFW_DI float BSDFPdf(const ShadingData& sd, f3 N, f3 wo, f3 wi) { return BSDFPdf(sd, N, wo, wi, half_of(wi, wo), Fr_of(sd, dot(N, wo))); }
RFW_DI void prepare_tint(ShadingData& sd)
{
    const f3 Cdlin = sd.color;
    const float Cdlum = .3f * Cdlin.x + .6f * Cdlin.y + .1f * Cdlin.z;
    const f3 Ctint = Cdlum > 0.0f ? Cdlin / Cdlum : mk3(1.0f);
    sd.cspec0 = gl_mix(sd.specular * .08f * gl_mix(mk3(1.0f), Ctint, sd.specular_tint), Cdlin, sd.metallic);
}
// disney.glsl:110-195
RFW_DI f3 BSDFEval(const ShadingData& sd, f3 N, f3 wo, f3 wi, float t, bool backfacing, const HalfVec& hh)
{
    const float NDotL = dot(N, wi);
    const float NDotV = dot(N, wo);
    const f3 H = hh.v * hh.inv; // normalize(wi + wo)
    const float NDotH = dot(N, H);
    const float LDotH = dot(wi, H);
    const f3 Cdlin = sd.color;
    const f3 Cspec0 = sd.cspec0;
    f3 bsdf = mk3(0.0f);
    f3 brdf = mk3(0.0f);
    if (sd.transmission > 0.0f) {
        if (NDotL <= 0.0f) {
            const float F = Fr_of(sd, NDotV);
            bsdf = mk3((1.0f - F) / gl_abs(NDotL) * (1.0f - sd.metallic) * sd.transmission);
        } else {
            const float a = sd.roughness;
            const float Ds = GTR2(NDotH, a);
            const float FH = Fr_of(sd, LDotH);
            const f3 Fs = gl_mix(Cspec0, mk3(1.0f), FH);
            const float Gs = SmithGGX(NDotV, a) * SmithGGX(NDotL, a);
            bsdf = (Gs * Ds) * Fs;
        }
    }
    if (sd.transmission < 1.0f) {
        if (NDotL <= 0.0f) {
            if (sd.subsurface > 0.0f) {
                const f3 s = mk3(__builtin_sqrtf(sd.color.x), __builtin_sqrtf(sd.color.y), __builtin_sqrtf(sd.color.z));
                const float FL = SchlickFresnel(gl_abs(NDotL)), FV = SchlickFresnel(NDotV);
                const float Fd = (1.0f - 0.5f * FL) * (1.0f - 0.5f * FV);
                brdf = RFW_INVPI * s * sd.subsurface * Fd * (1.0f - sd.metallic);
            }
        } else {
            const float a = sd.roughness;
            const float Ds = GTR2(NDotH, a);
            const float FH = SchlickFresnel(LDotH);
            const f3 Fs = gl_mix(Cspec0, mk3(1.0f), FH);
            const float Gs = SmithGGX(NDotV, a) * SmithGGX(NDotL, a);
            const float FL = SchlickFresnel(NDotL), FV = SchlickFresnel(NDotV);
            const float Fd90 = 0.5f + 2.0f * LDotH * LDotH * a;
            const float Fd = gl_mix(1.0f, Fd90, FL) * gl_mix(1.0f, Fd90, FV);
            const float Dr = GTR1_of(sd, NDotH);
            const float Fc = gl_mix(.04f, 1.0f, FH);
            const float Gr = SmithGGX(NDotL, .25f) * SmithGGX(NDotV, .25f);
            brdf = RFW_INVPI * Fd * Cdlin * (1.0f - sd.metallic) * (1.0f - sd.subsurface) + Gs * Fs * Ds + mk3(sd.clearcoat * Gr * Fc * Dr);
        }
    }
    const f3 fin = gl_mix(brdf, bsdf, sd.transmission);
    if (backfacing) {
        const f3 a = -sd.absorption * t;
        return fin * mk3(rfw_expf(a.x), rfw_expf(a.y), rfw_expf(a.z));
    }
    return fin;
}
RFW_DI f3 BSDFEval(const ShadingData& sd, f3 N, f3 wo, f3 wi, float t, bool backfacing) { return BSDFEval(sd, N, wo, wi, t, backfacing, half_of(wi, wo)); }
// disney.glsl:197-263.  Returns whether the pdf is still to be evaluated (BSDFPdf of the sampled direction): the callers do that with the half
// vector they share with the evaluation
RFW_DI bool BSDFSampleDirection(const ShadingData& sd, f3 T, f3 B, f3 N, f3 wo, f3& wi, float& pdf, int& type, float r3, float r4)
{
    if (r3 < sd.transmission) {
        const float F = Fr_of(sd, dot(N, wo));
        if (r4 < F) {
            const float r1 = r3 / sd.transmission;
            const float r2 = r4 / F;
            const float cosThetaHalf = __builtin_sqrtf((1.0f - r2) / (1.0f + (sqr(sd.roughness) - 1.0f) * r2));
            const float sinThetaHalf = __builtin_sqrtf(gl_max(0.0f, 1.0f - sqr(cosThetaHalf)));
            float sinPhiHalf, cosPhiHalf;
            rfw_sincosf(r1 * RFW_TWOPI, &sinPhiHalf, &cosPhiHalf);
            f3 halfway = T * (sinThetaHalf * cosPhiHalf) + B * (sinThetaHalf * sinPhiHalf) + N * cosThetaHalf;
            if (dot(halfway, wo) <= 0.0f) halfway = halfway * -1.0f;
            type = BSDF_TYPE_REFLECTED;
            wi = gl_reflect(wo * -1.0f, halfway);
        } else {
            pdf = 0.0f;
            if (Refract(wo, N, sd.eta, wi)) {
                type = BSDF_TYPE_SPECULAR;
                pdf = (1.0f - F) * sd.transmission;
            }
            return false;
        }
    } else {
        const float r1 = (r3 - sd.transmission) / (1.0f - sd.transmission);
        // sin / cos of the azimuth 2 pi r1: the diffuse lobe (utils.glsl:55-70: TWOPI * r0) and the specular lobe (r1 * TWOPI) evaluate the
        // same product, so it is computed once in front of the branch — half the lanes of a wavefront take either side, and the ~80
        // instructions of the shared sin / cos would otherwise run twice per wavefront
        float sinPhiHalf, cosPhiHalf;
        rfw_sincosf(r1 * RFW_TWOPI, &sinPhiHalf, &cosPhiHalf);
        if (r4 < 0.5f) {
            const float r2 = r4 * 2.0f;
            f3 d;
            if (r2 < sd.subsurface) {
                const float r5 = r2 / sd.subsurface;
                const float term2 = __builtin_sqrtf(1.0f - r5 * r5); // DiffuseReflectionUniform(r1, r5)
                d = mk3(cosPhiHalf * term2, sinPhiHalf * term2, r5);
                type = BSDF_TYPE_TRANSMITTED;
                d.z *= -1.0f;
            } else {
                const float r5 = (r2 - sd.subsurface) / (1.0f - sd.subsurface);
                const float term2 = __builtin_sqrtf(1.0f - r5);      // DiffuseReflectionCosWeighted(r1, r5)
                d = mk3(cosPhiHalf * term2, sinPhiHalf * term2, __builtin_sqrtf(r5));
                type = BSDF_TYPE_REFLECTED;
            }
            wi = T * d.x + B * d.y + N * d.z;
        } else {
            const float r2 = (r4 - 0.5f) * 2.0f;
            const float cosThetaHalf = __builtin_sqrtf((1.0f - r2) / (1.0f + (sqr(sd.roughness) - 1.0f) * r2));
            const float sinThetaHalf = __builtin_sqrtf(gl_max(0.0f, 1.0f - sqr(cosThetaHalf)));
            f3 halfway = T * (sinThetaHalf * cosPhiHalf) + B * (sinThetaHalf * sinPhiHalf) + N * cosThetaHalf;
            if (dot(halfway, wo) <= 0.0f) halfway = halfway * -1.0f;
            wi = gl_reflect(wo * -1.0f, halfway);
            type = BSDF_TYPE_REFLECTED;
        }
    }
    return true;
}
RFW_DI void BSDFSample(const ShadingData& sd, f3 T, f3 B, f3 N, f3 wo, f3& wi, float& pdf, int& type, float r3, float r4)
{
    if (BSDFSampleDirection(sd, T, B, N, wo, wi, pdf, type, r3, r4)) pdf = BSDFPdf(sd, N, wo, wi);
}
// disney.glsl:265-270
// (`F_of_iN` = Fr_of(sd, dot(iN, wo)); k_shade has it from the sampling step, whose pdf is taken over the same normal)
RFW_DI f3 EvaluateBSDF(const ShadingData& sd, f3 iN, f3 wo, f3 wi, float& pdf, const float F_of_iN)
{
    const HalfVec hh = half_of(wi, wo);
    const f3 bsdf = BSDFEval(sd, iN, wo, wi, 0.0f, false, hh);
    pdf = BSDFPdf(sd, iN, wo, wi, hh, F_of_iN);
    return bsdf;
}
// disney.glsl:272-283
// (`F_of_N` = Fr_of(sd, dot(N, wo)), N the normal the direction is sampled and its pdf taken over)
RFW_DI f3 SampleBSDF(const ShadingData& sd, f3 iN, f3 N, f3 T, f3 B, f3 wo, float t, bool backfacing, float r3, float r4, f3& wi, float& pdf, const float F_of_N)
{
    int type = BSDF_TYPE_REFLECTED;
    const bool pdf_to_do = BSDFSampleDirection(sd, T, B, N, wo, wi, pdf, type, r3, r4);
    const HalfVec hh = half_of(wi, wo);
    if (pdf_to_do) pdf = BSDFPdf(sd, N, wo, wi, hh, F_of_N);
    return BSDFEval(sd, iN, wo, wi, t, backfacing, hh);
}

// ---- texture sampling (shade.comp:268-281) with the sampler of gpu-rt/src/lib.rs:1026-1038: repeat addressing, linear at LOD 0,
// nearest at LOD >= 1, f32 weights, byte * (1/255) decoding — the one meaning fixed for parity (oracle.cpp texture_sample)
struct f4 {
    float x, y, z, w;
};
RFW_DI f4 operator+(f4 a, f4 b) { return f4{a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
RFW_DI f4 operator*(f4 a, float s) { return f4{a.x * s, a.y * s, a.z * s, a.w * s}; }
RFW_DI f4 operator*(float s, f4 a) { return f4{s * a.x, s * a.y, s * a.z, s * a.w}; }
RFW_DI f4 mix4(f4 a, f4 b, float t) { return a * (1.0f - t) + b * t; }
RFW_DI f4 texel_at(const uint32_t* __restrict__ data, const TexDesc& t, uint32_t level, int32_t x, int32_t y)
{
    uint32_t w = t.w, h = t.h, off = t.offset;
    for (uint32_t l = 0; l < level; l++) {
        off += w * h;
        w >>= 1; h >>= 1;
    }
    int32_t xi = x % (int32_t)w, yi = y % (int32_t)h;
    if (xi < 0) xi += (int32_t)w;
    if (yi < 0) yi += (int32_t)h;
    const uint32_t p = data[off + (uint32_t)yi * w + (uint32_t)xi];
    const float c0 = (float)(p & 255u) * (1.0f / 255.0f), c1 = (float)((p >> 8) & 255u) * (1.0f / 255.0f),
                c2 = (float)((p >> 16) & 255u) * (1.0f / 255.0f), c3 = (float)(p >> 24) * (1.0f / 255.0f);
    return t.format == RFW_FORMAT_BGRA8 ? f4{c2, c1, c0, c3} : f4{c0, c1, c2, c3};
}
RFW_DI f4 texture_sample(const uint32_t* __restrict__ data, const TexDesc& t, float u, float v, float LOD)
{
    if (t.mips == 0 || t.w == 0 || t.h == 0) return f4{0.0f, 0.0f, 0.0f, 0.0f};
    int32_t level = f2i(LOD);
    if (level < 0) level = 0;
    if (level > (int32_t)t.mips - 1) level = (int32_t)t.mips - 1;
    const uint32_t w = t.w >> level, h = t.h >> level;
    if (level == 0) {
        const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
        const float x0 = __builtin_floorf(x), y0 = __builtin_floorf(y);
        const float fx = x - x0, fy = y - y0;
        const int32_t ix = f2i(x0), iy = f2i(y0);
        const f4 t00 = texel_at(data, t, 0, ix, iy), t10 = texel_at(data, t, 0, ix + 1, iy), t01 = texel_at(data, t, 0, ix, iy + 1),
                 t11 = texel_at(data, t, 0, ix + 1, iy + 1);
        return mix4(mix4(t00, t10, fx), mix4(t01, t11, fx), fy);
    }
    return texel_at(data, t, (uint32_t)level, f2i(__builtin_floorf(u * (float)w)), f2i(__builtin_floorf(v * (float)h)));
}
// shade.comp:273-281
RFW_DI f4 fetchTexelTrilinear(const uint32_t* __restrict__ data, const TexDesc& t, float lambda, float u, float v)
{
    const int32_t MIPLEVELCOUNT = (int32_t)t.mips;
    int32_t level0 = f2i(lambda);
    if (level0 > MIPLEVELCOUNT - 1) level0 = MIPLEVELCOUNT - 1;
    int32_t level1 = level0 + 1;
    if (level1 > MIPLEVELCOUNT - 1) level1 = MIPLEVELCOUNT - 1;
    const float f = lambda - __builtin_floorf(lambda);
    const f4 p0 = texture_sample(data, t, u, v, (float)level0);
    const f4 p1 = texture_sample(data, t, u, v, (float)level1);
    return (1.0f - f) * p0 + f * p1;
}

// ---- shade.comp:283-528 (uniform light pick: ISLIGHTS undefined) ----
struct LightView {
    const rfw_area_light* area;
    const rfw_point_light* point;
    const rfw_spot_light* spot;
    const rfw_directional_light* directional;
    int n_area, n_point, n_spot, n_directional;
};
RFW_DI f3 ld3(const rfw_vec3& v) { return mk3(v.x, v.y, v.z); }

// shade.comp:325-328
RFW_DI float CalculateLightPDF(f3 D, float t, float lightArea, f3 lightNormal) { return (t * t) / (-dot(D, lightNormal) * lightArea); }

// shade.comp:371-411.  The reference walks 16 levels of a 4-way triangle subdivision, one base-4 digit of r0's 32 bits per level, and
// returns the centroid of the last cell.  Every vertex on the way is a dyadic rational, so the walk is exact in float32 and any exact
// evaluation gives the same bits.  Closed form: a digit d != 0 moves the centroid half-way towards vertex d - 1 of the current cell, a
// digit 0 (the middle cell) leaves it and flips the cell's orientation for all later levels; with the cell shrinking by 2 per level,
//   3 * centroid = (1, 1) + sum_i sign_i * 2^-(i+1) * c[d_i],   c[1] = (2, -1), c[2] = (-1, 2), c[3] = (-1, -1), sign_i = (-1)^(zeros before i).
// The three digit classes as 16-bit masks (digit i at bit 15 - i) ARE the weighted sums; the sign is a prefix parity.  ~50 integer
// instructions without a branch instead of 16 rounds of a divergent 4-way switch (~320 vector + ~200 scalar instructions of k_shade's
// ~2400 per wavefront); tests/test_oracle_kat.py::test_random_barycentrics_closed_form proves the identity on the oracle's loop.
RFW_DI uint32_t bary_even_bits(uint32_t x) // bits 0, 2, 4, ..., 30 -> bits 0 ... 15
{
    x &= 0x55555555u;
    x = (x | (x >> 1)) & 0x33333333u;
    x = (x | (x >> 2)) & 0x0f0f0f0fu;
    x = (x | (x >> 4)) & 0x00ff00ffu;
    x = (x | (x >> 8)) & 0x0000ffffu;
    return x;
}
RFW_DI f3 RandomBarycentrics(float r0)
{
    const uint32_t uf = f2u(r0 * 4294967296.0f);
    const uint32_t L = bary_even_bits(uf), H = bary_even_bits(uf >> 1);
    const uint32_t M1 = ~H & L, M2 = H & ~L & 0xffffu, M3 = H & L, Z = ~(H | L) & 0xffffu;
    uint32_t t = Z; // bit p <- parity of the zero digits at bits >= p, i.e. at this and more significant digits
    t ^= t >> 1; t ^= t >> 2; t ^= t >> 4; t ^= t >> 8;
    const uint32_t N = (t >> 1) & 0xffffu; // digits whose cell is flipped: an odd number of zero digits BEFORE them
    const int32_t d1 = (int32_t)(M1 & ~N) - (int32_t)(M1 & N), d2 = (int32_t)(M2 & ~N) - (int32_t)(M2 & N), d3 = (int32_t)(M3 & ~N) - (int32_t)(M3 & N);
    const float sx = (float)(65536 + 2 * d1 - d2 - d3) * (1.0f / 65536.0f); // = A.x + B.x + C.x of the last cell, exactly
    const float sy = (float)(65536 - d1 + 2 * d2 - d3) * (1.0f / 65536.0f);
    const float rx = sx * 0.3333333f, ry = sy * 0.3333333f;
    return mk3(rx, ry, 1.0f - rx - ry);
}

// shade.comp:413-528
RFW_DI f3 RandomPointOnLight(const LightView& lv, float r0, f3 I, f3 N, float& pickProb, float& lightPdf, f3& lightColor, int& pickedLight)
{
    const int AREA = lv.n_area, POINT = lv.n_point, SPOT = lv.n_spot;
    const uint32_t lightCount = (uint32_t)(lv.n_area + lv.n_point + lv.n_spot + lv.n_directional);
    const f3 bary = RandomBarycentrics(r0);
    pickProb = 1.0f / (float)lightCount;
    int lightIdx = f2i(r0 * (float)lightCount);
    lightIdx = lightIdx < 0 ? 0 : (lightIdx > (int)lightCount - 1 ? (int)lightCount - 1 : lightIdx);
    pickedLight = lightIdx;
    if (lightIdx < AREA) {
        const rfw_area_light* al = lv.area + lightIdx;
        lightColor = ld3(al->radiance);
        const f3 LN = ld3(al->normal);
        const f3 P = bary.x * ld3(al->vertex0) + bary.y * ld3(al->vertex1) + bary.z * ld3(al->vertex2);
        f3 L = I - P;
        const float sqDist = dot(L, L);
        L = normalize(L);
        const float LNdotL = dot(L, LN);
        const float reciSolidAngle = sqDist / (al->energy * LNdotL);
        lightPdf = (LNdotL > 0.0f && dot(L, N) < 0.0f) ? (reciSolidAngle * (1.0f / al->area)) : 0.0f;
        return P;
    }
    if (lightIdx < (AREA + POINT)) {
        const rfw_point_light* pl = lv.point + (lightIdx - AREA);
        lightColor = ld3(pl->radiance);
        const f3 L = I - ld3(pl->position);
        const float sqDist = dot(L, L);
        lightPdf = dot(L, N) < 0.0f ? (sqDist / pl->energy) : 0.0f;
        return ld3(pl->position);
    }
    if (lightIdx < (AREA + POINT + SPOT)) {
        const rfw_spot_light* sl = lv.spot + (lightIdx - (AREA + POINT));
        f3 L = I - ld3(sl->position);
        const float sqDist = dot(L, L);
        L = normalize(L);
        const float d = gl_max(0.0f, dot(L, ld3(sl->direction)) - sl->cos_outer) / (sl->cos_inner - sl->cos_outer);
        const float LNdotL = gl_min(1.0f, d);
        lightPdf = (LNdotL > 0.0f && dot(L, N) < 0.0f) ? (sqDist / (LNdotL * sl->energy)) : 0.0f;
        lightColor = ld3(sl->radiance);
        return ld3(sl->position);
    }
    const rfw_directional_light* dl = lv.directional + (lightIdx - (AREA + POINT + SPOT));
    const f3 L = ld3(dl->direction);
    lightColor = ld3(dl->radiance);
    const float NdotL = dot(L, N);
    lightPdf = NdotL < 0.0f ? (1.0f * (1.0f / dl->energy)) : 0.0f;
    return I - 1000.0f * L;
}

} // namespace rfwhip
