/*
 * oracle/glsl_ref_wrap.cpp — TEST INFRASTRUCTURE.  C entry points of the build of the reference's own GLSL text as C++
 * (oracle/make_glsl_ref.py writes _ref/glsl_ref_gen.inc from the shader files where they lie; oracle/glsl_shim.h supplies the types).
 * Same call shapes as orc_eval_shading / orc_glsl_twin of oracle.cpp, so tests/test_glsl_differential.py can hold one against the other.
 */
#include <cstring>

#include "glsl_shim.h"

namespace glslref {
#include "glsl_ref_gen.inc"
} // namespace glslref

using namespace glslref;

static inline float ubits(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
static inline uint32_t fbits_(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

// per case 48 input floats (the layout of rfw_hip_debug_eval_shading: [0,24) a 96-byte material, N, wo, wi, T, B, t, backfacing, r3, r4, area) and 12 outputs
extern "C" __attribute__((visibility("default"))) int glslref_eval_shading(int op, uint64_t n, const float* in, float* out)
{
    for (uint64_t i = 0; i < n; i++) {
        const float* q = in + 48 * i;
        float* r = out + 12 * i;
        for (int k = 0; k < 12; k++) r[k] = 0.0f;
        uvec4 par;
        std::memcpy(&par, q + 12, 16);
        const ShadingData sd = extractParameters(vec3(q[0], q[1], q[2]), vec3(q[4], q[5], q[6]), vec3(q[8], q[9], q[10]), par);
        const vec3 N(q[24], q[25], q[26]), wo(q[27], q[28], q[29]), wi(q[30], q[31], q[32]), T(q[33], q[34], q[35]), B(q[36], q[37], q[38]);
        switch (op) {
        case 0: { const vec3 f = BSDFEval(sd, N, wo, wi, q[39], q[40] != 0.0f); r[0] = f.x; r[1] = f.y; r[2] = f.z; break; }
        case 1: r[0] = BSDFPdf(sd, N, wo, wi); break;
        case 2: {
            vec3 w(0.0f); float pdf = 0.0f; int type = BSDF_TYPE_REFLECTED;
            BSDFSample(sd, T, B, N, wo, w, pdf, type, q[39], q[40] != 0.0f, q[41], q[42]);
            r[0] = w.x; r[1] = w.y; r[2] = w.z; r[3] = pdf; r[4] = (float)type; break;
        }
        default: return -1;
        }
    }
    return 0;
}

// per case 32 input floats and 24 output floats; see oracle.cpp, orc_glsl_twin, for the layouts (the two are written side by side)
extern "C" __attribute__((visibility("default"))) int glslref_twin(int op, uint64_t n, const float* in, float* out)
{
    for (uint64_t i = 0; i < n; i++) {
        const float* q = in + 32 * i;
        float* r = out + 24 * i;
        for (int k = 0; k < 24; k++) r[k] = 0.0f;
        switch (op) {
        case 10: case 11: { // intersect / intersect_occludes: v0 v1 v2 gn O D t_min t
            RTTriangle tr;
            std::memset(&tr, 0, sizeof(tr));
            tr.v0 = vec3(q[0], q[1], q[2]); tr.v1 = vec3(q[3], q[4], q[5]); tr.v2 = vec3(q[6], q[7], q[8]); tr.gn = vec3(q[9], q[10], q[11]);
            const vec3 O(q[12], q[13], q[14]), D(q[15], q[16], q[17]);
            if (op == 10) {
                float t = q[19]; vec2 uv(0.0f, 0.0f);
                const bool h = intersect(tr, O, D, q[18], t, uv);
                r[0] = h ? 1.0f : 0.0f; r[1] = t; r[2] = uv.x; r[3] = uv.y;
            } else {
                r[0] = intersect_occludes(tr, O, D, q[18], q[19]) ? 1.0f : 0.0f;
            }
            break;
        }
        case 12: { // intersect_mnode: min_x[4] max_x[4] min_y[4] max_y[4] min_z[4] max_z[4] origin dir_inverse t
            MBVHNode nd;
            std::memset(&nd, 0, sizeof(nd));
            nd.min_x = vec4(q[0], q[1], q[2], q[3]); nd.max_x = vec4(q[4], q[5], q[6], q[7]);
            nd.min_y = vec4(q[8], q[9], q[10], q[11]); nd.max_y = vec4(q[12], q[13], q[14], q[15]);
            nd.min_z = vec4(q[16], q[17], q[18], q[19]); nd.max_z = vec4(q[20], q[21], q[22], q[23]);
            vec4 tmin(0.0f); bvec4 res;
            const bool any_ = intersect_mnode(nd, vec3(q[24], q[25], q[26]), vec3(q[27], q[28], q[29]), q[30], tmin, res);
            r[0] = any_ ? 1.0f : 0.0f;
            for (int k = 0; k < 4; k++) { r[1 + k] = res[k] ? 1.0f : 0.0f; r[5 + k] = any_ ? tmin[k] : 0.0f; }
            break;
        }
        case 13: { const vec3 p = safe_origin(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), vec3(q[6], q[7], q[8]), q[9]); r[0] = p.x; r[1] = p.y; r[2] = p.z; break; }
        case 14: r[0] = ubits(PackNormal(vec3(q[0], q[1], q[2]))); break;
        case 16: {
            const vec3 a = DiffuseReflectionCosWeighted(q[0], q[1]), b = DiffuseReflectionUniform(q[0], q[1]);
            r[0] = a.x; r[1] = a.y; r[2] = a.z; r[3] = b.x; r[4] = b.y; r[5] = b.z; break;
        }
        case 17: { vec3 c(q[0], q[1], q[2]); CLAMPINTENSITY(c, q[3]); r[0] = c.x; r[1] = c.y; r[2] = c.z; break; }
        case 18: {
            uint s = fbits_(q[0]);
            r[0] = ubits(wang_hash(s));
            r[1] = ubits(randi(s)); r[2] = randf(s); r[3] = ubits(s);
            break;
        }
        case 20: {
            r[0] = Fr(q[0], q[1]); r[1] = SchlickFresnel(q[0]); r[2] = GTR1(q[0], q[1]); r[3] = GTR2(q[0], q[1]); r[4] = SmithGGX(q[0], q[1]);
            vec3 wt(0.0f);
            const bool ok = Refract(vec3(q[2], q[3], q[4]), vec3(q[5], q[6], q[7]), q[8], wt);
            r[5] = ok ? 1.0f : 0.0f; r[6] = wt.x; r[7] = wt.y; r[8] = wt.z;
            break;
        }
        case 21: { // extractParameters: colour absorption specular (3 each) + 4 parameter words
            uvec4 par;
            std::memcpy(&par, q + 9, 16);
            const ShadingData d = extractParameters(vec3(q[0], q[1], q[2]), vec3(q[3], q[4], q[5]), vec3(q[6], q[7], q[8]), par);
            const float v[] = {d.metallic, d.subsurface, d.specular_f, d.roughness, d.specular_tint, d.anisotropic, d.sheen, d.sheen_tint, d.clearcoat, d.clearcoat_gloss,
                               d.transmission, d.eta, d.custom0, d.custom1, d.custom2, d.custom3};
            for (int k = 0; k < 16; k++) r[k] = v[k];
            break;
        }
        default: return -1;
        }
    }
    return 0;
}
