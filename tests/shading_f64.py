"""An INDEPENDENT float64 formulation of the shading functions of the path — the Disney-style BSDF (evaluation, pdf, importance
sampling) and the light sampling of backends/gpu-rt/shaders/disney.glsl:89-285 and shade.comp:325-328,371-528 — written from the
formulas (GGX / Smith / Schlick / dielectric Fresnel, solid-angle pdfs), in scalar Python + numpy float64, NOT from oracle/oracle.cpp
or csrc/shade_device.h.  The oracle and the device functions are transliterations of the same GLSL text in float32; this file is what
a slip in either transliteration (a wrong term, constant, sign or branch) is caught against (tests/test_shading_kat.py)."""
import math

import numpy as np

PI = float(np.float32(3.14159265359))  # the GLSL constant as the float32 both implementations hold


def unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / math.sqrt(float(v @ v))


# ---- building blocks, by formula
def fresnel_dielectric(cos_i, eio):
    """Unpolarised Fresnel reflectance of a dielectric boundary; eio = eta_i / eta_o; total internal reflection -> 1."""
    sin2_t = eio * eio * (1.0 - cos_i * cos_i)
    if sin2_t > 1.0:
        return 1.0
    cos_t = math.sqrt(1.0 - sin2_t)
    n = 1.0 / eio
    rs = (cos_i - n * cos_t) / (cos_i + n * cos_t)
    rp = (cos_t - n * cos_i) / (cos_t + n * cos_i)
    return 0.5 * (rs * rs + rp * rp)


def ggx_d(cos_h, alpha):
    """GGX / Trowbridge-Reitz normal distribution (Burley's GTR with gamma = 2)."""
    a2 = alpha * alpha
    return a2 / (PI * ((a2 - 1.0) * cos_h * cos_h + 1.0) ** 2)


def gtr1_d(cos_h, alpha):
    """Burley's GTR with gamma = 1 (the clearcoat lobe)."""
    if alpha >= 1.0:
        return 1.0 / PI
    a2 = alpha * alpha
    return (a2 - 1.0) / (PI * math.log(a2) * (1.0 + (a2 - 1.0) * cos_h * cos_h))


def smith_g1(cos_v, alpha):
    """Separable Smith shadowing term in the 1 / (c + sqrt(a^2 + c^2 - a^2 c^2)) form (the 2 c factors of G cancel against 4 c c')."""
    a2 = alpha * alpha
    return 1.0 / (cos_v + math.sqrt(a2 + cos_v * cos_v * (1.0 - a2)))


def schlick_weight(u):
    m = min(max(1.0 - u, 0.0), 1.0)
    return m ** 5


def lerp(a, b, t):
    return a * (1.0 - t) + b * t


# ---- material: the 16 unorm8 parameters of a DeviceMaterial (crates/rfw-scene/src/material/list.rs:755-783)
PARAMS = ["metallic", "subsurface", "specular_f", "roughness", "specular_tint", "anisotropic", "sheen", "sheen_tint",
          "clearcoat", "clearcoat_gloss", "transmission", "eta", "custom0", "custom1", "custom2", "custom3"]


def material_from_bytes(color, absorption, specular, param_bytes):
    m = {"color": np.asarray(color, np.float64), "absorption": np.asarray(absorption, np.float64), "specular": np.asarray(specular, np.float64)}
    for name, b in zip(PARAMS, param_bytes):
        m[name] = float(b) / 255.0
    m["roughness"] = max(0.01, m["roughness"])  # structs.glsl:231
    return m


def pack_material(color, absorption, specular, param_bytes):
    """The same material as the 24 words of a rfw_device_material (include/rfw_pod.h): colour, absorption, specular as vec4s, then
    parameters[4] (low byte first), flags, five map ids, two dummies."""
    w = np.zeros(24, np.uint32)
    w[0:3] = np.asarray(color, np.float32).view(np.uint32)
    w[4:7] = np.asarray(absorption, np.float32).view(np.uint32)
    w[8:11] = np.asarray(specular, np.float32).view(np.uint32)
    pb = [int(b) for b in param_bytes]
    for k in range(4):
        w[12 + k] = pb[4 * k] | (pb[4 * k + 1] << 8) | (pb[4 * k + 2] << 16) | (pb[4 * k + 3] << 24)
    w[16] = 0
    w[17:22] = np.uint32(0xFFFFFFFF)  # no maps
    return w.view(np.float32)


# ---- disney.glsl:89-108
def bsdf_pdf(m, N, wo, wi):
    cos_i = float(wi @ N)
    tr, ss = m["transmission"], m["subsurface"]
    if cos_i <= 0.0:
        return lerp(0.5 * ss / (2.0 * PI), 0.0, tr)           # uniform lower hemisphere, chosen with probability subsurface / 2
    h = wi + wo
    l2 = float(h @ h)
    h = h / math.sqrt(l2) if l2 > 0.0 else np.zeros(3)
    cos_h = abs(float(h @ N))
    pdf_spec = 0.25 * ggx_d(cos_h, m["roughness"]) * cos_h / max(1e-6, float(wi @ h))   # D cos_h d_omega_h / (4 wi.h)
    pdf_diff = abs(cos_i) / PI * (1.0 - ss)
    F = fresnel_dielectric(float(N @ wo), m["eta"])
    return lerp(0.5 * (pdf_diff + pdf_spec), pdf_spec * F, tr)


# ---- disney.glsl:110-195
def bsdf_eval(m, N, wo, wi, t, backfacing):
    c = m["color"]
    cos_l, cos_v = float(N @ wi), float(N @ wo)
    h = unit(wi + wo)
    cos_h, l_h = float(N @ h), float(wi @ h)
    lum = 0.3 * c[0] + 0.6 * c[1] + 0.1 * c[2]
    tint = c / lum if lum > 0.0 else np.ones(3)
    spec0 = lerp(m["specular"] * 0.08 * lerp(np.ones(3), tint, m["specular_tint"]), c, m["metallic"])
    a = m["roughness"]
    tr = m["transmission"]
    bsdf = np.zeros(3)
    brdf = np.zeros(3)
    if tr > 0.0:
        if cos_l <= 0.0:
            F = fresnel_dielectric(cos_v, m["eta"])
            bsdf = np.full(3, (1.0 - F) / abs(cos_l) * (1.0 - m["metallic"]) * tr)
        else:
            Fs = lerp(spec0, np.ones(3), fresnel_dielectric(l_h, m["eta"]))
            bsdf = smith_g1(cos_v, a) * smith_g1(cos_l, a) * ggx_d(cos_h, a) * Fs
    if tr < 1.0:
        if cos_l <= 0.0:
            if m["subsurface"] > 0.0:
                Fd = (1.0 - 0.5 * schlick_weight(abs(cos_l))) * (1.0 - 0.5 * schlick_weight(cos_v))
                brdf = np.sqrt(c) / PI * m["subsurface"] * Fd * (1.0 - m["metallic"])
        else:
            FH = schlick_weight(l_h)
            Fs = lerp(spec0, np.ones(3), FH)
            Gs = smith_g1(cos_v, a) * smith_g1(cos_l, a)
            Fd90 = 0.5 + 2.0 * l_h * l_h * a
            Fd = lerp(1.0, Fd90, schlick_weight(cos_l)) * lerp(1.0, Fd90, schlick_weight(cos_v))
            coat = m["clearcoat"] * smith_g1(cos_l, 0.25) * smith_g1(cos_v, 0.25) * lerp(0.04, 1.0, FH) * gtr1_d(cos_h, lerp(0.1, 0.001, m["clearcoat_gloss"]))
            brdf = Fd * c / PI * (1.0 - m["metallic"]) * (1.0 - m["subsurface"]) + Gs * ggx_d(cos_h, a) * Fs + coat
    f = lerp(brdf, bsdf, tr)
    if backfacing:
        f = f * np.exp(-m["absorption"] * t)                 # Beer-Lambert over the distance travelled inside
    return f


# ---- disney.glsl:197-263
def ggx_half_vector(T, B, N, wo, alpha, u_phi, u_theta):
    """Half vector drawn with density D(h) cos(theta_h) (GGX with alpha), flipped into wo's hemisphere."""
    cos_t = math.sqrt((1.0 - u_theta) / (1.0 + (alpha * alpha - 1.0) * u_theta))
    sin_t = math.sqrt(max(0.0, 1.0 - cos_t * cos_t))
    phi = 2.0 * PI * u_phi
    h = T * (sin_t * math.cos(phi)) + B * (sin_t * math.sin(phi)) + N * cos_t
    if float(h @ wo) <= 0.0:
        h = -h
    return h


def bsdf_sample(m, T, B, N, wo, r3, r4):
    """-> (wi, pdf, type); type 0 reflected, 1 transmitted (diffuse, below the surface), 2 specular refraction."""
    tr, ss = m["transmission"], m["subsurface"]
    if r3 < tr:
        F = fresnel_dielectric(float(N @ wo), m["eta"])
        if r4 < F:
            h = ggx_half_vector(T, B, N, wo, m["roughness"], r3 / tr, r4 / F)
            wi = 2.0 * float(wo @ h) * h - wo                                         # mirror wo about h
            return wi, bsdf_pdf(m, N, wo, wi), 0
        cos_i = float(N @ wo)                                                            # refraction by Snell's law
        sin2_t = m["eta"] ** 2 * max(0.0, 1.0 - cos_i * cos_i)
        if sin2_t >= 1.0:
            return np.zeros(3), 0.0, 0
        wi = -m["eta"] * wo + (m["eta"] * cos_i - math.sqrt(1.0 - sin2_t)) * N
        return wi, (1.0 - F) * tr, 2
    u = (r3 - tr) / (1.0 - tr)
    if r4 < 0.5:
        v = 2.0 * r4
        phi = 2.0 * PI * u
        if v < ss:                                                                        # uniform over the LOWER hemisphere
            z = v / ss
            s = math.sqrt(1.0 - z * z)
            d, kind = np.array([math.cos(phi) * s, math.sin(phi) * s, -z]), 1
        else:                                                                             # cosine-weighted upper hemisphere
            z2 = (v - ss) / (1.0 - ss)
            s = math.sqrt(1.0 - z2)
            d, kind = np.array([math.cos(phi) * s, math.sin(phi) * s, math.sqrt(z2)]), 0
        wi = T * d[0] + B * d[1] + N * d[2]
        return wi, bsdf_pdf(m, N, wo, wi), kind
    h = ggx_half_vector(T, B, N, wo, m["roughness"], u, 2.0 * (r4 - 0.5))
    wi = 2.0 * float(wo @ h) * h - wo
    return wi, bsdf_pdf(m, N, wo, wi), 0


# ---- shade.comp:325-328: area pdf 1 / A converted to solid angle as seen along D at distance t
def light_pdf_solid_angle(D, t, area, light_normal):
    return t * t / (-float(D @ light_normal) * area)


# ---- shade.comp:371-411: r0's 32 bits as 16 base-4 digits walking a 4-way triangle subdivision; the centroid of the last cell
def random_barycentrics(r0):
    bits = min(int(float(np.float32(r0)) * 4294967296.0), 0xFFFFFFFF)
    A, B, C = np.array([1.0, 0.0]), np.array([0.0, 1.0]), np.array([0.0, 0.0])
    for i in range(16):
        d = (bits >> (2 * (15 - i))) & 3
        if d == 0:
            A, B, C = (B + C) / 2, (A + C) / 2, (A + B) / 2
        elif d == 1:
            A, B, C = A, (A + B) / 2, (A + C) / 2
        elif d == 2:
            A, B, C = (B + A) / 2, B, (B + C) / 2
        else:
            A, B, C = (C + A) / 2, (C + B) / 2, C
    r = (A + B + C) * float(np.float32(0.3333333))
    return np.array([r[0], r[1], 1.0 - r[0] - r[1]])


# ---- shade.comp:413-528 with ISLIGHTS undefined: uniform pick over area, point, spot, directional lights in that order
def random_point_on_light(lights, r0, I, N):
    """lights = {"area": [...], "point": [...], "spot": [...], "directional": [...]} of dicts -> (P, pick_prob, pdf, colour, index)."""
    n_a, n_p, n_s, n_d = (len(lights[k]) for k in ("area", "point", "spot", "directional"))
    count = n_a + n_p + n_s + n_d
    idx = min(max(int(float(np.float32(r0) * np.float32(count))), 0), count - 1)
    pick = 1.0 / count
    if idx < n_a:
        al = lights["area"][idx]
        b = random_barycentrics(r0)
        P = b[0] * al["vertex0"] + b[1] * al["vertex1"] + b[2] * al["vertex2"]
        L = I - P
        d2 = float(L @ L)
        Ln = L / math.sqrt(d2)
        cos_light = float(Ln @ al["normal"])
        ok = cos_light > 0.0 and float(Ln @ N) < 0.0
        return P, pick, (d2 / (al["energy"] * cos_light) / al["area"]) if ok else 0.0, al["radiance"], idx
    idx2 = idx - n_a
    if idx2 < n_p:
        pl = lights["point"][idx2]
        L = I - pl["position"]
        return pl["position"], pick, (float(L @ L) / pl["energy"]) if float(L @ N) < 0.0 else 0.0, pl["radiance"], idx
    idx2 -= n_p
    if idx2 < n_s:
        sl = lights["spot"][idx2]
        L = I - sl["position"]
        d2 = float(L @ L)
        Ln = L / math.sqrt(d2)
        falloff = min(1.0, max(0.0, float(Ln @ sl["direction"]) - sl["cos_outer"]) / (sl["cos_inner"] - sl["cos_outer"]))
        ok = falloff > 0.0 and float(Ln @ N) < 0.0
        return sl["position"], pick, (d2 / (falloff * sl["energy"])) if ok else 0.0, sl["radiance"], idx
    dl = lights["directional"][idx2 - n_s]
    return I - 1000.0 * dl["direction"], pick, (1.0 / dl["energy"]) if float(dl["direction"] @ N) < 0.0 else 0.0, dl["radiance"], idx
