"""Known answers for the shading block — BSDFEval, BSDFPdf, BSDFSample, CalculateLightPDF, RandomPointOnLight — from an INDEPENDENT
float64 formulation (tests/shading_f64.py, written from the formulas of disney.glsl:89-285 / shade.comp:325-528, not from the oracle's
text).  The oracle's float32 transliteration is checked here on the CPU; the device functions of k_shade are checked against the same
tables on the GPU (rfw_hip_debug_eval_shading).  Plus two properties: the sampling pdf integrates to the probability mass the sampler
can reach, and sampled directions follow it; the BRDF's directional albedo stays bounded (white furnace)."""
import math

import numpy as np
import pytest

import shading_f64 as ref
from conftest import has_gpu

N_CASES = 400
RTOL = 2e-4   # float32 evaluation of ~50-operation expressions against float64; a transliteration slip is O(1)


def frame(rng):
    n = ref.unit(rng.normal(size=3))
    t = ref.unit(np.cross(n, rng.normal(size=3)))
    return n, t, np.cross(n, t)


def materials(rng, k):
    """param bytes with the interesting corners: transmission 0 / 1, subsurface 0, metallic, clearcoat, rough and smooth"""
    out = []
    for i in range(k):
        pb = rng.integers(0, 256, 16)
        pb[11] = rng.integers(120, 256)              # eta: away from 0 (1 / eta is evaluated)
        pb[3] = rng.integers(8, 256)                 # roughness: a floor of 0.01 applies below 3
        if i % 4 == 0:
            pb[10] = 0                               # no transmission
        if i % 7 == 0:
            pb[10] = 255                             # pure transmission
        if i % 3 == 0:
            pb[1] = 0                                # no subsurface
        if i % 5 == 0:
            pb[8] = 0                                # no clearcoat
        color = rng.uniform(0.05, 0.95, 3)
        out.append((color, rng.uniform(0.0, 0.6, 3), rng.uniform(0.2, 1.0, 3), pb))
    return out


def cases(seed, n=N_CASES):
    rng = np.random.default_rng(seed)
    rows, meta = [], []
    for color, absorption, specular, pb in materials(rng, n):
        N, T, B = frame(rng)
        while True:
            wo = ref.unit(rng.normal(size=3))
            if wo @ N > 0.08:
                break
        while True:
            wi = ref.unit(rng.normal(size=3))
            if abs(wi @ N) > 0.05 and np.linalg.norm(wi + wo) > 0.2:
                break
        t = rng.uniform(0.1, 4.0)
        back = float(rng.integers(0, 2))
        r3, r4 = rng.uniform(0.001, 0.999, 2)
        row = np.zeros(48, np.float32)
        row[0:24] = ref.pack_material(color, absorption, specular, pb)
        row[24:27], row[27:30], row[30:33], row[33:36], row[36:39] = N, wo, wi, T, B
        row[39], row[40], row[41], row[42], row[43] = t, back, r3, r4, rng.uniform(0.2, 3.0)
        rows.append(row)
        meta.append(ref.material_from_bytes(np.float32(color).astype(np.float64), np.float32(absorption).astype(np.float64),
                                            np.float32(specular).astype(np.float64), pb))
    return np.stack(rows), meta


def vec(row, a):
    return row[a:a + 3].astype(np.float64)


def close(got, want, what, rtol=RTOL, atol=1e-6):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want) - (atol + rtol * np.abs(want))
    assert (err <= 0).all(), f"{what}: got {got}, want {want}"


def check_tables(evaluate):
    """`evaluate(op, rows) -> (n, 12)`: the implementation under test (oracle binding or device)."""
    rows, mats = cases(11)
    # ---- BSDFPdf and BSDFEval on arbitrary direction pairs (both hemispheres)
    pdf, ev = evaluate(1, rows), evaluate(0, rows)
    for k, (row, m) in enumerate(zip(rows, mats)):
        N, wo, wi = vec(row, 24), vec(row, 27), vec(row, 30)
        close(pdf[k, 0], ref.bsdf_pdf(m, N, wo, wi), f"BSDFPdf case {k}")
        close(ev[k, :3], ref.bsdf_eval(m, N, wo, wi, float(row[39]), row[40] != 0), f"BSDFEval case {k}")
    # ---- BSDFSample: direction, pdf and lobe for given random numbers
    smp = evaluate(2, rows)
    checked = 0
    for k, (row, m) in enumerate(zip(rows, mats)):
        r3, r4 = float(row[41]), float(row[42])
        N, wo = vec(row, 24), vec(row, 27)
        tr = m["transmission"]
        F = ref.fresnel_dielectric(float(N @ wo), m["eta"])
        # a random number within float32 rounding of a branch threshold may take the other branch: skip those
        edges = [abs(r3 - tr), abs(r4 - 0.5)] + ([abs(r4 - F)] if r3 < tr else [abs(2 * r4 - m["subsurface"])] if r4 < 0.5 else [])
        if min(edges) < 1e-4:
            continue
        wi, p, kind = ref.bsdf_sample(m, vec(row, 33), vec(row, 36), N, wo, r3, r4)
        assert int(smp[k, 4]) == kind, k
        close(smp[k, :3], wi, f"BSDFSample wi case {k}", rtol=5e-4, atol=2e-5)
        if abs(float(wi @ N)) > 0.02:                # the pdf switches formula at the horizon
            close(smp[k, 3], p, f"BSDFSample pdf case {k}", rtol=3e-3, atol=1e-5)   # pdf(wi) is steep in wi for smooth lobes
        checked += 1
    assert checked > 0.9 * len(rows)
    # ---- CalculateLightPDF
    lp = evaluate(3, rows)
    for k, row in enumerate(rows):
        close(lp[k, 0], ref.light_pdf_solid_angle(vec(row, 27), float(row[39]), float(row[43]), vec(row, 24)), f"CalculateLightPDF {k}")


def make_lights(rng):
    from rfw_rs_amd import pod
    L = {"area": [], "point": [], "spot": [], "directional": []}
    P = {"area": [], "point": [], "spot": [], "directional": []}
    f32 = lambda v: np.asarray(v, np.float32)
    v3 = lambda v: pod.Vec3(*[float(x) for x in v])
    for _ in range(3):
        v0, v1, v2 = (f32(rng.uniform(-2, 2, 3) + np.array([0, 4, 0])) for _ in range(3))
        n = f32(ref.unit(np.cross(v1 - v0, v2 - v0).astype(np.float64)))
        area = np.float32(0.5 * np.linalg.norm(np.cross((v1 - v0).astype(np.float64), (v2 - v0).astype(np.float64))))
        rad = f32(rng.uniform(2, 12, 3))
        energy = np.float32(rng.uniform(1, 9))
        a = pod.AreaLight()
        a.position, a.energy, a.normal, a.area = v3((v0 + v1 + v2) / 3), float(energy), v3(n), float(area)
        a.vertex0, a.vertex1, a.vertex2, a.radiance = v3(v0), v3(v1), v3(v2), v3(rad)
        P["area"].append(a)
        L["area"].append({"vertex0": v0.astype(np.float64), "vertex1": v1.astype(np.float64), "vertex2": v2.astype(np.float64),
                          "normal": n.astype(np.float64), "area": float(area), "energy": float(energy), "radiance": rad.astype(np.float64)})
    for _ in range(2):
        pos, rad, energy = f32(rng.uniform(-3, 3, 3)), f32(rng.uniform(1, 6, 3)), np.float32(rng.uniform(1, 5))
        p = pod.PointLight()
        p.position, p.energy, p.radiance = v3(pos), float(energy), v3(rad)
        P["point"].append(p)
        L["point"].append({"position": pos.astype(np.float64), "energy": float(energy), "radiance": rad.astype(np.float64)})
    for _ in range(2):
        pos, rad, energy = f32(rng.uniform(-3, 3, 3) + np.array([0, 3, 0])), f32(rng.uniform(1, 6, 3)), np.float32(rng.uniform(1, 5))
        d = f32(ref.unit(rng.normal(size=3) + np.array([0, -2.0, 0])))
        ci, co = np.float32(0.95), np.float32(0.6)
        s = pod.SpotLight()
        s.position, s.cos_inner, s.radiance, s.cos_outer, s.direction, s.energy = v3(pos), float(ci), v3(rad), float(co), v3(d), float(energy)
        P["spot"].append(s)
        L["spot"].append({"position": pos.astype(np.float64), "cos_inner": float(ci), "cos_outer": float(co), "direction": d.astype(np.float64),
                          "energy": float(energy), "radiance": rad.astype(np.float64)})
    for _ in range(1):
        d, rad, energy = f32(ref.unit(np.array([0.3, -1.0, 0.2]))), f32(rng.uniform(1, 6, 3)), np.float32(rng.uniform(1, 5))
        q = pod.DirectionalLight()
        q.direction, q.energy, q.radiance = v3(d), float(energy), v3(rad)
        P["directional"].append(q)
        L["directional"].append({"direction": d.astype(np.float64), "energy": float(energy), "radiance": rad.astype(np.float64)})
    return L, P


def check_light_sampling(backend, evaluate):
    rng = np.random.default_rng(5)
    L, P = make_lights(rng)
    backend.set_area_lights(P["area"]); backend.set_point_lights(P["point"])
    backend.set_spot_lights(P["spot"]); backend.set_directional_lights(P["directional"])
    backend.synchronize()
    rows = np.zeros((N_CASES, 48), np.float32)
    for row in rows:
        row[24:27] = ref.unit(rng.normal(size=3) + np.array([0, 1.5, 0]))   # normals mostly facing the lights above
        row[27:30] = rng.uniform(-2, 2, 3)
        row[41] = rng.uniform(0.0005, 0.9995)
    out = evaluate(4, rows)
    seen = set()
    lit = 0
    for k, row in enumerate(rows):
        r0 = float(row[41])
        if min(abs(r0 * 8 - j) for j in range(9)) < 1e-3:
            continue                                  # the light index flips within float32 rounding of k / 8
        Pp, pick, pdf, col, idx = ref.random_point_on_light(L, r0, vec(row, 27), vec(row, 24))
        assert int(out[k, 8]) == idx
        seen.add(idx)
        close(out[k, :3], Pp, f"light point {k}", rtol=1e-4, atol=1e-4)
        close(out[k, 3], pick, f"pickProb {k}")
        close(out[k, 5:8], col, f"light colour {k}")
        Ldir = vec(row, 27) - Pp
        if abs(float(ref.unit(Ldir) @ vec(row, 24))) > 1e-3 and (idx >= 3 or abs(float(ref.unit(Ldir) @ L["area"][idx]["normal"])) > 1e-3):
            close(out[k, 4], pdf, f"lightPdf {k}", rtol=1e-3)   # sides of the horizon tests agree away from grazing
            lit += pdf > 0
    assert seen == set(range(8)) and lit > 50


class _OracleSetters:
    """the Backend-trait light setters on the oracle binding (arrays of the same PODs)"""
    def __init__(self, orc):
        self.o = orc

    def _set(self, name, ctype, items):
        import ctypes as C
        arr = (ctype * len(items))(*items)
        fn = getattr(self.o._l, name)
        fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        assert fn(self.o._h, arr, len(items), None) == 0

    def set_area_lights(self, l):
        from rfw_rs_amd import pod
        self._set("orc_set_area_lights", pod.AreaLight, l)

    def set_point_lights(self, l):
        from rfw_rs_amd import pod
        self._set("orc_set_point_lights", pod.PointLight, l)

    def set_spot_lights(self, l):
        from rfw_rs_amd import pod
        self._set("orc_set_spot_lights", pod.SpotLight, l)

    def set_directional_lights(self, l):
        from rfw_rs_amd import pod
        self._set("orc_set_directional_lights", pod.DirectionalLight, l)

    def synchronize(self):
        self.o._l.orc_synchronize(self.o._h)


def test_oracle_shading_functions_against_float64_formulas():
    from oracle.bindings import Oracle
    o = Oracle(8, 8)
    check_tables(o.eval_shading)


def test_oracle_light_sampling_against_float64_formulas():
    from oracle.bindings import Oracle
    o = Oracle(8, 8)
    check_light_sampling(_OracleSetters(o), o.eval_shading)


def quadrature_dirs(n_theta=720, n_phi=180):
    """midpoint rule on the sphere in (theta, phi) with weights sin(theta) dtheta dphi: fine cells around the pole N, where the
    narrow lobes of smooth materials live"""
    th = (np.arange(n_theta) + 0.5) / n_theta * math.pi
    phi = (np.arange(n_phi) + 0.5) / n_phi * 2.0 * math.pi
    tt, pp = np.meshgrid(th, phi, indexing="ij")
    d = np.stack([np.sin(tt) * np.cos(pp), np.sin(tt) * np.sin(pp), np.cos(tt)], axis=-1).reshape(-1, 3)
    w = (np.sin(tt) * (math.pi / n_theta) * (2.0 * math.pi / n_phi)).reshape(-1)
    return d, w


FURNACE = [  # (name, colour, params: metallic, subsurface, roughness, clearcoat, transmission), eta = 1 / 1.5 where it matters
    ("plastic", (0.8, 0.8, 0.8), dict(roughness=128)),
    ("rough", (0.9, 0.9, 0.9), dict(roughness=255)),
    ("metal", (0.95, 0.9, 0.8), dict(metallic=255, roughness=77)),
    ("coated", (0.7, 0.2, 0.2), dict(roughness=102, clearcoat=255, clearcoat_gloss=200)),
    ("waxy", (0.8, 0.7, 0.6), dict(subsurface=128, roughness=150)),
]


def property_rows(params, wo_cos, dirs):
    pb = np.zeros(16, np.int64)
    pb[2], pb[11] = 128, 170
    for k, v in params.items():
        pb[ref.PARAMS.index(k)] = v
    return pb


def check_properties(evaluate):
    dirs, dw = quadrature_dirs()
    N, T, B = np.array([0.0, 0.0, 1.0]), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0])
    rng = np.random.default_rng(3)
    for name, color, params in FURNACE:
        pb = property_rows(params, None, None)
        mat = ref.pack_material(color, (0, 0, 0), (1, 1, 1), pb)
        m = ref.material_from_bytes(color, (0, 0, 0), (1, 1, 1), pb)
        for cos_o in (1.0, 0.7, 0.4):
            wo = np.array([math.sqrt(1 - cos_o * cos_o), 0.0, cos_o])
            rows = np.zeros((len(dirs), 48), np.float32)
            rows[:, 0:24], rows[:, 24:27], rows[:, 27:30], rows[:, 30:33] = mat, N, wo, dirs
            pdf = evaluate(1, rows)[:, 0].astype(np.float64)
            f = evaluate(0, rows)[:, :3].astype(np.float64)
            assert np.isfinite(pdf).all() and (pdf >= 0).all() and np.isfinite(f).all() and (f >= 0).all(), name
            mass = (pdf * dw).sum()
            # the sampler reaches: diffuse lobe (1/2 of non-transmitted samples; its subsurface share goes below), specular lobe
            # (1/2; the part of it reflected below the horizon is lost).  So 0.5 < mass <= 1, and ~1 for a smooth lobe seen head-on.
            assert 0.5 < mass <= 1.0 + 2e-3, (name, cos_o, mass)
            if cos_o == 1.0:
                # closed form at normal incidence: the half vector's polar angle is distributed as GGX(alpha = roughness) and the mirrored
                # direction leaves the upper hemisphere when it exceeds 45 degrees, which has probability a^2 / (1 + a^2)
                a2 = m["roughness"] ** 2
                assert abs(mass - (0.5 + 0.5 / (1.0 + a2))) < 2e-3, (name, mass, 0.5 + 0.5 / (1.0 + a2))
            # sampled directions follow that pdf: E[g / pdf] over samples = integral of g over the reachable set, g = 1 on a cap around N
            r = rng.uniform(0.0005, 0.9995, (20000, 2)).astype(np.float32)
            srows = np.zeros((len(r), 48), np.float32)
            srows[:, 0:24], srows[:, 24:27], srows[:, 27:30], srows[:, 33:36], srows[:, 36:39] = mat, N, wo, T, B
            srows[:, 41], srows[:, 42] = r[:, 0], r[:, 1]
            s = evaluate(2, srows)
            wi, sp = s[:, :3].astype(np.float64), s[:, 3].astype(np.float64)
            ok = sp > 1e-4
            cap = wi[:, 2] > 0.5
            est = np.where(ok & cap, 1.0 / np.maximum(sp, 1e-30), 0.0).mean()
            want = 2.0 * math.pi * 0.5                 # solid angle of the cap z > 0.5, entirely reachable (pdf > 0 there)
            assert abs(est - want) < 0.05 * want, (name, cos_o, est, want)
            # white furnace: directional albedo of the reflective lobes (upper hemisphere) stays bounded
            up = dirs[:, 2] > 0
            albedo = (f[up] * dirs[up, 2:3] * dw[up, None]).sum(axis=0)
            assert (albedo < 1.08).all() and (albedo > 0.02).all(), (name, cos_o, albedo)
            # and the float64 formulation integrates to the same numbers
            sub = np.arange(0, len(dirs), 97)
            want_pdf = np.array([ref.bsdf_pdf(m, N, wo, dirs[i]) for i in sub])
            np.testing.assert_allclose(pdf[sub], want_pdf, rtol=2e-4, atol=1e-7)


def test_oracle_pdf_mass_sampling_and_white_furnace():
    from oracle.bindings import Oracle
    check_properties(Oracle(8, 8).eval_shading)


@pytest.mark.gpu
def test_device_shading_functions_against_float64_formulas():
    from rfw_rs_amd import HipBackend
    be = HipBackend.init(16, 16)
    check_tables(be.eval_shading)
    check_light_sampling(be, be.eval_shading)
    be.close()


@pytest.mark.gpu
def test_device_pdf_mass_sampling_and_white_furnace():
    from rfw_rs_amd import HipBackend
    be = HipBackend.init(16, 16)
    check_properties(be.eval_shading)
    # and the device's values are the oracle's, bit for bit (the arithmetic contract), on the random table
    from oracle.bindings import Oracle
    rows, _ = cases(23)
    orc = Oracle(8, 8)
    for op in (0, 1, 2, 3):
        assert np.array_equal(be.eval_shading(op, rows).view(np.uint32), orc.eval_shading(op, rows).view(np.uint32)), op
    be.close()


def bary_rows(n, seed):
    rng = np.random.default_rng(seed)
    r0 = rng.uniform(0.0, 1.0, n).astype(np.float32)
    r0[:8] = np.float32([0.0, 0.25, 0.5, 0.75, 1.0 - 2.0 ** -24, 2.0 ** -32, 1.0 / 3.0, 0.1])
    rows = np.zeros((n, 48), np.float32)
    rows[:, 41] = r0
    return rows, r0


def closed_form_barycentrics(r0):
    """The identity the device's RandomBarycentrics uses (csrc/shade_device.h): 3 * centroid of the last cell of the 16-level walk =
    (1, 1) + sum_i sign_i 2^-(i+1) c[d_i] — in exact integer arithmetic."""
    bits = min(int(float(np.float32(r0)) * 4294967296.0), 0xFFFFFFFF)

    def even_bits(x):
        x &= 0x55555555
        x = (x | (x >> 1)) & 0x33333333
        x = (x | (x >> 2)) & 0x0F0F0F0F
        x = (x | (x >> 4)) & 0x00FF00FF
        return (x | (x >> 8)) & 0xFFFF
    L, H = even_bits(bits), even_bits(bits >> 1)
    M1, M2, M3, Z = ~H & L & 0xFFFF, H & ~L & 0xFFFF, H & L, ~(H | L) & 0xFFFF
    t = Z
    for sh in (1, 2, 4, 8):
        t ^= t >> sh
    N = (t >> 1) & 0xFFFF
    d1, d2, d3 = ((M & ~N & 0xFFFF) - (M & N) for M in (M1, M2, M3))
    f = np.float32
    sx, sy = f(65536 + 2 * d1 - d2 - d3) * f(1.0 / 65536.0), f(65536 - d1 + 2 * d2 - d3) * f(1.0 / 65536.0)
    rx, ry = sx * f(0.3333333), sy * f(0.3333333)
    return np.array([rx, ry, f(1.0) - rx - ry], np.float32)


def test_random_barycentrics_closed_form():
    """shade.comp:371-411 is a 16-round walk with a 4-way switch per round; the device evaluates a branch-free closed form of it.  The
    closed form equals the oracle's literal loop bit for bit (every vertex of the walk is a dyadic rational: float32 is exact)."""
    from oracle.bindings import Oracle
    rows, r0 = bary_rows(20000, 2)
    got = Oracle(8, 8).eval_shading(5, rows)[:, :3]
    want = np.stack([closed_form_barycentrics(x) for x in r0])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.allclose(got.sum(axis=1), 1.0, atol=1e-6) and (got > -1e-6).all()
    # and the float64 formulation of the walk agrees
    for k in range(0, 2000, 7):
        assert np.allclose(got[k], ref.random_barycentrics(r0[k]), atol=1e-6)


@pytest.mark.gpu
def test_device_random_barycentrics_equal_the_oracle_loop():
    from oracle.bindings import Oracle
    from rfw_rs_amd import HipBackend
    rows, _ = bary_rows(200000, 3)
    be = HipBackend.init(16, 16)
    assert np.array_equal(be.eval_shading(5, rows)[:, :3].view(np.uint32), Oracle(8, 8).eval_shading(5, rows)[:, :3].view(np.uint32))
    be.close()
