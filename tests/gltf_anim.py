"""Independent float64 evaluation of a glTF 2.0 document's animation and skin (numpy): what the importer's node graph must reproduce.
Written from the glTF 2.0 specification (sections 3.7.3.3 skins, 3.11 animations, appendix C interpolation), not from host/gltf.cpp:
recursive world matrices, per-channel key search with numpy, slerp through angles.  Only .gltf (JSON + external .bin / data: URIs)."""
import base64
import json
import os

import numpy as np

CT = {5120: np.int8, 5121: np.uint8, 5122: np.int16, 5123: np.uint16, 5125: np.uint32, 5126: np.float32}
NC = {"SCALAR": 1, "VEC2": 2, "VEC3": 3, "VEC4": 4, "MAT4": 16}


class Document:
    def __init__(self, path):
        self.j = json.load(open(path))
        d = os.path.dirname(path)
        self.buffers = []
        for b in self.j.get("buffers", []):
            uri = b["uri"]
            self.buffers.append(base64.b64decode(uri.split(",", 1)[1]) if uri.startswith("data:") else open(os.path.join(d, uri), "rb").read())

    def accessor(self, i):
        a = self.j["accessors"][i]
        bv = self.j["bufferViews"][a["bufferView"]]
        dt, nc = np.dtype(CT[a["componentType"]]), NC[a["type"]]
        stride = bv.get("byteStride", 0) or dt.itemsize * nc
        base = bv.get("byteOffset", 0) + a.get("byteOffset", 0)
        raw = self.buffers[bv["buffer"]]
        out = np.empty((a["count"], nc), np.float64)
        for k in range(a["count"]):
            out[k] = np.frombuffer(raw, dt, nc, base + k * stride)
        if a.get("normalized"):
            out = np.maximum(out / np.iinfo(dt).max, -1.0) if dt.kind == "i" else out / np.iinfo(dt).max
        return out


def quat_matrix(q):
    x, y, z, w = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def trs(t, q, s):
    m = np.eye(4)
    m[:3, :3] = quat_matrix(q) @ np.diag(s)
    m[:3, 3] = t
    return m


def slerp(a, b, u):
    d = float(np.dot(a, b))
    if d < 0.0:
        b, d = -b, -d
    if d > 0.9995:
        r = a + u * (b - a)
    else:
        th = np.arccos(d)
        r = (np.sin((1 - u) * th) * a + np.sin(u * th) * b) / np.sin(th)
    return r / np.linalg.norm(r)


def sample(times, values, interpolation, t, rotation):
    times = times[:, 0]
    cubic = interpolation == "CUBICSPLINE"
    val = (lambda k: values[3 * k + 1]) if cubic else (lambda k: values[k])
    norm = (lambda v: v / np.linalg.norm(v)) if rotation else (lambda v: v)
    if t <= times[0]:
        return norm(val(0))
    if t >= times[-1]:
        return norm(val(len(times) - 1))
    k = int(np.searchsorted(times, t, side="right")) - 1
    dt = times[k + 1] - times[k]
    u = (t - times[k]) / dt
    if interpolation == "STEP":
        return norm(val(k))
    if cubic:
        p0, m0, p1, m1 = values[3 * k + 1], values[3 * k + 2] * dt, values[3 * k + 4], values[3 * k + 3] * dt
        return norm((2 * u**3 - 3 * u**2 + 1) * p0 + (u**3 - 2 * u**2 + u) * m0 + (-2 * u**3 + 3 * u**2) * p1 + (u**3 - u**2) * m1)
    return slerp(val(k), val(k + 1), u) if rotation else val(k) + u * (val(k + 1) - val(k))


def evaluate(path, time, animation=0):
    """-> (world matrices per node, joint matrices per skin) at `time` seconds of animation `animation` (looping)."""
    doc = Document(path)
    nodes = doc.j["nodes"]
    T = [np.array(n.get("translation", [0, 0, 0]), float) for n in nodes]
    R = [np.array(n.get("rotation", [0, 0, 0, 1]), float) for n in nodes]
    S = [np.array(n.get("scale", [1, 1, 1]), float) for n in nodes]
    anims = doc.j.get("animations", [])
    if anims:
        an = anims[animation]
        samplers = [(doc.accessor(s["input"]), doc.accessor(s["output"]), s.get("interpolation", "LINEAR")) for s in an["samplers"]]
        duration = max(s[0][-1, 0] for s in samplers)
        t = np.fmod(time, duration) if duration > 0 else 0.0
        if t < 0:
            t += duration
        for ch in an["channels"]:
            tg = ch["target"]
            if tg["path"] not in ("translation", "rotation", "scale") or "matrix" in nodes[tg["node"]]:
                continue
            times, values, ip = samplers[ch["sampler"]]
            v = sample(times, values, ip, t, tg["path"] == "rotation")
            {"translation": T, "rotation": R, "scale": S}[tg["path"]][tg["node"]] = v
    world = [None] * len(nodes)

    def visit(i, parent):
        n = nodes[i]
        local = np.array(n["matrix"], float).reshape(4, 4).T if "matrix" in n else trs(T[i], R[i], S[i])
        world[i] = parent @ local
        for c in n.get("children", []):
            visit(c, world[i])

    for r in doc.j["scenes"][doc.j.get("scene", 0)]["nodes"]:
        visit(r, np.eye(4))
    skins = []
    for sk in doc.j.get("skins", []):
        ibm = doc.accessor(sk["inverseBindMatrices"]).reshape(-1, 4, 4).transpose(0, 2, 1) if "inverseBindMatrices" in sk else None
        skins.append(np.stack([world[j] @ (ibm[k] if ibm is not None else np.eye(4)) for k, j in enumerate(sk["joints"])]))
    return world, skins
