"""Writes small glTF 2.0 documents for the loader tests (authored here, not taken from any asset)."""
import base64
import json
import struct

import numpy as np


def cube():
    """24 vertices with face normals and uvs, 36 indices (two triangles per face), unit cube centred at the origin."""
    pos, nor, uv, idx = [], [], [], []
    for axis in range(3):
        for sign in (-1.0, 1.0):
            n = np.zeros(3); n[axis] = sign
            u = np.zeros(3); u[(axis + 1) % 3] = 1.0
            v = np.cross(n, u)
            base = len(pos)
            for a, b in ((-1, -1), (1, -1), (1, 1), (-1, 1)):
                pos.append(0.5 * (n + a * u + b * v)); nor.append(n); uv.append(((a + 1) / 2, (b + 1) / 2))
            idx += [base, base + 1, base + 2, base, base + 2, base + 3]
    return np.array(pos, np.float32), np.array(nor, np.float32), np.array(uv, np.float32), np.array(idx, np.uint16)


def quat_y(angle):
    return [0.0, float(np.sin(angle / 2)), 0.0, float(np.cos(angle / 2))]


def build_document(embed=None):
    """Returns (json dict, binary blob).  A lit room in miniature: a floor quad (no normals, no indices), a scaled / rotated /
    translated cube under a parent node, and a downward-facing emissive quad (u32 indices) above them; one camera."""
    cp, cn, cuv, ci = cube()
    floor = np.array([[-4, 0, -4], [-4, 0, 4], [4, 0, 4], [-4, 0, -4], [4, 0, 4], [4, 0, -4]], np.float32)       # +y facing, 2 triangles, non-indexed
    lamp = np.array([[-0.5, 0, -0.5], [0.5, 0, -0.5], [0.5, 0, 0.5], [-0.5, 0, 0.5]], np.float32)
    lamp_idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)                                                       # -y facing (clockwise seen from above)
    chunks, views, accessors = [], [], []
    offset = 0

    def add(arr, ctype, atype, target=None, minmax=False):
        nonlocal offset
        raw = np.ascontiguousarray(arr).tobytes()
        pad = (-len(raw)) % 4
        views.append({"buffer": 0, "byteOffset": offset, "byteLength": len(raw), **({"target": target} if target else {})})
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": atype}
        if minmax:
            acc["min"] = [float(x) for x in arr.min(axis=0)]
            acc["max"] = [float(x) for x in arr.max(axis=0)]
        accessors.append(acc)
        chunks.append(raw + b"\0" * pad)
        offset += len(raw) + pad
        return len(accessors) - 1

    a_cp = add(cp, 5126, "VEC3", 34962, True); a_cn = add(cn, 5126, "VEC3", 34962); a_cuv = add(cuv, 5126, "VEC2", 34962)
    a_ci = add(ci, 5123, "SCALAR", 34963)
    a_fl = add(floor, 5126, "VEC3", 34962, True)
    a_lp = add(lamp, 5126, "VEC3", 34962, True); a_li = add(lamp_idx, 5125, "SCALAR", 34963)
    blob = b"".join(chunks)
    doc = {
        "asset": {"version": "2.0", "generator": "tests/gltf_util.py"},
        "scene": 0,
        "scenes": [{"nodes": [0, 3, 4, 5]}],
        "nodes": [
            {"name": "rig", "translation": [0.5, 0.0, 0.25], "rotation": quat_y(np.pi / 2), "scale": [2.0, 2.0, 2.0], "children": [1, 2]},
            {"name": "cube", "mesh": 0, "translation": [0.0, 0.25, 0.5]},
            {"name": "cube again", "mesh": 0, "matrix": [0.5, 0, 0, 0, 0, 0.5, 0, 0, 0, 0, 0.5, 0, -0.75, 0.125, 0.0, 1]},
            {"name": "floor", "mesh": 1},
            {"name": "lamp", "mesh": 2, "translation": [0.0, 3.0, 0.0]},
            {"name": "eye", "camera": 0, "translation": [0.0, 1.5, 6.0]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7, "znear": 0.01, "aspectRatio": 1.5}}],
        "meshes": [
            {"name": "cube", "primitives": [{"attributes": {"POSITION": a_cp, "NORMAL": a_cn, "TEXCOORD_0": a_cuv}, "indices": a_ci, "material": 0}]},
            {"name": "floor", "primitives": [{"attributes": {"POSITION": a_fl}, "material": 1}]},
            {"name": "lamp", "primitives": [{"attributes": {"POSITION": a_lp}, "indices": a_li, "material": 2, "mode": 4}]},
        ],
        "materials": [
            {"name": "red", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.1, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.6}},
            {"name": "grey", "pbrMetallicRoughness": {"baseColorFactor": [0.6, 0.6, 0.6, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9}},
            {"name": "emitter", "emissiveFactor": [1.0, 0.9, 0.8], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 12.0}}},
        ],
        "accessors": accessors,
        "bufferViews": views,
        "buffers": [{"byteLength": len(blob)}],
    }
    if embed == "base64":
        doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    elif embed is None:
        doc["buffers"][0]["uri"] = "scene.bin"
    return doc, blob


def write_gltf(directory, embed=None, mutate=None):
    """embed: None -> scene.gltf + scene.bin, "base64" -> scene.gltf with a data: uri, "glb" -> scene.glb.  Returns the path."""
    doc, blob = build_document(embed)
    if mutate:
        mutate(doc)
    if embed == "glb":
        js = json.dumps(doc).encode()
        js += b" " * ((-len(js)) % 4)
        bin_chunk = blob + b"\0" * ((-len(blob)) % 4)
        total = 12 + 8 + len(js) + 8 + len(bin_chunk)
        path = directory / "scene.glb"
        with open(path, "wb") as f:
            f.write(b"glTF" + struct.pack("<II", 2, total))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(bin_chunk), 0x004E4942) + bin_chunk)
        return path
    path = directory / "scene.gltf"
    path.write_text(json.dumps(doc))
    if embed is None:
        (directory / "scene.bin").write_bytes(blob)
    return path


def write_skinned_gltf(directory, bend=0.6):
    """A vertical strip (4 quads, x in [-0.2, 0.2], y in [0, 2]) bound to two joints: joint 0 at the origin, joint 1 at y = 1 (child
    of joint 0) rotated by `bend` about z.  Vertices at y <= 0.5 follow joint 0, at y >= 1.5 joint 1, in between they blend linearly.
    Returns (path, positions (n,3), joints (n,4) u8, weights (n,4), indices)."""
    ys = np.linspace(0.0, 2.0, 5)
    pos = np.array([[x, y, 0.0] for y in ys for x in (-0.2, 0.2)], np.float32)
    idx = []
    for k in range(4):
        a, b, c, d = 2 * k, 2 * k + 1, 2 * k + 2, 2 * k + 3
        idx += [a, b, d, a, d, c]
    idx = np.array(idx, np.uint8)
    w1 = np.clip((pos[:, 1] - 0.5) / 1.0, 0.0, 1.0).astype(np.float32)
    weights = np.stack([1.0 - w1, w1, np.zeros_like(w1), np.zeros_like(w1)], axis=1).astype(np.float32)
    joints = np.tile(np.array([0, 1, 0, 0], np.uint8), (len(pos), 1))
    ibm = np.stack([np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)])
    ibm[1][1, 3] = -1.0                                   # inverse bind of joint 1: translate by -1 in y
    ibm_cm = np.stack([m.T for m in ibm]).astype(np.float32)   # column-major
    chunks, views, accessors = [], [], []
    offset = 0

    def add(arr, ctype, atype):
        nonlocal offset
        raw = np.ascontiguousarray(arr).tobytes()
        pad = (-len(raw)) % 4
        views.append({"buffer": 0, "byteOffset": offset, "byteLength": len(raw)})
        accessors.append({"bufferView": len(views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": atype})
        chunks.append(raw + b"\0" * pad)
        offset += len(raw) + pad
        return len(accessors) - 1

    a_p = add(pos, 5126, "VEC3"); a_j = add(joints, 5121, "VEC4"); a_w = add(weights, 5126, "VEC4"); a_i = add(idx, 5121, "SCALAR")
    a_m = add(ibm_cm.reshape(2, 16), 5126, "MAT4")
    blob = b"".join(chunks)
    doc = {
        "asset": {"version": "2.0"},
        "scenes": [{"nodes": [0, 1]}],
        "nodes": [
            {"name": "strip", "mesh": 0, "skin": 0, "translation": [5.0, 5.0, 5.0]},   # ignored for a skinned mesh, as glTF prescribes
            {"name": "joint0", "children": [2]},
            {"name": "joint1", "translation": [0.0, 1.0, 0.0], "rotation": [0.0, 0.0, float(np.sin(bend / 2)), float(np.cos(bend / 2))]},
        ],
        "skins": [{"joints": [1, 2], "inverseBindMatrices": a_m}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": a_p, "JOINTS_0": a_j, "WEIGHTS_0": a_w}, "indices": a_i}]}],
        "accessors": accessors, "bufferViews": views,
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
    }
    path = directory / "skinned.gltf"
    path.write_text(json.dumps(doc))
    return path, pos, joints, weights, idx


def write_duplicates_gltf(directory, copies=3000):
    """One triangle repeated `copies` times (coincident centroids: no SAH plane separates them) next to a second, distinct one."""
    pos = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [2, 0, 1], [3, 0, 1], [2, 1, 1]], np.float32)
    idx = np.concatenate([np.tile(np.array([0, 1, 2], np.uint32), copies), np.array([3, 4, 5], np.uint32)])
    raw_p, raw_i = pos.tobytes(), idx.tobytes()
    blob = raw_p + raw_i
    doc = {
        "asset": {"version": "2.0"}, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0}, "indices": 1}]}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 6, "type": "VEC3"},
                      {"bufferView": 1, "componentType": 5125, "count": int(len(idx)), "type": "SCALAR"}],
        "bufferViews": [{"buffer": 0, "byteOffset": 0, "byteLength": len(raw_p)}, {"buffer": 0, "byteOffset": len(raw_p), "byteLength": len(raw_i)}],
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
    }
    path = directory / "dups.gltf"
    path.write_text(json.dumps(doc))
    return path


def encode_png(rgba, filters=(0, 1, 2, 3, 4), colour=6, palette=None, trns=None):
    """Minimal PNG writer (8-bit, non-interlaced) that cycles through the five scanline filters so that a decoder's
    un-filtering is exercised.  rgba: (h, w, c) uint8 with c = 4 (colour 6, RGBA), 3 (colour 2, RGB), 2 (colour 4, grey + alpha) or
    1 (colour 0, grey; colour 3, palette indices with `palette` (n, 3) and optional `trns` (m,))."""
    import zlib
    h, w, c = rgba.shape
    bpp, stride = c, w * c
    raw = bytearray()
    prev = np.zeros(stride, np.int32)
    for y in range(h):
        cur = rgba[y].reshape(-1).astype(np.int32)
        f = filters[y % len(filters)]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if f == 0:
            out = cur
        elif f == 1:
            out = cur - left
        elif f == 2:
            out = cur - prev
        elif f == 3:
            out = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out = cur - pred
        raw.append(f)
        raw += bytes((out & 0xFF).astype(np.uint8))
        prev = cur

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    extra = b""
    if palette is not None:
        extra += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        extra += chunk(b"tRNS", np.asarray(trns, np.uint8).tobytes())
    data = zlib.compress(bytes(raw), 6)
    half = len(data) // 2                                   # two IDAT chunks: a decoder has to concatenate them
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, colour, 0, 0, 0)) + extra
            + chunk(b"IDAT", data[:half]) + chunk(b"IDAT", data[half:]) + chunk(b"IEND", b""))


def write_textured_gltf(directory, glb=False, png=None):
    """A lit floor with a base-colour PNG (8 x 8, every texel different) and a lamp above it.  Returns (path, texels (8, 8, 4) RGBA)."""
    rng = np.random.default_rng(9)
    tex = rng.integers(30, 255, size=(8, 8, 4), dtype=np.uint8)
    tex[..., 3] = 255
    png = encode_png(tex) if png is None else png
    floor = np.array([[-2, 0, -2], [-2, 0, 2], [2, 0, 2], [2, 0, -2]], np.float32)
    uv = np.array([[0, 0], [0, 1], [1, 1], [1, 0]], np.float32)
    fidx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    lamp = np.array([[-0.5, 3, -0.5], [0.5, 3, -0.5], [0.5, 3, 0.5], [-0.5, 3, 0.5]], np.float32)
    lidx = np.array([0, 1, 2, 0, 2, 3], np.uint16)
    parts = [floor.tobytes(), uv.tobytes(), fidx.tobytes(), lamp.tobytes(), lidx.tobytes()]
    if glb:
        parts.append(png)
    views, offset, blob = [], 0, b""
    for raw in parts:
        views.append({"buffer": 0, "byteOffset": offset, "byteLength": len(raw)})
        pad = (-len(raw)) % 4
        blob += raw + b"\0" * pad
        offset += len(raw) + pad
    doc = {
        "asset": {"version": "2.0"}, "scenes": [{"nodes": [0, 1, 2]}],
        "nodes": [{"mesh": 0}, {"mesh": 1}, {"camera": 0, "translation": [0, 1.5, 5]}],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.8, "znear": 0.01}}],
        "meshes": [{"primitives": [{"attributes": {"POSITION": 0, "TEXCOORD_0": 1}, "indices": 2, "material": 0}]},
                   {"primitives": [{"attributes": {"POSITION": 3}, "indices": 4, "material": 1}]}],
        "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}, "metallicFactor": 0.0, "roughnessFactor": 0.8}},
                      {"emissiveFactor": [1, 1, 1], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 15.0}}}],
        "textures": [{"source": 0}],
        "images": [{"bufferView": 5, "mimeType": "image/png"} if glb else {"uri": "floor.png"}],
        "accessors": [{"bufferView": 0, "componentType": 5126, "count": 4, "type": "VEC3"}, {"bufferView": 1, "componentType": 5126, "count": 4, "type": "VEC2"},
                      {"bufferView": 2, "componentType": 5123, "count": 6, "type": "SCALAR"}, {"bufferView": 3, "componentType": 5126, "count": 4, "type": "VEC3"},
                      {"bufferView": 4, "componentType": 5123, "count": 6, "type": "SCALAR"}],
        "bufferViews": views, "buffers": [{"byteLength": len(blob)}],
    }
    if glb:
        js = json.dumps(doc).encode()
        js += b" " * ((-len(js)) % 4)
        path = directory / "textured.glb"
        with open(path, "wb") as f:
            f.write(b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)))
            f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
            f.write(struct.pack("<II", len(blob), 0x004E4942) + blob)
        return path, tex
    doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    (directory / "floor.png").write_bytes(png)
    path = directory / "textured.gltf"
    path.write_text(json.dumps(doc))
    return path, tex


def write_animated_gltf(directory, jpeg=None):
    """A small lit room with everything the animation importer handles: a skinned tube (8-sided, 3 joints in a chain) whose joints turn
    (LINEAR rotation keys, one as normalised int16 with a sign flip between keys: shortest-arc slerp) and whose root joint slides (LINEAR
    translation), a rigid cube under a parent node moved by a CUBICSPLINE translation and a STEP scale, a static matrix node, a floor, an
    emissive lamp, a camera.  One animation of 2 s.  `jpeg`: bytes of a JPEG file for the floor's base colour (None: untextured).
    Returns the path of animated.gltf (buffers embedded)."""
    cp, cn, cuv, ci = cube()
    rings, sides = 9, 8
    ys = np.linspace(0.0, 2.0, rings)
    pos = np.array([[0.15 * np.cos(2 * np.pi * k / sides), y, 0.15 * np.sin(2 * np.pi * k / sides)] for y in ys for k in range(sides)], np.float32)
    idx = []
    for r in range(rings - 1):
        for k in range(sides):
            a, b = r * sides + k, r * sides + (k + 1) % sides
            c, d = a + sides, b + sides
            idx += [a, d, b, a, c, d]
    idx = np.array(idx, np.uint16)
    # joints at y = 0, 0.8, 1.4; weights blend over 0.4 around each joint's height
    w1 = np.clip((pos[:, 1] - 0.6) / 0.4, 0.0, 1.0)
    w2 = np.clip((pos[:, 1] - 1.2) / 0.4, 0.0, 1.0)
    weights = np.stack([1.0 - w1, w1 - w2, w2, np.zeros_like(w1)], axis=1).astype(np.float32)
    joints = np.tile(np.array([0, 1, 2, 0], np.uint8), (len(pos), 1))
    ibm = np.stack([np.eye(4, dtype=np.float32)] * 3)
    ibm[1][1, 3] = -0.8
    ibm[2][1, 3] = -1.4
    floor = np.array([[-4, 0, -4], [-4, 0, 4], [4, 0, 4], [-4, 0, -4], [4, 0, 4], [4, 0, -4]], np.float32)
    floor_uv = np.array([[0, 0], [0, 1], [1, 1], [0, 0], [1, 1], [1, 0]], np.float32)
    lamp = np.array([[-0.7, 0, -0.7], [0.7, 0, -0.7], [0.7, 0, 0.7], [-0.7, 0, 0.7]], np.float32)
    lamp_idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    chunks, views, accessors = [], [], []
    offset = 0

    def add(arr, ctype, atype, normalized=False):
        nonlocal offset
        raw = np.ascontiguousarray(arr).tobytes()
        pad = (-len(raw)) % 4
        views.append({"buffer": 0, "byteOffset": offset, "byteLength": len(raw)})
        acc = {"bufferView": len(views) - 1, "componentType": ctype, "count": int(arr.shape[0]), "type": atype}
        if normalized:
            acc["normalized"] = True
        if atype == "SCALAR" and ctype == 5126:
            acc["min"], acc["max"] = [float(arr.min())], [float(arr.max())]
        accessors.append(acc)
        chunks.append(raw + b"\0" * pad)
        offset += len(raw) + pad
        return len(accessors) - 1

    def qz(a):
        return [0.0, 0.0, float(np.sin(a / 2)), float(np.cos(a / 2))]

    def qx(a):
        return [float(np.sin(a / 2)), 0.0, 0.0, float(np.cos(a / 2))]

    a_p = add(pos, 5126, "VEC3"); a_j = add(joints, 5121, "VEC4"); a_w = add(weights, 5126, "VEC4"); a_i = add(idx, 5123, "SCALAR")
    a_m = add(np.stack([m.T for m in ibm]).reshape(3, 16).astype(np.float32), 5126, "MAT4")
    a_cp = add(cp, 5126, "VEC3"); a_cn = add(cn, 5126, "VEC3"); a_ci = add(ci, 5123, "SCALAR")
    a_fl = add(floor, 5126, "VEC3"); a_fuv = add(floor_uv, 5126, "VEC2")
    a_lp = add(lamp, 5126, "VEC3"); a_li = add(lamp_idx, 5125, "SCALAR")
    # samplers
    t5 = add(np.array([0.0, 0.5, 1.0, 1.5, 2.0], np.float32), 5126, "SCALAR")
    t3 = add(np.array([0.25, 1.0, 1.75], np.float32), 5126, "SCALAR")
    r1 = add(np.array([qz(0.0), qz(0.7), qz(-0.5), qz(0.9), qz(0.0)], np.float32), 5126, "VEC4")            # joint 1 about z
    q2 = np.array([qx(0.0), qx(0.8), qx(-0.6)], np.float64)
    q2[1] *= -1.0                                                                                            # same rotation, opposite sign
    r2 = add(np.round(q2 * 32767.0).astype(np.int16), 5122, "VEC4", normalized=True)                          # joint 2 about x, int16 keys
    tr0 = add(np.array([[0, 0, 0], [0.3, 0, 0], [0.3, 0, 0.4], [-0.2, 0, 0.4], [0, 0, 0]], np.float32), 5126, "VEC3")  # root joint slides
    # cubic spline: (in-tangent, value, out-tangent) per key
    cs = np.array([[[0, 0, 0], [1.5, 0.5, 0.0], [0.0, 1.0, 0.0]],
                   [[0.5, 0.0, 0.0], [1.5, 1.2, 0.6], [0.5, 0.0, -0.5]],
                   [[0.0, -1.0, 0.0], [1.2, 0.5, -0.4], [0, 0, 0]]], np.float32).reshape(9, 3)
    tc = add(cs, 5126, "VEC3")
    sc = add(np.array([[1, 1, 1], [1.5, 0.5, 1.5], [0.8, 1.6, 0.8]], np.float32), 5126, "VEC3")
    blob = b"".join(chunks)
    doc = {
        "asset": {"version": "2.0", "generator": "tests/gltf_util.py"},
        "scene": 0,
        "scenes": [{"nodes": [0, 1, 4, 6, 7, 8, 9]}],
        "nodes": [
            {"name": "tube", "mesh": 0, "skin": 0},
            {"name": "joint0", "translation": [0.0, 0.0, 0.0], "children": [2]},
            {"name": "joint1", "translation": [0.0, 0.8, 0.0], "children": [3]},
            {"name": "joint2", "translation": [0.0, 0.6, 0.0]},
            {"name": "arm", "translation": [0.0, 0.0, -0.5], "rotation": quat_y(0.4), "children": [5]},
            {"name": "cube", "mesh": 1, "translation": [1.5, 0.5, 0.0], "scale": [1.0, 1.0, 1.0]},
            {"name": "cube static", "mesh": 1, "matrix": [0.4, 0, 0, 0, 0, 0.4, 0, 0, 0, 0, 0.4, 0, -1.5, 0.2, 0.3, 1]},
            {"name": "floor", "mesh": 2},
            {"name": "lamp", "mesh": 3, "translation": [0.0, 3.2, 0.5]},
            {"name": "eye", "camera": 0, "translation": [0.0, 1.3, 5.5]},
        ],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.75, "znear": 0.01, "aspectRatio": 1.5}}],
        "skins": [{"joints": [1, 2, 3], "inverseBindMatrices": a_m}],
        "meshes": [
            {"name": "tube", "primitives": [{"attributes": {"POSITION": a_p, "JOINTS_0": a_j, "WEIGHTS_0": a_w}, "indices": a_i, "material": 0}]},
            {"name": "cube", "primitives": [{"attributes": {"POSITION": a_cp, "NORMAL": a_cn}, "indices": a_ci, "material": 1}]},
            {"name": "floor", "primitives": [{"attributes": {"POSITION": a_fl, "TEXCOORD_0": a_fuv}, "material": 2}]},
            {"name": "lamp", "primitives": [{"attributes": {"POSITION": a_lp}, "indices": a_li, "material": 3}]},
        ],
        "materials": [
            {"name": "skin", "pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.6, 0.8, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.7}},
            {"name": "red", "pbrMetallicRoughness": {"baseColorFactor": [0.8, 0.1, 0.1, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.6}},
            {"name": "grey", "pbrMetallicRoughness": {"baseColorFactor": [0.7, 0.7, 0.7, 1.0], "metallicFactor": 0.0, "roughnessFactor": 0.9}},
            {"name": "emitter", "emissiveFactor": [1.0, 0.9, 0.8], "extensions": {"KHR_materials_emissive_strength": {"emissiveStrength": 14.0}}},
        ],
        "animations": [{
            "name": "all",
            "samplers": [
                {"input": t5, "output": r1, "interpolation": "LINEAR"},
                {"input": t3, "output": r2},                                  # LINEAR is the default
                {"input": t5, "output": tr0, "interpolation": "LINEAR"},
                {"input": t3, "output": tc, "interpolation": "CUBICSPLINE"},
                {"input": t3, "output": sc, "interpolation": "STEP"},
            ],
            "channels": [
                {"sampler": 0, "target": {"node": 2, "path": "rotation"}},
                {"sampler": 1, "target": {"node": 3, "path": "rotation"}},
                {"sampler": 2, "target": {"node": 1, "path": "translation"}},
                {"sampler": 3, "target": {"node": 5, "path": "translation"}},
                {"sampler": 4, "target": {"node": 5, "path": "scale"}},
                {"sampler": 0, "target": {"node": 0, "path": "weights"}},      # morph weights: ignored (no morph targets anywhere)
            ],
        }],
        "accessors": accessors, "bufferViews": views,
        "buffers": [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}],
    }
    if jpeg is not None:
        doc["images"] = [{"uri": "data:image/jpeg;base64," + base64.b64encode(jpeg).decode()}]
        doc["textures"] = [{"source": 0}]
        doc["materials"][2]["pbrMetallicRoughness"]["baseColorTexture"] = {"index": 0}
    path = directory / "animated.gltf"
    path.write_text(json.dumps(doc))
    return path


def encode_tga(img, rle=False, top_down=False):
    """(h, w, 4) RGBA, (h, w, 3) RGB or (h, w) grey uint8 -> TGA bytes (types 2 / 3, run-length encoded 10 / 11)."""
    img = np.asarray(img, np.uint8)
    grey = img.ndim == 2
    h, w = img.shape[:2]
    ch = 1 if grey else img.shape[2]
    px = img.reshape(h, w, 1) if grey else img[..., [2, 1, 0] + ([3] if ch == 4 else [])]        # stored B, G, R (, A)
    rows = px if top_down else px[::-1]
    flat = rows.reshape(-1, ch)
    head = struct.pack("<BBBHHBHHHHBB", 0, 0, (3 if grey else 2) + (8 if rle else 0), 0, 0, 0, 0, 0, w, h, 8 * ch, (0x20 if top_down else 0) | (8 if ch == 4 else 0))
    if not rle:
        return head + flat.tobytes()
    out = bytearray()
    i, n = 0, len(flat)
    while i < n:
        run = 1
        while i + run < n and run < 128 and (flat[i + run] == flat[i]).all():
            run += 1
        if run > 1:
            out += bytes([0x80 | (run - 1)]) + flat[i].tobytes()
            i += run
        else:
            lit = 1
            while i + lit < n and lit < 128 and not (i + lit + 1 < n and (flat[i + lit] == flat[i + lit + 1]).all()):
                lit += 1
            out += bytes([lit - 1]) + flat[i:i + lit].tobytes()
            i += lit
    return head + bytes(out)


def encode_png_general(samples, colour, depth, interlace=False, palette=None, trns=None, filters=(0, 1, 2, 3, 4)):
    """PNG writer for every colour type / bit depth / interlace method of the specification.  samples: (h, w, c) integers as the file
    stores them (0 .. 2**depth - 1; palette indices for colour 3).  Packs sub-byte depths most significant bits first, 16-bit samples
    big-endian; Adam7 writes the seven passes in order, each with its own filter state."""
    import zlib
    samples = np.asarray(samples)
    h, w, c = samples.shape
    bits_pp = c * depth
    bpp = max(1, bits_pp // 8)

    def pack_row(row):                                      # (n, c) -> bytes
        flat = row.reshape(-1).astype(np.int64)
        if depth == 8:
            return flat.astype(np.uint8)
        if depth == 16:
            return np.stack([flat >> 8, flat & 255], axis=1).reshape(-1).astype(np.uint8)
        per = 8 // depth
        pad = (-len(flat)) % per
        flat = np.concatenate([flat, np.zeros(pad, np.int64)]).reshape(-1, per)
        shifts = np.array([8 - depth * (k + 1) for k in range(per)])
        return (flat << shifts).sum(axis=1).astype(np.uint8)

    def filter_rows(rows, counter):
        out = bytearray()
        prev = None
        for row in rows:
            cur = pack_row(row).astype(np.int32)
            if prev is None:
                prev = np.zeros_like(cur)
            f = filters[counter[0] % len(filters)]
            counter[0] += 1
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            upleft = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]]) if len(cur) > bpp else np.zeros_like(cur)
            if f == 0:
                o = cur
            elif f == 1:
                o = cur - left
            elif f == 2:
                o = cur - prev
            elif f == 3:
                o = cur - ((left + prev) >> 1)
            else:
                p = left + prev - upleft
                pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
                o = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            out.append(f)
            out += bytes((o & 0xFF).astype(np.uint8))
            prev = cur
        return out

    counter = [0]
    raw = bytearray()
    if interlace:
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += filter_rows(sub, counter)
    else:
        raw += filter_rows(samples, counter)

    def chunk(tag, body):
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)
    extra = b""
    if palette is not None:
        extra += chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        extra += chunk(b"tRNS", bytes(trns))
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, colour, 0, 0, 1 if interlace else 0)) + extra
            + chunk(b"IDAT", zlib.compress(bytes(raw), 6)) + chunk(b"IEND", b""))
