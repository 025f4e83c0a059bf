"""glTF 2.0 import of the C++ host (rfw-rs_amd/host/gltf.cpp): the scene that reaches the boundary."""
import numpy as np
import pytest

from gltf_util import cube, write_gltf
from oracle.bindings import Oracle
from rfw_rs_amd import Scene


def load(path):
    scene = Scene().load_gltf(str(path))
    orc = Oracle(48, 32, threads=2)
    scene.sync(orc)
    return scene, orc


def test_meshes_materials_instances_lights(tmp_path):
    scene, orc = load(write_gltf(tmp_path))
    c = scene.counts()
    assert c["meshes"] == 3 and c["materials"] == 3
    assert c["instances"] == 4                       # two cubes, floor, lamp
    assert c["area_lights"] == 2                     # the lamp's two emissive triangles
    assert orc.stats()["n_tris"] == 12 + 2 + 2 and orc.validate_bvh() == 0
    tris = orc.triangles()                           # mesh-local, de-indexed, meshes in id order: cube, floor, lamp
    cp, cn, _, ci = cube()
    assert np.array_equal(tris[:12, [0, 1, 2]], cp[ci[0::3]]) and np.array_equal(tris[:12, [4, 5, 6]], cp[ci[1::3]])
    assert np.array_equal(tris[:12, [16, 17, 18]], cn[ci[0::3]])          # vertex normals as given
    assert np.allclose(tris[12:14, 12:15], [0, 1, 0])                      # floor: normals generated, +y
    assert np.allclose(tris[14:16, 12:15], [0, -1, 0])                     # lamp faces down


def test_node_hierarchy_places_the_instances(tmp_path):
    scene, orc = load(write_gltf(tmp_path))
    # world = T(.5,0,.25) * Ry(90deg) * S(2) * T(0,.25,.5) for the first cube: local (x,y,z) -> (2(z+.5) + .5, 2(y+.25), -2x + .25)
    # => centre (1.5, 0.5, 0.25), edge 2: top face at y = 1.5.  Second cube: rig * (0.5 p + (-.75,.125,0)) => centre (0.5, 0.25, 1.75), edge 1: top at 0.75
    o = np.array([[1.5, 10, 0.25], [0.5, 10, 1.75], [-3, 10, -3], [0.2, 10, 0.1]], np.float32)
    d = np.tile(np.array([0, -1, 0], np.float32), (4, 1))
    hit = orc.intersect(o, d, brute=True)
    assert np.allclose(hit["t"], [8.5, 9.25, 10.0, 7.0], atol=1e-5)       # cube tops, floor, lamp (y = 3, seen from above)
    assert list(hit["inst"]) == [0, 1, 2, 3]
    v = scene.view(48, 32)
    assert np.allclose([v.pos.x, v.pos.y, v.pos.z], [0, 1.5, 6])          # the document's camera
    orc.render(v)
    st = orc.stats()
    assert st["shadow"] > 100 and st["extension"] > 100               # the surfaces shade: generated tangents give a valid frame


@pytest.mark.parametrize("embed", ["base64", "glb"])
def test_containers_give_the_same_scene(tmp_path, embed):
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    _, ref = load(write_gltf(tmp_path / "a"))
    _, other = load(write_gltf(tmp_path / "b", embed))
    assert np.array_equal(ref.triangles().view(np.uint32), other.triangles().view(np.uint32))
    o = np.random.default_rng(1).uniform(-3, 3, (500, 3)).astype(np.float32); o[:, 1] = 8
    d = np.tile(np.array([0, -1, 0], np.float32), (500, 1))
    a, b = ref.intersect(o, d), other.intersect(o, d)
    assert np.array_equal(a["tri"], b["tri"]) and np.array_equal(a["t"].view(np.uint32), b["t"].view(np.uint32))


def test_malformed_documents_are_errors_not_crashes(tmp_path):
    def expect(mutate, text, sub):
        d = tmp_path / sub
        d.mkdir()
        with pytest.raises(ValueError, match=text):
            Scene().load_gltf(str(write_gltf(d, mutate=mutate)))
    expect(lambda doc: doc["accessors"][3].update(count=60), "past the end", "a")                 # index accessor longer than its view
    expect(lambda doc: doc["meshes"][2]["primitives"][0].update(indices=3), "index out of range", "b")  # the cube's indices (0..23) on the 4-vertex lamp
    expect(lambda doc: doc["asset"].update(version="1.0"), "version", "c")
    expect(lambda doc: doc["buffers"][0].update(uri="../scene.bin"), "below the document", "d")
    expect(lambda doc: doc["accessors"][0].update(sparse={"count": 1}), "sparse", "e")
    # offsets that would wrap the bounds arithmetic into range (round 6, found by a mutation run under the address sanitizer: a view at byte -1
    # made `offset + length` small again and the accessor read in front of its buffer)
    expect(lambda doc: doc["bufferViews"][0].update(byteOffset=-1), "out of range", "f")
    expect(lambda doc: doc["accessors"][0].update(byteOffset=-4), "out of range", "g")
    expect(lambda doc: doc["bufferViews"][1].update(byteOffset=2 ** 63 - 1), "out of range", "h")
    expect(lambda doc: doc["bufferViews"][0].update(byteStride=2 ** 40), "out of range", "i")
    bad = tmp_path / "bad.gltf"
    bad.write_text('{"asset": {"version": "2.0"}, "meshes": [')
    with pytest.raises(ValueError, match="json"):
        Scene().load_gltf(str(bad))
    with pytest.raises(ValueError, match="cannot read"):
        Scene().load_gltf(str(tmp_path / "missing.gltf"))


def test_skinned_document(tmp_path):
    """JOINTS_0 / WEIGHTS_0 / skins -> JointData + SkinData; the skinned copy the backend sees is M v with
    M = sum w_k (world(joint_k) * inverseBind_k)."""
    from gltf_util import write_skinned_gltf
    bend = 0.6
    path, pos, joints, weights, idx = write_skinned_gltf(tmp_path, bend)
    scene, orc = load(path)
    assert scene.counts()["meshes"] == 1 and scene.counts()["instances"] == 1
    tris = orc.triangles()
    assert tris.shape[0] == 16                                   # 8 bind-pose triangles + their skinned copy
    c, s = np.cos(bend), np.sin(bend)
    j1 = np.array([[c, -s, 0, 0], [s, c, 0, 1], [0, 0, 1, 0], [0, 0, 0, 1]], np.float64)   # world(joint1) = T(0,1,0) Rz(bend)
    ibm1 = np.eye(4); ibm1[1, 3] = -1.0
    J = [np.eye(4), j1 @ ibm1]
    want = []
    for i in idx:
        M = sum(float(weights[i, k]) * J[int(joints[i, k])] for k in range(4))
        want.append((M @ np.append(pos[i].astype(np.float64), 1.0))[:3])
    want = np.array(want).reshape(8, 3, 3)
    got = tris[8:, [0, 1, 2, 4, 5, 6, 8, 9, 10]].reshape(8, 3, 3)
    assert np.allclose(got, want, atol=1e-6)
    assert np.array_equal(tris[:8, [0, 1, 2]], pos[idx[0::3]])   # the bind pose stays as the static mesh
    # the skinned mesh's own node transform (5,5,5) is ignored: a ray down the bent tip finds it near the origin, not at x = 5
    tip = want[-1].mean(axis=0)
    hit = orc.intersect(np.array([[tip[0], tip[1], 4.0]], np.float32), np.array([[0, 0, -1]], np.float32), brute=True)
    assert hit["inst"][0] == 0 and abs(hit["t"][0] - 4.0) < 1e-4


@pytest.mark.parametrize("glb", [False, True])
def test_png_textures_reach_the_sampler(tmp_path, glb):
    """images (PNG: all five scanline filters) -> BGRA8 textures with mips -> material.diffuse_map; the sampler returns the texels."""
    from gltf_util import write_textured_gltf
    path, tex = write_textured_gltf(tmp_path, glb)
    scene, orc = load(path)
    for (x, y) in [(0, 0), (3, 5), (7, 7), (6, 1)]:
        got = orc.sample_texture(0, (x + 0.5) / 8, (y + 0.5) / 8, 0.0)
        assert np.array_equal(got, tex[y, x].astype(np.float32) * np.float32(1.0 / 255.0)), (x, y)
    # as a layer of gpu-rt's 1024 x 1024 x 5 texture array every level still resolves the 8 x 8 source texels (level 4 is 64 x 64)
    assert np.array_equal(orc.sample_texture(0, 0.3, 0.3, 4.0), tex[2, 2].astype(np.float32) * np.float32(1.0 / 255.0))
    orc2 = Oracle(48, 32, threads=2)
    orc2.set_option("texture_array", 0)                          # at its native size the importer's own mip chain is sampled
    scene.mark_all_changed(); scene.sync(orc2)
    assert np.allclose(orc2.sample_texture(0, 0.3, 0.3, 3.0)[:3], tex[..., :3].reshape(-1, 3).mean(axis=0) / 255.0, atol=0.02)   # 1x1 mip = the mean
    # the floor is lit and textured: a render is not uniform grey
    v = scene.view(48, 32)
    orc.render(v)
    acc = orc.accumulator()[..., :3]
    assert acc.max() > 0 and acc[20:, :, :].std() > 1e-3


def test_png_colour_types_and_broken_images(tmp_path):
    """Grey, grey + alpha, RGB and palette (+ tRNS) PNGs expand to the same RGBA; an image that cannot be decoded leaves the material
    untextured instead of failing the import."""
    from gltf_util import encode_png, write_textured_gltf
    rng = np.random.default_rng(3)
    idx = rng.integers(0, 5, size=(8, 8, 1), dtype=np.uint8)
    pal = rng.integers(0, 255, size=(5, 3), dtype=np.uint8)
    trns = np.array([255, 10, 200], np.uint8)
    grey = rng.integers(0, 255, size=(8, 8, 1), dtype=np.uint8)
    ga = rng.integers(0, 255, size=(8, 8, 2), dtype=np.uint8)
    rgb = rng.integers(0, 255, size=(8, 8, 3), dtype=np.uint8)
    cases = {
        "palette": (encode_png(idx, colour=3, palette=pal, trns=trns),
                    lambda x, y: np.append(pal[idx[y, x, 0]], trns[idx[y, x, 0]] if idx[y, x, 0] < 3 else 255)),
        "grey": (encode_png(grey, colour=0), lambda x, y: np.array([grey[y, x, 0]] * 3 + [255])),
        "grey_alpha": (encode_png(ga, colour=4), lambda x, y: np.array([ga[y, x, 0]] * 3 + [ga[y, x, 1]])),
        "rgb": (encode_png(rgb, colour=2), lambda x, y: np.append(rgb[y, x], 255)),
    }
    for name, (png, want) in cases.items():
        d = tmp_path / name
        d.mkdir()
        _, orc = load(write_textured_gltf(d, png=png)[0])
        for (x, y) in [(0, 0), (5, 2), (7, 7)]:
            got = orc.sample_texture(0, (x + 0.5) / 8, (y + 0.5) / 8, 0.0)
            assert np.array_equal(got, want(x, y).astype(np.float32) * np.float32(1.0 / 255.0)), (name, x, y)
    good = encode_png(rgb, colour=2)
    for name, png in {"truncated": good[: len(good) // 2], "not_png": b"JFIF" + good[4:], "interlaced": good[:28] + b"\x01" + good[29:]}.items():
        d = tmp_path / name
        d.mkdir()
        scene, orc = load(write_textured_gltf(d, png=png)[0])        # imports; the floor is simply untextured
        assert scene.counts()["meshes"] == 2
        orc.render(scene.view(32, 24))
        assert orc.stats()["shadow"] > 0


def test_mutated_documents_never_crash_the_importer(tmp_path):
    """A deterministic mutation run over the four containers (flip / delete / insert bytes, half of them inside the binary chunk): every
    document either loads or is rejected with an error.  (6000 mutations under ASan + UBSan were clean in round 1; round 6, after the
    JPEG / animation / OBJ importers had been added: tools/probes/fuzz_importers — 1.98 M byte-level and 72 000 JSON-level mutations, one finding, fixed:
    test_malformed_documents_are_errors_not_crashes cases f - i.)"""
    import struct
    from gltf_util import write_skinned_gltf, write_textured_gltf
    rng = np.random.default_rng(7)
    seeds = []
    for sub, fn in (("a", lambda p: write_gltf(p, "glb")), ("b", lambda p: write_textured_gltf(p, True)[0]),
                    ("c", lambda p: write_skinned_gltf(p)[0]), ("e", lambda p: write_gltf(p, "base64"))):
        (tmp_path / sub).mkdir()
        seeds.append((tmp_path / sub / "x").parent.joinpath(fn(tmp_path / sub).name).read_bytes())
    loaded = rejected = 0
    for it in range(300):
        raw = bytearray(seeds[it % 4])
        for _ in range(int(rng.integers(1, 6))):
            lo = 0
            if it % 2 == 0 and raw[:4] == b"glTF" and len(raw) > 20:
                lo = min(20 + struct.unpack("<I", bytes(raw[12:16]))[0] + 8, len(raw) - 1)
            pos = int(rng.integers(lo, len(raw)))
            mode = int(rng.integers(0, 3))
            if mode == 0:
                raw[pos] = int(rng.integers(0, 256))
            elif mode == 1:
                del raw[pos:pos + int(rng.integers(1, 64))]
            else:
                raw[pos:pos] = bytes(rng.integers(0, 256, int(rng.integers(1, 16)), dtype=np.uint8))
            if len(raw) < 2:
                break
        p = tmp_path / ("f.glb" if it % 4 < 2 else "f.gltf")
        p.write_bytes(bytes(raw))
        try:
            Scene().load_gltf(str(p))
            loaded += 1
        except ValueError:
            rejected += 1
    assert loaded + rejected == 300 and rejected > 100


@pytest.mark.parametrize("kind,a,b", [("cornell", 0, 0), ("soup", 600, 3), ("atrium", 20000, 0)])
def test_save_glb_round_trip_is_lossless(tmp_path, kind, a, b):
    """host/gltf_export.cpp writes the procedural scenes as .glb; the importer reads back the SAME scene: triangles bit for bit, the
    materials (Disney parameters through extras), area + punctual lights, the camera — so the oracle renders the same image from both."""
    src = Scene().build(kind, a, b, 0.0, 7)
    src.set_aspect(48 / 32)
    path = src.save_glb(str(tmp_path / "scene.glb"))
    raw = open(path, "rb").read()
    assert raw[:4] == b"glTF" and int.from_bytes(raw[8:12], "little") == len(raw)
    back = Scene().load_gltf(path)
    back.set_aspect(48 / 32)
    assert back.counts() == src.counts() and back.triangle_count == src.triangle_count
    va, vb = src.view(48, 32), back.view(48, 32)
    fa, fb = np.frombuffer(bytes(va), np.float32), np.frombuffer(bytes(vb), np.float32)
    assert np.array_equal(fa, fb)                                         # camera position, frame, field of view (0.0 == -0.0)
    oa, ob = Oracle(48, 32, threads=4, max_path_length=3), Oracle(48, 32, threads=4, max_path_length=3)
    src.sync(oa); back.sync(ob)
    assert np.array_equal(oa.triangles().view(np.uint32), ob.triangles().view(np.uint32))
    oa.render(va); ob.render(vb)
    assert oa.stats()["shadow"] > 50
    assert np.array_equal(oa.accumulator().view(np.uint32), ob.accumulator().view(np.uint32))


# ---------------------------------------------------------------- animations (glTF 3.11) and JPEG images: what examples/animated loads
def _jpeg(img, **kw):
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def _pil_decode(data, mode="RGB"):
    import io
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert(mode))


def _test_image(h=97, w=131):
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([128 + 100 * np.sin(xx / 9.0) * np.cos(yy / 7.0), 128 + 90 * np.cos(xx / 5.0 + yy / 11.0), xx * 255.0 / (w - 1)], -1).clip(0, 255).astype(np.uint8)


@pytest.mark.parametrize("subsampling,restart", [(0, 0), (1, 0), (2, 0), (0, 3), (2, 5)])
def test_jpeg_decoder_agrees_with_pillow(subsampling, restart):
    """host/jpeg.cpp (baseline sequential JPEG) against libjpeg through Pillow: 4:4:4 / 4:2:2 / 4:2:0, odd sizes (partial MCUs), restart
    intervals.  The inverse DCT is evaluated in double precision here and in fixed point there: a level or two after the colour
    conversion, nothing systematic."""
    pytest.importorskip("PIL")
    from rfw_rs_amd.scene import decode_image
    img = _test_image()
    kw = {"quality": 90, "subsampling": subsampling}
    if restart:
        kw["restart_marker_blocks"] = restart
    data = _jpeg(img, **kw)
    mine, ref = decode_image(data), _pil_decode(data)
    assert mine.shape == (97, 131, 4) and (mine[..., 3] == 255).all()
    diff = np.abs(mine[..., :3].astype(int) - ref.astype(int))
    assert diff.max() <= 3 and diff.mean() < 0.1


@pytest.mark.parametrize("subsampling,restart,quality", [(0, 0, 90), (2, 0, 50), (1, 4, 75), (2, 3, 95)])
def test_progressive_jpeg_agrees_with_pillow(subsampling, restart, quality):
    """SOF2 files: DC and AC scans, first passes and refinements (spectral selection + successive approximation), end-of-band runs,
    restart intervals inside progressive scans; a noisy image so that the refinement scans carry many correction bits."""
    pytest.importorskip("PIL")
    from rfw_rs_amd.scene import decode_image
    rng = np.random.default_rng(0)
    img = (_test_image().astype(int) + rng.integers(-20, 20, (97, 131, 3))).clip(0, 255).astype(np.uint8)
    kw = {"quality": quality, "subsampling": subsampling, "progressive": True}
    if restart:
        kw["restart_marker_blocks"] = restart
    data = _jpeg(img, **kw)
    assert b"\xff\xc2" in data
    diff = np.abs(decode_image(data)[..., :3].astype(int) - _pil_decode(data).astype(int))
    assert diff.max() <= 3 and diff.mean() < 0.1
    grey = _jpeg(img[..., 1], quality=80, progressive=True)
    assert np.abs(decode_image(grey)[..., 0].astype(int) - _pil_decode(grey, "L")).max() <= 1
    rng = np.random.default_rng(6)
    for it in range(100):                                                      # damaged progressive files: decode or refuse
        raw = bytearray(data)
        for _ in range(int(rng.integers(1, 5))):
            raw[int(rng.integers(2, len(raw)))] = int(rng.integers(0, 256))
        try:
            decode_image(bytes(raw))
        except ValueError:
            pass


def test_jpeg_grey_optimised_tables_and_refusals():
    pytest.importorskip("PIL")
    from rfw_rs_amd.scene import decode_image
    img = _test_image(40, 56)
    grey = _jpeg(img[..., 0], quality=85)
    got = decode_image(grey)
    assert np.abs(got[..., 0].astype(int) - _pil_decode(grey, "L")).max() <= 1 and np.array_equal(got[..., 0], got[..., 2])
    opt = _jpeg(img, quality=75, optimize=True)                    # per-image Huffman tables
    assert np.abs(decode_image(opt)[..., :3].astype(int) - _pil_decode(opt).astype(int)).max() <= 3
    good = _jpeg(img, quality=90)
    with pytest.raises(ValueError, match="unsupported coding process"):
        decode_image(good.replace(b"\xff\xc0", b"\xff\xc9", 1))      # the frame header of an arithmetic-coded file
    for name, bad in {"truncated": good[: len(good) // 2], "no tables": good[:2] + good[good.index(b"\xff\xc0"):], "garbage": b"\xff\xd8" + bytes(range(256))}.items():
        with pytest.raises(ValueError):
            decode_image(bad)
    # deterministic mutations of a good file: decode or refuse, never crash
    rng = np.random.default_rng(5)
    for it in range(200):
        raw = bytearray(good)
        for _ in range(int(rng.integers(1, 5))):
            raw[int(rng.integers(2, len(raw)))] = int(rng.integers(0, 256))
        try:
            decode_image(bytes(raw))
        except ValueError:
            pass


def test_animation_sampling_matches_the_specification(tmp_path):
    """The importer keeps the node graph; set_animation_time evaluates the channels (LINEAR incl. shortest-arc slerp over normalised int16
    keys, STEP, CUBICSPLINE), loops the time, and rebuilds instance and joint matrices.  Held against tests/gltf_anim.py, an independent
    float64 evaluation of the same document written from the glTF specification."""
    from gltf_anim import evaluate
    from gltf_util import write_animated_gltf
    path = write_animated_gltf(tmp_path)
    scene = Scene().load_gltf(str(path))
    assert scene.animation_info(0) == {"duration": 2.0, "channels": 5}       # the weights channel is dropped
    with pytest.raises(KeyError):
        scene.animation_info(1)
    load_pose = scene.skin_matrices(0).copy()
    world, skins = evaluate(str(path), 0.0)
    for t in (0.0, 0.1, 0.25, 0.6, 0.99, 1.0, 1.3, 1.75, 1.9, 2.0, 2.6, -0.4, 41.3):
        assert scene.set_animation_time(t) == 1
        world, skins = evaluate(str(path), t)
        assert np.allclose(scene.skin_matrices(0), skins[0], rtol=0, atol=2e-7), t
        m, skin = scene.instance_matrix(1, 0)                                # the animated cube (node 5 under "arm")
        assert skin == -1 and np.allclose(m, world[5], rtol=0, atol=3e-7), t
        m, _ = scene.instance_matrix(1, 1)                                   # the matrix node never moves
        assert np.allclose(m, world[6], rtol=0, atol=1e-7)
        m, skin = scene.instance_matrix(0, 0)                                # the skinned instance: identity, the joints carry the transform
        assert skin == 0 and np.array_equal(m, np.eye(4, dtype=np.float32))
    scene.set_animation_time(0.6)
    assert np.abs(scene.skin_matrices(0) - load_pose).max() > 0.1
    # the whole graph under a transform (GraphHandle::get_transform): joints and instances follow
    scene.set_graph_transform(-1, translation=(1.0, 0.5, -2.0), rotation=(0.0, np.sin(0.3), 0.0, np.cos(0.3)), scale=(2.0, 2.0, 2.0))
    world, skins = evaluate(str(path), 0.6)
    c, s = np.cos(0.6), np.sin(0.6)
    root = np.array([[2 * c, 0, 2 * s, 1.0], [0, 2, 0, 0.5], [-2 * s, 0, 2 * c, -2.0], [0, 0, 0, 1]])
    assert np.allclose(scene.skin_matrices(0), root @ skins[0], atol=1e-6)
    assert np.allclose(scene.instance_matrix(1, 0)[0], root @ world[5], atol=1e-6)


def test_a_graph_instantiated_twice_shares_meshes_not_skins(tmp_path):
    """examples/animated/src/main.rs:84-103 adds the CesiumMan descriptor to the scene twice, each with its own transform: the second graph
    reuses the meshes, gets its own instances and its own skin (the backend skins a copy per (mesh, skin) pair), and follows the clock."""
    from gltf_anim import evaluate
    from gltf_util import write_animated_gltf
    path = write_animated_gltf(tmp_path)
    scene = Scene().load_gltf(str(path))
    before = scene.counts()
    g = scene.instantiate_graph()
    after = scene.counts()
    assert g == 1 and after["meshes"] == before["meshes"] and after["instances"] == 2 * before["instances"] and after["area_lights"] == 2 * before["area_lights"]
    scene.set_graph_transform(g, translation=(2.0, 0.0, 0.0))
    scene.set_animation_time(0.6)
    world, skins = evaluate(str(path), 0.6)
    shift = np.eye(4); shift[0, 3] = 2.0
    assert np.allclose(scene.skin_matrices(0), skins[0], atol=2e-7) and np.allclose(scene.skin_matrices(1), shift @ skins[0], atol=3e-7)
    assert scene.instance_matrix(0, 0)[1] == 0 and scene.instance_matrix(0, 1)[1] == 1          # two instances of the tube, two skins
    assert np.allclose(scene.instance_matrix(1, 2)[0], shift @ world[5], atol=3e-7)             # the second graph's animated cube
    orc = Oracle(60, 40, threads=4, max_path_length=2)
    scene.set_aspect(1.5)
    scene.sync(orc)
    assert orc.stats()["n_tris"] == (128 + 12 + 2 + 2) + 2 * 128 and orc.validate_bvh() == 0   # the four meshes once + one skinned copy of the tube per skin
    with pytest.raises(KeyError):
        scene.instantiate_graph(7)


def test_animated_scene_renders_differently_over_time(tmp_path):
    """set_animation_time -> sync -> render on the oracle: skinned triangles and moved instances reach the image; the JPEG base colour
    of the floor reaches the sampler."""
    pytest.importorskip("PIL")
    from gltf_util import write_animated_gltf
    tex = _test_image(64, 64)
    data = _jpeg(tex, quality=95, subsampling=0)
    scene = Scene().load_gltf(str(write_animated_gltf(tmp_path, jpeg=data)))
    orc = Oracle(60, 40, threads=4, max_path_length=2)
    scene.set_aspect(1.5)
    view = scene.view(60, 40)
    images = []
    for t in (0.0, 0.6, 1.3):
        scene.set_animation_time(t)
        scene.sync(orc)
        orc.reset(); orc.render(view)
        images.append(orc.accumulator().copy())
        assert orc.validate_bvh() == 0 and orc.stats()["shadow"] > 0
    assert not np.array_equal(images[0], images[1]) and not np.array_equal(images[1], images[2])
    scene.set_animation_time(2.0)                                            # one full loop later: the first image again, bit for bit
    scene.sync(orc); orc.reset(); orc.render(view)
    assert np.array_equal(orc.accumulator().view(np.uint32), images[0].view(np.uint32))
    want = _pil_decode(data)[10, 20].astype(np.float32) / 255.0              # texel (x = 20, y = 10) of the decoded JPEG, within the decoders' tolerance
    got = orc.sample_texture(0, (20 + 0.5) / 64, (10 + 0.5) / 64, 0.0)
    assert np.abs(got[:3] - want).max() <= 3.5 / 255.0


REF_MODELS = "/root/reference/assets/models"


@pytest.mark.skipif(not __import__("os").path.exists(REF_MODELS + "/CesiumMan/CesiumMan.gltf"), reason="the reference checkout is not on this machine")
def test_reference_sample_assets_import_in_place():
    """The two models the reference's examples/animated loads (examples/animated/src/main.rs:80,108), read where they lie: CesiumMan
    (skinned, animated, JPEG texture) and pica (170 nodes, PNG textures).  Counts are checked against the documents' own JSON, the
    animated joints against the independent evaluation, the texture against Pillow."""
    import json
    from gltf_anim import evaluate
    man = REF_MODELS + "/CesiumMan/CesiumMan.gltf"
    doc = json.load(open(man))
    scene = Scene().load_gltf(man)
    acc = doc["accessors"]
    tris = sum(acc[p["indices"]]["count"] // 3 for m in doc["meshes"] for p in m["primitives"])
    assert scene.triangle_count == tris == 4672 and scene.counts()["materials"] == len(doc["materials"])
    info = scene.animation_info(0)
    assert info["channels"] == len(doc["animations"][0]["channels"]) == 57 and abs(info["duration"] - 2.0) < 1e-6
    assert len(scene.skin_matrices(0)) == len(doc["skins"][0]["joints"]) == 19
    for t in (0.0, 0.37, 1.1, 1.99, 2.5):
        scene.set_animation_time(t)
        _, skins = evaluate(man, t)
        assert np.allclose(scene.skin_matrices(0), skins[0], rtol=0, atol=2e-7), t
    orc = Oracle(48, 64, threads=4, max_path_length=1)
    scene.sync(orc)
    pytest.importorskip("PIL")
    ref = _pil_decode(open(REF_MODELS + "/CesiumMan/CesiumMan.jpg", "rb").read())   # 1024 x 1024: one layer of the texture array as it is
    for (x, y) in ((100, 200), (512, 512), (900, 40)):
        got = orc.sample_texture(0, (x + 0.5) / 1024, (y + 0.5) / 1024, 0.0)
        assert np.abs(got[:3] - ref[y, x] / 255.0).max() <= 3.5 / 255.0
    pica = REF_MODELS + "/pica/scene.gltf"
    doc = json.load(open(pica))
    scene = Scene().load_gltf(pica)
    with_mesh = [n for n in doc["nodes"] if "mesh" in n]
    tris = 0
    for n in with_mesh:
        for p in doc["meshes"][n["mesh"]]["primitives"]:
            tris += (doc["accessors"][p["indices"]]["count"] if "indices" in p else doc["accessors"][p["attributes"]["POSITION"]]["count"]) // 3
    c = scene.counts()
    assert c["instances"] == len(with_mesh) and scene.triangle_count == tris
    assert c["materials"] >= len(doc["materials"])


@pytest.mark.parametrize("interlace", [False, True])
def test_png_every_colour_type_depth_and_interlace(interlace):
    """host/gltf.cpp decode_png over the whole matrix of the PNG specification — grey 1/2/4/8/16, RGB 8/16, palette 1/2/4/8 (+ tRNS),
    grey + alpha 8/16, RGBA 8/16, colour keys, Adam7 — on odd sizes (passes with empty rows / columns): against the source samples, and
    the writer used here against Pillow where Pillow reads the format."""
    from gltf_util import encode_png_general
    from rfw_rs_amd.scene import decode_image
    rng = np.random.default_rng(21)
    for (h, w) in ((1, 1), (3, 5), (9, 13), (16, 8)):
        for colour, channels, depths in ((0, 1, (1, 2, 4, 8, 16)), (2, 3, (8, 16)), (3, 1, (1, 2, 4, 8)), (4, 2, (8, 16)), (6, 4, (8, 16))):
            for depth in depths:
                smp = rng.integers(0, 1 << depth, (h, w, channels))
                palette = rng.integers(0, 256, (1 << depth, 3)).astype(np.uint8) if colour == 3 else None
                trns = None
                if colour == 3:
                    trns = bytes(rng.integers(0, 256, max(1, (1 << depth) // 2)).astype(np.uint8))
                elif colour == 0:
                    trns = int(smp[0, 0, 0]).to_bytes(2, "big")
                elif colour == 2:
                    trns = b"".join(int(v).to_bytes(2, "big") for v in smp[0, 0])
                data = encode_png_general(smp, colour, depth, interlace, palette, trns)
                got = decode_image(data)
                to8 = (lambda v: v >> 8) if depth == 16 else ((lambda v: v) if depth == 8 else (lambda v: v * 255 // ((1 << depth) - 1)))
                want = np.zeros((h, w, 4), np.int64)
                if colour == 3:
                    want[..., :3] = palette[smp[..., 0]]
                    ta = np.frombuffer(trns, np.uint8)
                    want[..., 3] = np.where(smp[..., 0] < len(ta), ta[np.minimum(smp[..., 0], len(ta) - 1)], 255)
                elif colour in (0, 4):
                    want[..., :3] = to8(smp[..., :1])
                    want[..., 3] = to8(smp[..., 1]) if colour == 4 else np.where(smp[..., 0] == smp[0, 0, 0], 0, 255)
                else:
                    want[..., :3] = to8(smp[..., :3])
                    want[..., 3] = to8(smp[..., 3]) if colour == 6 else np.where((smp == smp[0, 0]).all(axis=2), 0, 255)
                assert np.array_equal(got, want.astype(np.uint8)), (h, w, colour, depth, interlace)
    try:                                                     # the writer itself, checked by an independent reader
        import io
        from PIL import Image
        smp = rng.integers(0, 256, (9, 13, 4))
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(encode_png_general(smp, 6, 8, interlace))).convert("RGBA")), smp.astype(np.uint8))
        idx = rng.integers(0, 4, (9, 13, 1)); pal = rng.integers(0, 256, (4, 3)).astype(np.uint8)
        assert np.array_equal(np.asarray(Image.open(io.BytesIO(encode_png_general(idx, 3, 2, interlace, pal))).convert("RGB")), pal[idx[..., 0]])
    except ImportError:
        pass
