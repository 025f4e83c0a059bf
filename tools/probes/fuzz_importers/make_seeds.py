"""Seed documents for the importer fuzzers, from the tests' own writers -> /tmp/fuzz/seeds (one directory per document and its files)."""
import sys, os, pathlib, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from gltf_util import write_gltf, write_skinned_gltf, write_textured_gltf
import test_obj, test_gltf
from gltf_util import write_animated_gltf
out = pathlib.Path("/tmp/fuzz/seeds")
def sub(n):
    p = out / n; shutil.rmtree(p, ignore_errors=True); p.mkdir(parents=True); return p
print(write_gltf(sub("glb"), "glb"))
print(write_gltf(sub("b64"), "base64"))
print(write_gltf(sub("ext")))
print(write_textured_gltf(sub("tex_glb"), True)[0])
print(write_textured_gltf(sub("tex"), False)[0])
print(write_skinned_gltf(sub("skin"))[0])
img = test_gltf._test_image()
print(write_animated_gltf(sub("anim")))
print(write_animated_gltf(sub("anim_jpeg"), jpeg=test_gltf._jpeg(img, quality=85)))
p = sub("obj"); test_obj.write_scene(p); print(list(p.iterdir()))
# raw image streams
(sub("img") / "a.jpg").write_bytes(test_gltf._jpeg(img, quality=85))
(out / "img" / "b.jpg").write_bytes(test_gltf._jpeg(img, quality=70, progressive=True))
(out / "img" / "c.jpg").write_bytes(test_gltf._jpeg(img, quality=90, subsampling=0, restart_marker_blocks=2))
