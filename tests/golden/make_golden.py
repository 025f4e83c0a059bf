#!/usr/bin/env python3
"""Generates the committed golden vectors with the CPU oracle (run from the repo root: python tests/golden/make_golden.py).
Each case = a synthetic scene of the host library + camera + options -> accumulator (RGBA32F sums) and primary hit records.
The reference has no fixtures for this path; these pin the oracle against regressions and give the GPU
tests an oracle-free target."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

CASES = {
    # name: (kind, a, b, seed, width, height, max_path_length, spp[, seed of the synthetic blue-noise tables])
    "cornell_primary_shadow": ("cornell", 0, 0, 1, 64, 64, 1, 1),
    "cornell_path3_4spp": ("cornell", 0, 0, 1, 64, 64, 3, 4),
    "soup_instanced_path3_2spp": ("soup", 900, 5, 11, 64, 48, 3, 2),
    "gallery_textured_path3_2spp": ("gallery", 0, 0, 5, 96, 64, 3, 2),
    "skinned_pose07_path3_2spp": ("skinned", 0, 0, 3, 96, 64, 3, 2),
    # the blue-noise sampler of the first 256 samples, with seeded tables in the layout of gpu_rt::blue_noise::create_blue_noise_buffer()
    "soup_blue_noise_path3_3spp": ("soup", 900, 5, 11, 160, 136, 3, 3, 7),
}


def blue_noise_table(seed):
    return np.random.default_rng(seed).integers(0, 256, 5 * 65536).astype(np.uint32)


def run_case(case):
    from oracle.bindings import Oracle
    from rfw_rs_amd import Scene
    kind, a, b, seed, w, h, mpl, spp = case[:8]
    scene = Scene().build(kind, a, b, 0.0, seed)
    scene.set_aspect(w / h)
    view = scene.view(w, h)
    orc = Oracle(w, h, threads=4, max_path_length=mpl)
    if len(case) > 8:
        orc.set_blue_noise(blue_noise_table(case[8]))
    scene.sync(orc)
    for _ in range(spp):
        orc.render(view)
    o, d = orc.primary_rays(view, 0)
    hits = orc.intersect(o, d)
    return orc.accumulator(), hits, orc.stats()


def main():
    out = os.path.dirname(os.path.abspath(__file__))
    only_missing = "--missing" in sys.argv
    for name, case in CASES.items():
        if only_missing and os.path.exists(os.path.join(out, name + ".npz")):
            continue
        acc, hits, st = run_case(case)
        np.savez_compressed(os.path.join(out, name + ".npz"), acc=acc, hit_inst=hits["inst"], hit_tri=hits["tri"], hit_t=hits["t"],
                            hit_u=hits["u"], hit_v=hits["v"], rays=np.array([st["primary"], st["extension"], st["shadow"]], dtype=np.int64))
        print(name, acc.shape, "mean", acc[..., :3].mean(), st["primary"], st["extension"], st["shadow"])


if __name__ == "__main__":
    main()
