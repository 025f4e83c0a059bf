import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, time
import torch
from oracle.bindings import Oracle
from rfw_rs_amd import HipBackend, Scene
w,h=480,270
scene = Scene().build("atrium", 1048576, 0, 0.0, 0xC0FFEE); scene.set_aspect(w/h)
view = scene.view(w,h)
orc = Oracle(w,h,threads=64,max_path_length=1); scene.sync(orc); orc.render(view); print("oracle", orc.stats())
for trial in range(3):
    be = HipBackend.init(w,h,1.0,max_path_length=1)
    scene.mark_all_changed(); scene.sync(be)
    be.render(view); s=be.frame_stats(); print("gpu", trial, s["primary_rays"], s["shadow_rays"], s["ms_trace_primary"], s["ms_trace_shadow"], s["ms_shade"])
    a=be.accumulator(); b=orc.accumulator()
    print("  equal bits:", np.array_equal(a.view(np.uint32), b.view(np.uint32)), "diff px", int((a!=b).any(axis=2).sum()))
    be.close()
