"""Three ranks of ONE process exchanging tiles through the library's loop-back hub (rfw_hip_comm_init_loopback): run by
tests/test_gpu_parity.py::test_library_exchange_through_the_loopback_hub in a process of its own (GPU_MAX_HW_QUEUES is read once).
    python3 tests/loopback_ranks.py <gather format 0|1|2> <present rank, -1 = every rank de-tiles>
What the ranks end up with is the single-GPU frame, bit for bit — after every frame, and again after six frames in flight over three frame
slots per rank (the slots' gathers chained by the same events as with ncclAllGather).  Then the failure only N > 1 has: a rank that never
renders its frame is reported by the others' reads."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
fmt, present_rank = int(sys.argv[1]), int(sys.argv[2])


class raises:  # (pytest.raises without pytest)
    def __init__(self, exc, match=None):
        self.exc, self.match = exc, match

    def __enter__(self):
        return self

    def __exit__(self, t, v, tb):
        assert t is not None and issubclass(t, self.exc), f"expected {self.exc.__name__}, got {t}"
        assert self.match is None or self.match in str(v), (self.match, str(v))
        return True


from rfw_rs_amd import BackendError, HipBackend, Scene
w, h, world = 200, 136, 3
scene = Scene().build("soup", 1500, 5, 0.0, 4)
scene.set_aspect(w / h)
base = scene.view(w, h)
from rfw_rs_amd import pod
views = []
for k in range(6):                                       # six different images: the camera slides sideways
    v = pod.CameraView3D.from_buffer_copy(base)
    v.pos.x += 0.07 * k; v.p1.x += 0.07 * k
    views.append(v)
full = HipBackend.init(w, h, 1.0)
scene.sync(full)
ref = []
for v in views:
    full.reset_accumulation()
    full.render(v)
    pres = full.host_frame(presented=True)
    full.download_frame(pres); full.wait_downloads()
    ref.append((full.accumulator().copy(), full.framebuffer().copy(), pres.copy()))
ranks = []
key = 0x10C0 + fmt * 16 + (present_rank + 1)
for r in range(world):
    be = HipBackend.init(w, h, 1.0, rank=r, world=world, tile_size=32, frames_in_flight=3)
    be.set_option("gather_format", fmt)
    be.set_option("present_rank", present_rank)
    be.set_option("p2p_timeout_ms", 5000)
    scene.mark_all_changed()
    scene.sync(be)
    be.comm_init_loopback(key, r, world)
    ranks.append(be)

def check(i, what):
    acc, frame, pres = ref[i]
    for r, be in enumerate(ranks):
        # (an all-gather leaves every rank's tiles on every rank: the presenting rank de-tiles at once, the others when somebody reads)
        if fmt == 0:
            assert np.array_equal(be.accumulator().view(np.uint32), acc.view(np.uint32)), (what, i, r)
        elif fmt == 1:
            want = frame[..., :3].astype(np.float16).astype(np.float32)
            assert np.array_equal(be.framebuffer()[..., :3].view(np.uint32), want.view(np.uint32)), (what, i, r)
        else:
            dst = be.host_frame(presented=True)
            be.download_frame(dst); be.wait_downloads()
            assert np.array_equal(dst, pres), (what, i, r)

for i, v in enumerate(views):                          # one frame at a time
    for be in ranks:
        be.render(v)
    check(i, "frame by frame")
for rep in range(3):                                   # six frames in flight over three slots per rank, ranks issued in changing order
    for i, v in enumerate(views):
        for be in (ranks if (i + rep) % 2 == 0 else ranks[::-1]):
            be.render(v)
    check(len(views) - 1, "six frames in flight")
# a rank that never arrives
for be in ranks:
    be.set_option("p2p_timeout_ms", 300)
ranks[0].render(views[0])
ranks[1].render(views[0])
reader = ranks[present_rank if present_rank >= 0 else 0]
with raises(BackendError, match="did not arrive"):
    reader.device_synchronize()
    reader.framebuffer() if fmt != 2 else reader.download_frame(reader.host_frame(presented=True))
    reader.wait_downloads()
for be in ranks + [full]:
    be.close()

print("LOOPBACK OK", fmt, present_rank)
