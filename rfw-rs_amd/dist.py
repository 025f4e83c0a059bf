"""Multi-GPU plumbing: primary rays shard by image tile, one process per GPU, and ONE all-gather
of the per-rank accumulator slabs per frame (RCCL over xGMI through torch.distributed; gloo in the CPU tests).
No other exchange: the scene is replicated, the BVH build is deterministic, pixels are independent.

The tile -> rank dealing and the slab layout are defined by the kernels (kernels.hip slab_to_pixel /
pixel_to_slab); `slab_index_map` restates them in numpy so that host code and tests can assemble gathered slabs."""
import numpy as np


def shard_geometry(width, height, world, tile_size=64, streams=1):
    """`streams` = sub-shards per rank (rfw_hip_options.streams): tiles are dealt round-robin to world * streams virtual ranks,
    virtual rank r * streams + s being sub-shard s of rank r; a rank's slab is its sub-slabs back to back."""
    tiles_x = (width + tile_size - 1) // tile_size
    tiles_y = (height + tile_size - 1) // tile_size
    total = tiles_x * tiles_y
    wv = world * streams
    local_v = (total + wv - 1) // wv
    return {"tiles_x": tiles_x, "tiles_y": tiles_y, "tiles_total": total, "tiles_local": local_v * streams,
            "sub_elems": local_v * tile_size * tile_size, "slab_elems": local_v * streams * tile_size * tile_size}


def slab_index_map(width, height, world, tile_size=64, streams=1):
    """owner[y, x] = rank that renders pixel (x, y); slot[y, x] = its index in that rank's slab."""
    ys, xs = np.mgrid[0:height, 0:width]
    g = shard_geometry(width, height, world, tile_size, streams)
    tx, ty = xs // tile_size, ys // tile_size
    tile = ty * g["tiles_x"] + tx
    wv = world * streams
    vowner = tile % wv
    owner, sub = vowner // streams, vowner % streams
    lt = tile // wv
    ix, iy = xs - tx * tile_size, ys - ty * tile_size
    bpr = tile_size >> 3
    if bpr <= 16 and (bpr & (bpr - 1)) == 0:   # the 8x8 blocks of a tile follow a Z curve (kernels.hip: morton_tile)
        def spread(v):
            return (v & 1) | ((v & 2) << 1) | ((v & 4) << 2) | ((v & 8) << 3)
        block = spread(ix >> 3) | (spread(iy >> 3) << 1)
    else:
        block = (iy >> 3) * bpr + (ix >> 3)
    slot = sub * g["sub_elems"] + lt * tile_size * tile_size + block * 64 + ((iy & 7) << 3) + (ix & 7)
    return owner.astype(np.int64), slot.astype(np.int64)


def extract_slab(frame, rank, world, tile_size=64, streams=1):
    """The slab rank `rank` would produce for a full frame (H, W, C): used by the CPU tests as a stand-in renderer."""
    h, w, c = frame.shape
    owner, slot = slab_index_map(w, h, world, tile_size, streams)
    slab = np.zeros((shard_geometry(w, h, world, tile_size, streams)["slab_elems"], c), dtype=frame.dtype)
    m = owner == rank
    slab[slot[m]] = frame[m]
    return slab


def assemble(gathered, width, height, tile_size=64, streams=1):
    """gathered: (world, slab_elems, C) -> frame (H, W, C); the numpy twin of the k_assemble kernel."""
    world = gathered.shape[0]
    owner, slot = slab_index_map(width, height, world, tile_size, streams)
    return gathered[owner, slot]


def all_gather_slabs(local_slab, group=None):
    """One collective per frame: every rank contributes its slab, receives all.  `local_slab` is a torch tensor (any device);
    returns (world, *local_slab.shape).  With backend "nccl" on ROCm this is RCCL over xGMI."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    out = torch.empty((world,) + tuple(local_slab.shape), dtype=local_slab.dtype, device=local_slab.device)
    dist.all_gather_into_tensor(out.view(-1), local_slab.contiguous().view(-1), group=group)
    return out
