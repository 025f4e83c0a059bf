// lbvh.hip — device BVH build: scene bounds -> 30-bit Morton keys -> radix sort (hipCUB) -> Karras hierarchy ->
// bottom-up box fit (agent-scope release/acquire hand-off between the two children of a node) -> every even-depth
// internal node becomes one 4-wide Node4 whose children are its grandchildren.  One primitive per leaf.
#include "lbvh.h"
#include "env_switches.h"

#include <hipcub/hipcub.hpp>

namespace rfwhip {
namespace {

constexpr int kBlock = 256;
__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Layout {
    size_t bounds, keys_in, keys_out, vals_in, left, right, parent, flags, nbox, flag4, idx4, cub, total;
    size_t cub_bytes;
};

Layout make_layout(uint32_t n, size_t cub_bytes)
{
    Layout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t m = n > 0 ? n : 1;
    L.bounds = take(8 * sizeof(uint32_t));
    L.keys_in = take(m * 4);
    L.keys_out = take(m * 4);
    L.vals_in = take(m * 4);
    L.left = take(m * 4);
    L.right = take(m * 4);
    L.parent = take(2 * m * 4);
    L.flags = take(m * 4);
    L.nbox = take(2 * m * sizeof(DevBox));
    L.flag4 = take(m * 4);
    L.idx4 = take(m * 4);
    L.cub = take(cub_bytes);
    L.cub_bytes = cub_bytes;
    L.total = off;
    return L;
}

size_t cub_temp_bytes(uint32_t n)
{
    size_t a = 0, b = 0;
    const uint32_t m = n > 0 ? n : 1;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)m, 0, 30);
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, b, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)m);
    return (a > b ? a : b) + 256;
}

__device__ inline uint32_t f_order(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float f_unorder(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__global__ void k_init_bounds(uint32_t* bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0xffffffffu;      // min
    else if (threadIdx.x < 6) bounds[threadIdx.x] = 0u;          // max
}

__global__ void k_scene_bounds(const DevBox* __restrict__ boxes, uint32_t n, uint32_t* bounds)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        for (int a = 0; a < 3; a++) {
            const float c = 0.5f * (boxes[i].lo[a] + boxes[i].hi[a]);
            lo[a] = c;
            hi[a] = c;
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off));
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) {
            atomicMin(&bounds[a], f_order(lo[a]));
            atomicMax(&bounds[3 + a], f_order(hi[a]));
        }
}

__device__ inline uint32_t expand10(uint32_t v)
{
    v = (v * 0x00010001u) & 0xFF0000FFu;
    v = (v * 0x00000101u) & 0x0F00F00Fu;
    v = (v * 0x00000011u) & 0xC30C30C3u;
    v = (v * 0x00000005u) & 0x49249249u;
    return v;
}

// 30-bit Morton key of a box's centre inside the bounds of all centres (lo / hi per axis)
__device__ inline uint32_t morton_key_of_centre(const float* centre, const float* lo3, const float* hi3)
{
    uint32_t q[3];
    for (int a = 0; a < 3; a++) {
        const float lo = lo3[a], hi = hi3[a];
        const float c = centre[a];
        const float ext = hi - lo;
        float t = ext > 0.0f ? (c - lo) / ext : 0.0f;
        t = fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
        q[a] = (uint32_t)t;
    }
    return (expand10(q[0]) << 2) | (expand10(q[1]) << 1) | expand10(q[2]);
}
__device__ inline uint32_t morton_key(const DevBox& box, const float* lo3, const float* hi3)
{
    const float c[3] = {0.5f * (box.lo[0] + box.hi[0]), 0.5f * (box.lo[1] + box.hi[1]), 0.5f * (box.lo[2] + box.hi[2])};
    return morton_key_of_centre(c, lo3, hi3);
}
__global__ void k_morton(const DevBox* __restrict__ boxes, uint32_t n, const uint32_t* __restrict__ bounds, uint32_t* keys, uint32_t* vals)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    float lo[3], hi[3];
    for (int a = 0; a < 3; a++) { lo[a] = f_unorder(bounds[a]); hi[a] = f_unorder(bounds[3 + a]); }
    keys[i] = morton_key(boxes[i], lo, hi);
    vals[i] = i;
}

// longest common prefix of keys i and j (ties broken by index), -1 outside the array.  `Keys`: anything indexable that yields the sorted 30-bit
// keys — the sorted array in memory (k_hierarchy) or the high words of the (key, index) pairs the fused TLAS build sorts in LDS
template <class Keys> __device__ inline int delta(const Keys& keys, int n, int i, int j)
{
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    if (a == b) return 32 + __clz((uint32_t)i ^ (uint32_t)j);
    return __clz(a ^ b);
}

// child encoding in left/right: >= 0 internal node index, < 0 leaf ~index
template <class Keys> __device__ inline void hierarchy_node(const Keys& keys, const int ni, const int i, int32_t* left, int32_t* right, uint32_t* parent)
{
    const int d = (delta(keys, ni, i, i + 1) - delta(keys, ni, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, ni, i, i - d);
    int lmax = 2;
    while (delta(keys, ni, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1)
        if (delta(keys, ni, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, ni, i, j);
    int s = 0;
    int t = l;
    do {
        t = (t + 1) >> 1;
        if (delta(keys, ni, i, i + (s + t) * d) > dnode) s += t;
    } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? -1 : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const bool left_leaf = lo == gamma, right_leaf = hi == gamma + 1;
    left[i] = left_leaf ? ~gamma : gamma;
    right[i] = right_leaf ? ~(gamma + 1) : gamma + 1;
    parent[left_leaf ? (ni - 1 + gamma) : gamma] = (uint32_t)i;
    parent[right_leaf ? (ni - 1 + gamma + 1) : gamma + 1] = (uint32_t)i;
    if (i == 0) parent[0] = 0xffffffffu;
}
__global__ void k_hierarchy(const uint32_t* __restrict__ keys, uint32_t n, int32_t* left, int32_t* right, uint32_t* parent)
{
    const int i = (int)(blockIdx.x * kBlock + threadIdx.x);
    if (i >= (int)n - 1) return;
    hierarchy_node(keys, (int)n, i, left, right, parent);
}

__device__ inline uint32_t node_slot(int32_t child, uint32_t n) { return child < 0 ? (n - 1 + (uint32_t)(~child)) : (uint32_t)child; }

// bottom-up fit.  nbox[slot]: slot < n-1 internal node, slot >= n-1 leaf (n-1 + sorted position).
// Hand-off between the two children of a node without fences: a subtree's box is published with write-through (sc1) stores that are
// drained before the arrival counter is bumped, and the second arrival reads its sibling's box with sc1 loads (past the vector L1, which
// another CU's stores never refresh).  Round 2 fenced twice per step (__threadfence = L2 write-back + L1 invalidate, ~3.5 us, for every
// thread on every level): 12 ms for 1 M primitives.
// FENCED (environment RFW_LBVH_FENCED=1, read once): the textbook hand-off instead — plain stores, __threadfence(), the arrival counter,
// __threadfence(), plain loads — i.e. a release / acquire pair at device scope around the counter (an L2 write-back and an L1 invalidate per
// step: 9.4 instead of 4.8 ms for 1 M primitives).  The known-good fallback should the fence-free form ever fail
// rfw_hip_debug_lbvh_stress (tests/test_gpu_api.py::test_fence_free_tlas_fit_survives_ten_thousand_rebuilds) on some driver or firmware.
template <bool FENCED>
__device__ inline void fit_climb(DevBox b, const uint32_t i, const uint32_t n, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                 const uint32_t* __restrict__ parent, uint32_t* flags, DevBox* nbox)
{
    uint32_t me = n - 1 + i;
    uint32_t node = parent[me];
    for (;;) {
        if (FENCED) {
            for (int a = 0; a < 3; a++) { nbox[me].lo[a] = b.lo[a]; nbox[me].hi[a] = b.hi[a]; }
        } else {
            for (int a = 0; a < 3; a++) {
                __hip_atomic_store(&nbox[me].lo[a], b.lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&nbox[me].hi[a], b.hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        if (node == 0xffffffffu) return;
        if (FENCED) __threadfence();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the box is out before the arrival counts
        const uint32_t old = atomicAdd(&flags[node], 1u);
        if (old == 0u) return; // the second arrival at a node owns it
        if (FENCED) __threadfence();
        const uint32_t ls = node_slot(left[node], n), rs = node_slot(right[node], n);
        const uint32_t sib = ls == me ? rs : ls;
        if (FENCED) {
            const volatile DevBox* sb = nbox + sib;
            for (int a = 0; a < 3; a++) { b.lo[a] = fminf(b.lo[a], sb->lo[a]); b.hi[a] = fmaxf(b.hi[a], sb->hi[a]); }
        } else {
            for (int a = 0; a < 3; a++) {
                b.lo[a] = fminf(b.lo[a], __hip_atomic_load(&nbox[sib].lo[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                b.hi[a] = fmaxf(b.hi[a], __hip_atomic_load(&nbox[sib].hi[a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
        }
        me = node;
        node = parent[node];
    }
}
template <bool FENCED>
__global__ void k_fit(const DevBox* __restrict__ boxes, const uint32_t* __restrict__ order, uint32_t n, const int32_t* __restrict__ left,
                      const int32_t* __restrict__ right, const uint32_t* __restrict__ parent, uint32_t* flags, DevBox* nbox)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    fit_climb<FENCED>(boxes[order[i]], i, n, left, right, parent, flags, nbox);
}

// internal nodes at even depth become Node4s
// The same fit for ONE workgroup whose threads own several leaves each (k_tlas_fused).  One CU has ONE texture-address unit, and a
// wave instruction whose 64 lanes touch 64 different lines keeps it busy for a few hundred cycles: with k_fit's six 4-byte stores and six
// 4-byte loads per box the fit of 10 000 leaves took 0.3-0.4 ms of it (measured with cycle stamps per phase).  Here a box moves as two
// 16-byte accesses, the arrival counters are LDS atomics, and — all waves sharing one CU and its L1 — the hand-off needs no cache-bypassing
// accesses: a workgroup-scope release / acquire around the counter (the stores have left the wave before the counter is bumped).  A lane
// that has handed its box over takes its next leaf in the same trip of the loop instead of waiting for the slowest climb of its wavefront.
// Same boxes as k_fit's: min / max are exact and the order of arrival does not matter.
__device__ inline void fit_leaves_of_thread(const DevBox* __restrict__ boxes, const uint32_t* leaf_box, uint32_t first, const uint32_t step, const uint32_t n,
                                            const int32_t* __restrict__ left, const int32_t* __restrict__ right, const uint32_t* __restrict__ parent, uint32_t* arrivals, DevBox* nbox)
{
    uint32_t next = first, me = 0, node = 0xffffffffu;
    bool climbing = false;
    float4 blo = make_float4(0.0f, 0.0f, 0.0f, 0.0f), bhi = blo;
    for (;;) {
        if (!climbing && next < n) {
            const float4* src = reinterpret_cast<const float4*>(boxes + leaf_box[next]);
            blo = src[0]; bhi = src[1];
            me = n - 1 + next;
            node = parent[me];
            next += step;
            climbing = true;
        }
        if (__ballot(climbing) == 0ull) break; // (wave-uniform: every lane has published its last box)
        if (climbing) {
            float4* dst = reinterpret_cast<float4*>(nbox + me);
            dst[0] = blo; dst[1] = bhi;
            if (node == 0xffffffffu) climbing = false; // the root
            else {
                const uint32_t old = __hip_atomic_fetch_add(&arrivals[node], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (old == 0u) climbing = false; // the second arrival at a node owns it
                else {
                    const uint32_t ls = node_slot(left[node], n), rs = node_slot(right[node], n);
                    const float4* sib = reinterpret_cast<const float4*>(nbox + (ls == me ? rs : ls));
                    const float4 slo = sib[0], shi = sib[1];
                    blo = make_float4(fminf(blo.x, slo.x), fminf(blo.y, slo.y), fminf(blo.z, slo.z), 0.0f);
                    bhi = make_float4(fmaxf(bhi.x, shi.x), fmaxf(bhi.y, shi.y), fmaxf(bhi.z, shi.z), 0.0f);
                    me = node;
                    node = parent[node];
                }
            }
        }
    }
}

__device__ inline uint32_t even_depth(const uint32_t i, const uint32_t* __restrict__ parent)
{
    uint32_t depth = 0, p = parent[i];
    while (p != 0xffffffffu) {
        depth++;
        p = parent[p];
    }
    return (depth & 1u) ? 0u : 1u;
}
__global__ void k_flag_even_depth(uint32_t n_internal, const uint32_t* __restrict__ parent, uint32_t* flag4)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_internal) return;
    flag4[i] = even_depth(i, parent);
}

__device__ inline void emit4_node(const uint32_t i, const uint32_t n, const int32_t* __restrict__ left, const int32_t* __restrict__ right,
                                  const uint32_t* __restrict__ flag4, const uint32_t* __restrict__ idx4, const DevBox* __restrict__ nbox, Node4* __restrict__ nodes,
                                  uint32_t* node_count_out)
{
    if (i == n - 2 && node_count_out) *node_count_out = idx4[i] + flag4[i];
    if (!flag4[i]) return;
    int32_t kids[4];
    int nk = 0;
    const int32_t c2[2] = {left[i], right[i]};
    for (int k = 0; k < 2; k++) {
        if (c2[k] < 0) kids[nk++] = c2[k];
        else {
            kids[nk++] = left[c2[k]];
            kids[nk++] = right[c2[k]];
        }
    }
    Node4 out;
    for (int k = 0; k < 4; k++) {
        if (k < nk) {
            const DevBox b = nbox[node_slot(kids[k], n)];
            out.lox[k] = b.lo[0]; out.loy[k] = b.lo[1]; out.loz[k] = b.lo[2];
            out.hix[k] = b.hi[0]; out.hiy[k] = b.hi[1]; out.hiz[k] = b.hi[2];
            out.child[k] = kids[k] < 0 ? make_leaf((uint32_t)(~kids[k]), 1u) : idx4[kids[k]];
        } else {
            out.lox[k] = out.loy[k] = out.loz[k] = INFINITY;
            out.hix[k] = out.hiy[k] = out.hiz[k] = -INFINITY;
            out.child[k] = kInvalidRef;
        }
        out.pad[k] = 0;
    }
    nodes[idx4[i]] = out;
}
__global__ void k_emit4(uint32_t n, const int32_t* __restrict__ left, const int32_t* __restrict__ right, const uint32_t* __restrict__ flag4,
                        const uint32_t* __restrict__ idx4, const DevBox* __restrict__ nbox, Node4* __restrict__ nodes, uint32_t* node_count_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n - 1) return;
    emit4_node(i, n, left, right, flag4, idx4, nbox, nodes, node_count_out);
}

// n == 0 / n == 1: a root with no / one leaf child
__global__ void k_tiny_tree(const DevBox* __restrict__ boxes, uint32_t n, Node4* nodes, uint32_t* order, uint32_t* node_count_out)
{
    Node4 out;
    for (int k = 0; k < 4; k++) {
        out.lox[k] = out.loy[k] = out.loz[k] = INFINITY;
        out.hix[k] = out.hiy[k] = out.hiz[k] = -INFINITY;
        out.child[k] = kInvalidRef;
        out.pad[k] = 0;
    }
    if (n == 1) {
        out.lox[0] = boxes[0].lo[0]; out.loy[0] = boxes[0].lo[1]; out.loz[0] = boxes[0].lo[2];
        out.hix[0] = boxes[0].hi[0]; out.hiy[0] = boxes[0].hi[1]; out.hiz[0] = boxes[0].hi[2];
        out.child[0] = make_leaf(0u, 1u);
        order[0] = 0;
    }
    nodes[0] = out;
    if (node_count_out) *node_count_out = 1;
}

__device__ inline DevBox instance_box(const rfw_mat4* __restrict__ matrices, const uint32_t* __restrict__ mesh_of_instance, const DevBox* __restrict__ mesh_local,
                                      const uint32_t gid)
{
    const float* m = matrices[gid].m;
    const DevBox lb = mesh_local[mesh_of_instance[gid]];
    DevBox b;
    for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    for (int c = 0; c < 8; c++) {
        const float x = (c & 1) ? lb.hi[0] : lb.lo[0], y = (c & 2) ? lb.hi[1] : lb.lo[1], z = (c & 4) ? lb.hi[2] : lb.lo[2];
        const float w[3] = {m[0] * x + m[4] * y + m[8] * z + m[12], m[1] * x + m[5] * y + m[9] * z + m[13], m[2] * x + m[6] * y + m[10] * z + m[14]};
        for (int a = 0; a < 3; a++) { b.lo[a] = fminf(b.lo[a], w[a]); b.hi[a] = fmaxf(b.hi[a], w[a]); }
    }
    // pad for the transform's rounding and the BLAS' own padding (conservative traversal)
    for (int a = 0; a < 3; a++) {
        const float ext = b.hi[a] - b.lo[a];
        const float e = 2e-4f + 1e-5f * fmaxf(fabsf(b.lo[a]), fabsf(b.hi[a])) + 1e-5f * ext;
        b.lo[a] -= e;
        b.hi[a] += e;
    }
    b.lo[3] = 0.0f; b.hi[3] = 0.0f;
    return b;
}
__global__ void k_instance_boxes(const rfw_mat4* __restrict__ matrices, const uint32_t* __restrict__ mesh_of_instance,
                                 const DevBox* __restrict__ mesh_local, const uint32_t* __restrict__ valid_gids, uint32_t n, DevBox* out)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    out[k] = instance_box(matrices, mesh_of_instance, mesh_local, valid_gids[k]);
}

__global__ void k_gather_u32(const uint32_t* __restrict__ src, const uint32_t* __restrict__ order, uint32_t n, uint32_t* dst)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k < n) dst[k] = src[order[k]];
}

// `tris`: records of `stride16` float4s whose first three hold the vertices — the 176-B rfw_rt_triangle (11), or its 48-B head alone (3: TriHead)
__global__ void k_triangle_boxes(const float4* __restrict__ tris, uint32_t stride16, uint32_t n, DevBox* out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float4* tp = tris + (size_t)i * stride16;
    const float4 a = tp[0], b = tp[1], c = tp[2];
    DevBox bx;
    const float va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z}, vc[3] = {c.x, c.y, c.z};
    for (int k = 0; k < 3; k++) {
        float lo = fminf(va[k], fminf(vb[k], vc[k])), hi = fmaxf(va[k], fmaxf(vb[k], vc[k]));
        const float e = 1e-4f + 4e-6f * fmaxf(fabsf(lo), fabsf(hi));
        bx.lo[k] = lo - e;
        bx.hi[k] = hi + e;
    }
    bx.lo[3] = 0.0f; bx.hi[3] = 0.0f;
    out[i] = bx;
}

struct SplitPieceDev { uint32_t index; float lo[3], hi[3]; uint32_t pad; };
__global__ void k_patch_boxes(const SplitPieceDev* __restrict__ pieces, uint32_t n, DevBox* boxes)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const SplitPieceDev p = pieces[i];
    DevBox bx;
    for (int k = 0; k < 3; k++) {
        const float e = 1e-4f + 4e-6f * fmaxf(fabsf(p.lo[k]), fabsf(p.hi[k]));
        bx.lo[k] = p.lo[k] - e;
        bx.hi[k] = p.hi[k] + e;
    }
    bx.lo[3] = 0.0f; bx.hi[3] = 0.0f;
    boxes[p.index] = bx;
}
__global__ void k_resolve_duplicates(TriPacket* packets, uint32_t n, uint32_t id_offset, uint32_t n_orig, const rfw_rt_triangle* __restrict__ tris)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t local = packets[k].tri_id - id_offset;
    if (local < n_orig) return;
    const uint32_t orig = __float_as_uint(reinterpret_cast<const float*>(tris + local)[15]);
    packets[k].tri_id = id_offset + orig;
}

__global__ void k_make_packets(const rfw_rt_triangle* __restrict__ tris, const uint32_t* __restrict__ order, uint32_t n, uint32_t id_offset,
                               TriPacket* __restrict__ out)
{
    const uint32_t k = blockIdx.x * kBlock + threadIdx.x;
    if (k >= n) return;
    const uint32_t id = order[k];
    const float4* tp = reinterpret_cast<const float4*>(tris + id);
    const float4 a = tp[0], b = tp[1], c = tp[2], g = tp[3];
    TriPacket p;
    p.v0x = a.x; p.v0y = a.y; p.v0z = a.z;
    p.tri_id = id + id_offset;
    // single IEEE subtractions / one division: the same arithmetic the host builder and the oracle perform (-ffp-contract=off)
    p.e1x = b.x - a.x; p.e1y = b.y - a.y; p.e1z = b.z - a.z;
    p.e2x = c.x - a.x; p.e2y = c.y - a.y; p.e2z = c.z - a.z;
    p.inv_gn2 = 1.0f / (g.x * g.x + g.y * g.y + g.z * g.z);
    p.pad = 0.0f;
    out[k] = p;
}

// explicit 4x4 inverse, term by term the expansion of k_prepare_instances / oracle.cpp inverse() (-ffp-contract=off)
__device__ inline void mat4_inverse(const float* m, float* inv)
{
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    det = 1.0f / det;
    for (int k = 0; k < 16; k++) inv[k] = inv[k] * det;
}

__global__ void k_skin_triangles(const rfw_rt_triangle* __restrict__ src, const rfw_joint_data* __restrict__ skin, const rfw_mat4* __restrict__ joints,
                                 uint32_t n_joints, uint32_t n, rfw_rt_triangle* __restrict__ dst)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    rfw_rt_triangle t = src[i];
    float* vs[3] = {&t.vertex0.x, &t.vertex1.x, &t.vertex2.x};
    float* ns[3] = {&t.n0.x, &t.n1.x, &t.n2.x};
    float* ts[3] = {&t.tangent0.x, &t.tangent1.x, &t.tangent2.x};
    const float tw = t.tangent2.w; // every tangent takes tangent2[3] (structs.rs:838-840, 851-853, 864-866)
    for (int k = 0; k < 3; k++) {
        const rfw_joint_data jd = skin[3 * i + k];
        const float w[4] = {jd.weight.x, jd.weight.y, jd.weight.z, jd.weight.w};
        float m[16], inv[16];
        for (int j = 0; j < 4; j++) { // M = w0*J0, then M = M + w_j*J_j, element by element
            const uint32_t ji = jd.joint[j] < n_joints ? jd.joint[j] : n_joints - 1;
            const float* J = joints[ji].m;
            for (int e = 0; e < 16; e++) m[e] = j == 0 ? w[0] * J[e] : m[e] + w[j] * J[e];
        }
        mat4_inverse(m, inv);
        const float x = vs[k][0], y = vs[k][1], z = vs[k][2];
        const float nx = ns[k][0], ny = ns[k][1], nz = ns[k][2];
        const float tx = ts[k][0], ty = ts[k][1], tz = ts[k][2];
        for (int r = 0; r < 3; r++) {
            vs[k][r] = ((m[r] * x + m[4 + r] * y) + m[8 + r] * z) + m[12 + r] * 1.0f;                       // M * (v, 1)
            ns[k][r] = ((inv[4 * r] * nx + inv[4 * r + 1] * ny) + inv[4 * r + 2] * nz) + inv[4 * r + 3] * 0.0f; // transpose(inverse(M)) * (n, 0)
            ts[k][r] = ((inv[4 * r] * tx + inv[4 * r + 1] * ty) + inv[4 * r + 2] * tz) + inv[4 * r + 3] * 0.0f;
        }
        ts[k][3] = tw;
    }
    // RTTriangle::normal (structs.rs:970-975): normalize(cross(v1 - v0, v2 - v0)), normalize(v) = v * (1 / sqrt(dot))
    const float ax = t.vertex1.x - t.vertex0.x, ay = t.vertex1.y - t.vertex0.y, az = t.vertex1.z - t.vertex0.z;
    const float bx = t.vertex2.x - t.vertex0.x, by = t.vertex2.y - t.vertex0.y, bz = t.vertex2.z - t.vertex0.z;
    const float cx = ay * bz - az * by, cy = az * bx - ax * bz, cz = ax * by - ay * bx;
    const float il = 1.0f / __builtin_sqrtf(cx * cx + cy * cy + cz * cz);
    t.normal.x = cx * il; t.normal.y = cy * il; t.normal.z = cz * il;
    dst[i] = t;
}

__global__ void k_bounds_init(uint32_t* scratch)
{
    if (threadIdx.x < 3) scratch[threadIdx.x] = 0xffffffffu;
    else if (threadIdx.x < 6) scratch[threadIdx.x] = 0u;
}
__global__ void k_bounds_reduce(const rfw_rt_triangle* __restrict__ tris, uint32_t n, uint32_t* scratch)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        const float4* tp = reinterpret_cast<const float4*>(tris + i);
        const float4 a = tp[0], b = tp[1], c = tp[2];
        const float va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z}, vc[3] = {c.x, c.y, c.z};
        for (int k = 0; k < 3; k++) { lo[k] = fminf(va[k], fminf(vb[k], vc[k])); hi[k] = fmaxf(va[k], fmaxf(vb[k], vc[k])); }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int k = 0; k < 3; k++) { lo[k] = fminf(lo[k], __shfl_down(lo[k], off)); hi[k] = fmaxf(hi[k], __shfl_down(hi[k], off)); }
    if ((threadIdx.x & 63) == 0)
        for (int k = 0; k < 3; k++) { atomicMin(&scratch[k], f_order(lo[k])); atomicMax(&scratch[3 + k], f_order(hi[k])); }
}
__global__ void k_bounds_store(const uint32_t* scratch, DevBox* out)
{
    if (threadIdx.x == 0) {
        DevBox b;
        for (int k = 0; k < 3; k++) { b.lo[k] = f_unorder(scratch[k]); b.hi[k] = f_unorder(scratch[3 + k]); }
        b.lo[3] = 0.0f; b.hi[3] = 0.0f;
        *out = b;
    }
}


// ---------------------------------------------------------------- the whole TLAS build as ONE workgroup
// A scene whose instances move every frame rebuilds its TLAS every frame (the reference: on the CPU, backends/gpu-rt/src/lib.rs:1576-1615).
// lbvh_build is a chain of 17 launches, each a few microseconds of work for 10 000 instances: as a dependent chain 0.43 ms of launch gaps
// per frame, and the HOST thread that issues them was what bound BASELINE config 3 (round 5: 6366 Mrays/s where the kernels allow 7300).
// For up to kTlasFusedMax instances one workgroup of 1024 threads does every step — instance boxes, bounds of the centres, Morton keys, the
// sort (bitonic, in LDS, on (key, index) pairs: the order a stable radix sort of the keys gives), Karras' hierarchy, the bottom-up fit, the
// 4-wide collapse — with workgroup barriers where the chain had launches.  Same arithmetic, same tree, node for node
// (tests/test_gpu_api.py::test_fused_tlas_build_equals_the_chain).
constexpr uint32_t kFusedThreads = 1024;
__global__ __launch_bounds__(kFusedThreads) void k_tlas_fused(const rfw_mat4* __restrict__ matrices, const uint32_t* __restrict__ mesh_of_instance,
                                                               const DevBox* __restrict__ mesh_local, const uint32_t* __restrict__ valid_gids, const uint32_t n,
                                                               DevBox* inst_boxes, int32_t* left, int32_t* right, uint32_t* parent, DevBox* nbox,
                                                               uint32_t* flag4, uint32_t* idx4, Node4* __restrict__ nodes_out, uint32_t* __restrict__ tlas_prims,
                                                               uint32_t* node_count_out)
{
    // 128 KB of the CU's 160, used three ways one after the other: the (key, index) pairs while they are sorted; then the sorted keys
    // (first half) and the sorted indices (second half) as 32-bit words; then — keys dead after the hierarchy, indices dead after the fit —
    // the fit's arrival counters and the parents of the interior nodes.  One workgroup has 16 wavefronts to hide latency with where the
    // chain's launches had thousands: everything that is walked (counters, parent chains, sorted keys) therefore lives in LDS.
    __shared__ __attribute__((aligned(16))) unsigned long long s_pairs[kTlasFusedMax];
    uint32_t* const s_lo = reinterpret_cast<uint32_t*>(s_pairs);
    uint32_t* const s_hi = s_lo + kTlasFusedMax;
    __shared__ float s_red[2][3][kFusedThreads / 64];
    __shared__ float s_bounds[2][3];
    __shared__ uint32_t s_scan[kFusedThreads / 64];
    __builtin_amdgcn_s_setprio(3); // a frame waits for this workgroup: it goes first where it shares its CU with other frames' trace wavefronts
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    constexpr uint32_t kPer = kTlasFusedMax / kFusedThreads;
    // ---- instance boxes, the bounds of the boxes' centres (a thread keeps its boxes' centres: the keys are made from them)
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float centre[kPer][3];
#pragma unroll
    for (uint32_t q = 0; q < kPer; q++) {
        const uint32_t k = tid + q * kFusedThreads;
        for (int a = 0; a < 3; a++) centre[q][a] = 0.0f;
        if (k < n) {
            const DevBox b = instance_box(matrices, mesh_of_instance, mesh_local, valid_gids[k]);
            inst_boxes[k] = b;
            for (int a = 0; a < 3; a++) {
                const float c = 0.5f * (b.lo[a] + b.hi[a]);
                centre[q][a] = c;
                lo[a] = fminf(lo[a], c);
                hi[a] = fmaxf(hi[a], c);
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off));
            hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off));
        }
    if (lane == 0)
        for (int a = 0; a < 3; a++) { s_red[0][a][wave] = lo[a]; s_red[1][a][wave] = hi[a]; }
    __syncthreads();
    if (tid < 6) {
        const int a = (int)tid % 3, which = (int)tid / 3;
        float v = s_red[which][a][0];
        for (uint32_t w = 1; w < kFusedThreads / 64; w++) v = which ? fmaxf(v, s_red[1][a][w]) : fminf(v, s_red[0][a][w]);
        s_bounds[which][a] = v;
    }
    __syncthreads();
    // ---- (Morton key, index) pairs, padded with the largest pair to a power of two, sorted in LDS: a bitonic network whose exchanges at
    // distances below 16 happen in REGISTERS (a thread owns 16 consecutive pairs for those: one load and one store of its pairs stand for
    // four — at the start ten — passes over LDS with a barrier each).  Pair e lives at word e ^ ((e >> 4) & 15): a thread's 16 consecutive
    // pairs then lie in 16 different bank pairs across neighbouring lanes (unswizzled, all 64 lanes would hit the same two banks), and the
    // passes at distance >= 16 stay conflict-free (the swizzle permutes inside aligned groups of 16).
    uint32_t m = 16;
    while (m < n) m <<= 1;
    auto phys = [](const uint32_t e) { return e ^ ((e >> 4) & 15u); };
    {
        const float blo[3] = {s_bounds[0][0], s_bounds[0][1], s_bounds[0][2]}, bhi[3] = {s_bounds[1][0], s_bounds[1][1], s_bounds[1][2]};
#pragma unroll
        for (uint32_t q = 0; q < kPer; q++) {
            const uint32_t k = tid + q * kFusedThreads;
            if (k < m) s_pairs[phys(k)] = k < n ? ((unsigned long long)morton_key_of_centre(centre[q], blo, bhi) << 32) | k : ~0ull;
        }
    }
    __syncthreads();
    {
        const uint32_t base = tid * kPer, rot = tid & 15u;
        const bool mine_in = base < m;
        unsigned long long r[kPer];
        auto exchange = [&](const int i, const int j, const bool up) {
            const bool sw = (r[i] > r[j]) == up;
            const unsigned long long a_ = sw ? r[j] : r[i], b_ = sw ? r[i] : r[j];
            r[i] = a_; r[j] = b_;
        };
        static_assert(kPer == 16, "the register passes below are written for 16 pairs per thread");
        if (mine_in) {
#pragma unroll
            for (int q = 0; q < 16; q++) r[q] = s_pairs[base + ((uint32_t)q ^ rot)];
            // sizes 2 .. 16 entirely in registers
#pragma unroll
            for (int size = 2; size <= 16; size <<= 1)
#pragma unroll
                for (int stride = size >> 1; stride > 0; stride >>= 1)
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        if ((q & stride) == 0) exchange(q, q | stride, size == 16 ? ((base & 16u) == 0u) : ((q & size) == 0));
#pragma unroll
            for (int q = 0; q < 16; q++) s_pairs[base + ((uint32_t)q ^ rot)] = r[q];
        }
        __syncthreads();
        for (uint32_t size = 32; size <= m; size <<= 1) {
            for (uint32_t stride = size >> 1; stride >= 16u; stride >>= 1) {
                for (uint32_t t = tid; t < (m >> 1); t += kFusedThreads) {
                    const uint32_t i = ((t & ~(stride - 1u)) << 1) | (t & (stride - 1u)), j = i | stride;
                    const unsigned long long a = s_pairs[phys(i)], b = s_pairs[phys(j)];
                    const bool up = (i & size) == 0u;
                    if ((a > b) == up) { s_pairs[phys(i)] = b; s_pairs[phys(j)] = a; }
                }
                __syncthreads();
            }
            if (mine_in) {
                const bool up = (base & size) == 0u;
#pragma unroll
                for (int q = 0; q < 16; q++) r[q] = s_pairs[base + ((uint32_t)q ^ rot)];
#pragma unroll
                for (int stride = 8; stride > 0; stride >>= 1)
#pragma unroll
                    for (int q = 0; q < 16; q++)
                        if ((q & stride) == 0) exchange(q, q | stride, up);
#pragma unroll
                for (int q = 0; q < 16; q++) s_pairs[base + ((uint32_t)q ^ rot)] = r[q];
            }
            __syncthreads();
        }
    }
    // ---- pairs -> sorted keys (s_lo) | sorted indices (s_hi), in place: every thread holds its pairs across the barrier
    {
        unsigned long long mine[kPer];
        for (uint32_t q = 0; q < kPer; q++) mine[q] = tid + q * kFusedThreads < m ? s_pairs[phys(tid + q * kFusedThreads)] : 0ull;
        __syncthreads();
        for (uint32_t q = 0; q < kPer; q++) {
            const uint32_t k = tid + q * kFusedThreads;
            if (k < m) { s_lo[k] = (uint32_t)(mine[q] >> 32); s_hi[k] = (uint32_t)mine[q]; }
        }
    }
    __syncthreads();
    // ---- the instance ids in leaf order; Karras' hierarchy over the sorted keys
    const uint32_t* const keys = s_lo;
    for (uint32_t k = tid; k < n; k += kFusedThreads) tlas_prims[k] = valid_gids[s_hi[k]];
    for (uint32_t i = tid; i + 1 < n; i += kFusedThreads) hierarchy_node(keys, (int)n, (int)i, left, right, parent);
    __syncthreads(); // (workgroup-scope release / acquire: the stores above are visible to every wave of this workgroup; the keys are dead)
    uint32_t* const arrivals = s_lo;
    for (uint32_t k = tid; k < n; k += kFusedThreads) arrivals[k] = 0u;
    __syncthreads();
    // ---- bottom-up fit: the hand-off of lbvh_build's fence-free k_fit between the waves of one workgroup, its arrival counters in LDS
    fit_leaves_of_thread(inst_boxes, s_hi, tid, kFusedThreads, n, left, right, parent, arrivals, nbox);
    __syncthreads(); // (every box is published; the sorted indices are dead)
    // ---- flag4[i] = interior node i lies at even depth (the parent chains are walked in LDS); idx4 = its exclusive prefix sum
    uint32_t* const up = s_hi;
    for (uint32_t i = tid; i + 1 < n; i += kFusedThreads) up[i] = parent[i];
    __syncthreads();
    uint32_t mine[kPer], sum = 0;
    for (uint32_t q = 0; q < kPer; q++) { // (a thread owns 16 consecutive nodes)
        const uint32_t i = tid * kPer + q;
        mine[q] = i + 1 < n ? even_depth(i, up) : 0u;
        sum += mine[q];
    }
    uint32_t incl = sum;
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(incl, off);
        if ((int)lane >= off) incl += v;
    }
    if (lane == 63) s_scan[wave] = incl;
    __syncthreads();
    uint32_t before = incl - sum;
    for (uint32_t w = 0; w < wave; w++) before += s_scan[w];
    for (uint32_t q = 0; q < kPer; q++) {
        const uint32_t i = tid * kPer + q;
        if (i + 1 < n) { flag4[i] = mine[q]; idx4[i] = before; }
        before += mine[q];
    }
    __syncthreads();
    for (uint32_t i = tid; i + 1 < n; i += kFusedThreads) emit4_node(i, n, left, right, flag4, idx4, nbox, nodes_out, node_count_out);
}

inline uint32_t blocks(uint32_t n) { return (n + kBlock - 1) / kBlock; }

} // namespace

size_t lbvh_workspace_bytes(uint32_t n) { return make_layout(n, cub_temp_bytes(n)).total; }

size_t sort_pairs_workspace_bytes(uint32_t n)
{
    size_t a = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, a, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)std::max<uint32_t>(n, 1u), 0, 32);
    return a + 256;
}
hipError_t sort_pairs_u32(hipStream_t s, void* workspace, size_t workspace_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, int end_bit)
{
    if (n == 0) return hipSuccess;
    return hipcub::DeviceRadixSort::SortPairs(workspace, workspace_bytes, keys_in, keys_out, vals_in, vals_out, (int)n, 0, end_bit, s);
}

hipError_t lbvh_build(hipStream_t s, const DevBox* boxes, uint32_t n, void* workspace, size_t workspace_bytes, Node4* nodes_out, uint32_t* order_out,
                      uint32_t* node_count_out)
{
    if (n < 2) {
        hipLaunchKernelGGL(k_tiny_tree, dim3(1), dim3(1), 0, s, boxes, n, nodes_out, order_out, node_count_out);
        return hipGetLastError();
    }
    const Layout L = make_layout(n, cub_temp_bytes(n));
    if (L.total > workspace_bytes) return hipErrorInvalidValue;
    char* w = static_cast<char*>(workspace);
    uint32_t* bounds = (uint32_t*)(w + L.bounds);
    uint32_t* keys_in = (uint32_t*)(w + L.keys_in);
    uint32_t* keys_out = (uint32_t*)(w + L.keys_out);
    uint32_t* vals_in = (uint32_t*)(w + L.vals_in);
    int32_t* left = (int32_t*)(w + L.left);
    int32_t* right = (int32_t*)(w + L.right);
    uint32_t* parent = (uint32_t*)(w + L.parent);
    uint32_t* flags = (uint32_t*)(w + L.flags);
    DevBox* nbox = (DevBox*)(w + L.nbox);
    uint32_t* flag4 = (uint32_t*)(w + L.flag4);
    uint32_t* idx4 = (uint32_t*)(w + L.idx4);
    void* cub = w + L.cub;
    size_t cub_bytes = L.cub_bytes;

    hipLaunchKernelGGL(k_init_bounds, dim3(1), dim3(64), 0, s, bounds);
    hipLaunchKernelGGL(k_scene_bounds, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, n, bounds);
    hipLaunchKernelGGL(k_morton, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, n, bounds, keys_in, vals_in);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, keys_in, keys_out, vals_in, order_out, (int)n, 0, 30, s);
    if (e != hipSuccess) return e;
    e = hipMemsetAsync(flags, 0, (size_t)n * 4, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_hierarchy, dim3(blocks(n - 1)), dim3(kBlock), 0, s, keys_out, n, left, right, parent);
    const bool fenced = env_switches().lbvh_fenced;
    if (fenced) hipLaunchKernelGGL(k_fit<true>, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, order_out, n, left, right, parent, flags, nbox);
    else hipLaunchKernelGGL(k_fit<false>, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, order_out, n, left, right, parent, flags, nbox);
    hipLaunchKernelGGL(k_flag_even_depth, dim3(blocks(n - 1)), dim3(kBlock), 0, s, n - 1, parent, flag4);
    cub_bytes = L.cub_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(cub, cub_bytes, flag4, idx4, (int)(n - 1), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit4, dim3(blocks(n - 1)), dim3(kBlock), 0, s, n, left, right, flag4, idx4, nbox, nodes_out, node_count_out);
    return hipGetLastError();
}

hipError_t tlas_build_fused(hipStream_t s, const rfw_mat4* matrices, const uint32_t* mesh_of_instance, const DevBox* mesh_local_boxes, const uint32_t* valid_gids,
                            uint32_t n, void* workspace, size_t workspace_bytes, DevBox* inst_boxes, Node4* nodes_out, uint32_t* tlas_prims, uint32_t* node_count_out)
{
    if (n < 2 || n > kTlasFusedMax) return hipErrorInvalidValue;
    const Layout L = make_layout(n, 0);
    if (L.total > workspace_bytes) return hipErrorInvalidValue;
    char* w = static_cast<char*>(workspace);
    hipLaunchKernelGGL(k_tlas_fused, dim3(1), dim3(kFusedThreads), 0, s, matrices, mesh_of_instance, mesh_local_boxes, valid_gids, n, inst_boxes, (int32_t*)(w + L.left),
                       (int32_t*)(w + L.right), (uint32_t*)(w + L.parent), (DevBox*)(w + L.nbox), (uint32_t*)(w + L.flag4), (uint32_t*)(w + L.idx4), nodes_out, tlas_prims,
                       node_count_out);
    return hipGetLastError();
}

// ---- stress test of the builder (rfw_hip_debug_lbvh_stress): n jittered boxes -> tree -> exact structural check, `iterations` times, all
// on the stream.  Box k sits near cell k of a cube grid and moves and changes size with (seed, iteration): neighbouring keys, deep
// common prefixes and coincident centres all occur.
__global__ void k_stress_boxes(DevBox* out, uint32_t n, uint32_t side, uint32_t seed, uint32_t iteration)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    uint32_t h = (i * 2654435761u) ^ (seed * 40503u) ^ (iteration * 2246822519u);
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return (float)(h & 0xffffu) * (1.0f / 65536.0f); };
    const bool twin = (h & 0x30000u) == 0u; // a quarter of the boxes sit exactly on the corner of their group of four's cell: equal Morton keys
    const uint32_t j = twin ? (i & ~3u) : i;
    const float cx = (float)(j % side), cy = (float)((j / side) % side), cz = (float)(j / (side * side));
    DevBox b;
    const float c[3] = {cx + (twin ? 0.0f : 2.0f * rnd()), cy + (twin ? 0.0f : 2.0f * rnd()), cz + (twin ? 0.0f : 2.0f * rnd())};
    for (int a = 0; a < 3; a++) {
        const float e = 0.05f + 0.6f * rnd();
        b.lo[a] = c[a] - e; b.hi[a] = c[a] + e;
    }
    b.lo[3] = 0.0f; b.hi[3] = 0.0f;
    out[i] = b;
}
// every child box of every wide node must EQUAL the union of what lies below it (the fit is min / max only: exact), every primitive must be
// in exactly one leaf: a sibling's box read too early shows up as a mismatch one level up
__global__ void k_stress_validate(const Node4* __restrict__ nodes, const uint32_t* __restrict__ node_count, const DevBox* __restrict__ boxes,
                                  const uint32_t* __restrict__ order, uint32_t n, uint32_t* seen, unsigned long long* result)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= *node_count) return;
    const Node4 nd = nodes[i];
    unsigned long long errors = 0, checked = 0;
    for (int k = 0; k < 4; k++) {
        const uint32_t c = nd.child[k];
        if (c == kInvalidRef) continue;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        if (c & kLeafBit) {
            const uint32_t first = c & kLeafFirstMask, count = ((c >> 27) & 15u) + 1u;
            for (uint32_t q = 0; q < count; q++) {
                if (first + q >= n) { errors++; continue; }
                atomicAdd(&seen[order[first + q]], 1u);
                const DevBox b = boxes[order[first + q]];
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], b.lo[a]); hi[a] = fmaxf(hi[a], b.hi[a]); }
            }
        } else {
            if (c >= *node_count) { errors++; continue; }
            const Node4 ch = nodes[c];
            const float* clo[3] = {ch.lox, ch.loy, ch.loz};
            const float* chi[3] = {ch.hix, ch.hiy, ch.hiz};
            for (int kk = 0; kk < 4; kk++) {
                if (ch.child[kk] == kInvalidRef) continue;
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], clo[a][kk]); hi[a] = fmaxf(hi[a], chi[a][kk]); }
            }
        }
        const float mlo[3] = {nd.lox[k], nd.loy[k], nd.loz[k]}, mhi[3] = {nd.hix[k], nd.hiy[k], nd.hiz[k]};
        for (int a = 0; a < 3; a++)
            if (!(mlo[a] == lo[a]) || !(mhi[a] == hi[a])) errors++;
        checked++;
    }
    if (errors) atomicAdd(&result[0], errors);
    atomicAdd(&result[1], checked);
}
__global__ void k_stress_seen(const uint32_t* __restrict__ seen, uint32_t n, unsigned long long* result)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n && seen[i] != 1u) atomicAdd(&result[0], 1ull);
}
hipError_t lbvh_stress(hipStream_t s, uint32_t n, uint32_t iterations, uint32_t seed, void* workspace, size_t workspace_bytes, DevBox* boxes, Node4* nodes,
                       uint32_t* order, uint32_t* node_count, uint32_t* seen, unsigned long long* result /* [0] errors, [1] child boxes checked */)
{
    uint32_t side = 1;
    while (side * side * side < n) side++;
    for (uint32_t it = 0; it < iterations; it++) {
        hipLaunchKernelGGL(k_stress_boxes, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, n, side, seed, it);
        hipError_t e = lbvh_build(s, boxes, n, workspace, workspace_bytes, nodes, order, node_count);
        if (e != hipSuccess) return e;
        e = hipMemsetAsync(seen, 0, (size_t)n * 4, s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_stress_validate, dim3(blocks(n)), dim3(kBlock), 0, s, nodes, node_count, boxes, order, n, seen, result);
        hipLaunchKernelGGL(k_stress_seen, dim3(blocks(n)), dim3(kBlock), 0, s, seen, n, result);
    }
    return hipGetLastError();
}

void launch_instance_boxes(hipStream_t s, const rfw_mat4* matrices, const uint32_t* mesh_of_instance, const DevBox* mesh_local_boxes,
                           const uint32_t* valid_gids, uint32_t n_valid, DevBox* out)
{
    if (n_valid) hipLaunchKernelGGL(k_instance_boxes, dim3(blocks(n_valid)), dim3(kBlock), 0, s, matrices, mesh_of_instance, mesh_local_boxes, valid_gids, n_valid, out);
}
void launch_gather_u32(hipStream_t s, const uint32_t* src, const uint32_t* order, uint32_t n, uint32_t* dst)
{
    if (n) hipLaunchKernelGGL(k_gather_u32, dim3(blocks(n)), dim3(kBlock), 0, s, src, order, n, dst);
}
void launch_triangle_boxes(hipStream_t s, const rfw_rt_triangle* tris, uint32_t n, DevBox* out)
{
    if (n) hipLaunchKernelGGL(k_triangle_boxes, dim3(blocks(n)), dim3(kBlock), 0, s, reinterpret_cast<const float4*>(tris), (uint32_t)(sizeof(rfw_rt_triangle) / 16), n, out);
}
void launch_triangle_boxes(hipStream_t s, const TriHead* heads, uint32_t n, DevBox* out)
{
    if (n) hipLaunchKernelGGL(k_triangle_boxes, dim3(blocks(n)), dim3(kBlock), 0, s, reinterpret_cast<const float4*>(heads), (uint32_t)(sizeof(TriHead) / 16), n, out);
}
void launch_patch_boxes(hipStream_t s, const void* pieces, uint32_t n, DevBox* boxes)
{
    if (n) hipLaunchKernelGGL(k_patch_boxes, dim3(blocks(n)), dim3(kBlock), 0, s, static_cast<const SplitPieceDev*>(pieces), n, boxes);
}
void launch_resolve_duplicates(hipStream_t s, TriPacket* packets, uint32_t n, uint32_t id_offset, uint32_t n_orig, const rfw_rt_triangle* tris)
{
    if (n && n_orig < n) hipLaunchKernelGGL(k_resolve_duplicates, dim3(blocks(n)), dim3(kBlock), 0, s, packets, n, id_offset, n_orig, tris);
}
void launch_make_packets(hipStream_t s, const rfw_rt_triangle* tris, const uint32_t* order, uint32_t n, uint32_t id_offset, TriPacket* out)
{
    if (n) hipLaunchKernelGGL(k_make_packets, dim3(blocks(n)), dim3(kBlock), 0, s, tris, order, n, id_offset, out);
}

void launch_skin_triangles(hipStream_t s, const rfw_rt_triangle* src, const rfw_joint_data* skin, const rfw_mat4* joints, uint32_t n_joints,
                           uint32_t n_tris, rfw_rt_triangle* dst)
{
    if (n_tris && n_joints) hipLaunchKernelGGL(k_skin_triangles, dim3(blocks(n_tris)), dim3(kBlock), 0, s, src, skin, joints, n_joints, n_tris, dst);
}
void launch_mesh_bounds(hipStream_t s, const rfw_rt_triangle* tris, uint32_t n, uint32_t* scratch, DevBox* out)
{
    hipLaunchKernelGGL(k_bounds_init, dim3(1), dim3(64), 0, s, scratch);
    if (n) hipLaunchKernelGGL(k_bounds_reduce, dim3(blocks(n)), dim3(kBlock), 0, s, tris, n, scratch);
    hipLaunchKernelGGL(k_bounds_store, dim3(1), dim3(64), 0, s, scratch, out);
}

} // namespace rfwhip
