// sah_build.hip — binned-SAH BVH construction on the device (replaces the reference's per-mesh rtbvh builds on rayon,
// backends/gpu-rt/src/lib.rs:1345-1383, and BinnedSahBuilder, :1576-1581).
//
// Top-down, in two phases:
//   1. nodes with more than kSmall primitives, one LEVEL per round of four kernels over the whole primitive array:
//      bin (16 bins x 3 axes per node, aggregated per workgroup in LDS, then a few global atomics), split (one thread per node
//      sweeps its 48 bins), partition (wave-aggregated append into the two children's ranges), classify (children above
//      kSmall form the next level, the others queue for phase 2);
//   2. every queued node (<= kSmall primitives) is finished by ONE workgroup entirely in LDS: boxes and the permutation stay in
//      LDS while the workgroup walks the subtree with a small explicit stack (larger child pushed, so depth <= log2(kSmall)).
// The result is a BVH2 with multi-primitive leaves; every internal node at even depth becomes a 4-wide node whose children are
// its grandchildren (as lbvh.hip does), in the Node4 layout the traversal kernels' quantiser consumes.
// The order of primitives inside a leaf depends on atomics and is not reproducible run to run; ray results do not depend on
// the tree (padded boxes, exact-tie rule), so images are still bit-identical to the oracle.
#include "sah_build.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace rfwhip {
namespace {

constexpr int kBlock = 256;
constexpr int kBins = 16;
constexpr uint32_t kSmall = 256;   // primitives a workgroup finishes in LDS (one per thread): the CAPACITY of phase 2
// The range size at which phase 1 hands over to phase 2.  A workgroup of phase 2 walks its subtree one split after the other (~15 us
// each), so a 256-primitive range takes ~2 ms — invisible next to thousands of such workgroups of a large mesh, but the whole latency of
// a small one (a 5120-triangle mesh: 0.5 ms of levels + 2.0 ms of phase 2).  Small meshes therefore hand over at 64 primitives: two more
// levels (~0.1 ms each, launch- and read-back-bound), a quarter of the serial walk (measured: 2.5 -> 1.7 ms for that mesh, device done)
inline uint32_t small_limit_for(uint32_t n) { return n <= 32768u ? 64u : kSmall; }
constexpr uint32_t kNone = 0xffffffffu;
constexpr int kMaxLevels = 96;

struct SNode {              // 64 B
    uint32_t cb[6];         // bounds of the primitives' centroids, order-preserving uint encoding (lo xyz, hi xyz)
    uint32_t first, count;  // range in the primitive order
    float lo[3];
    uint32_t left;          // children are left, left + 1; kNone = leaf
    float hi[3];
    uint32_t parent;
};
struct Bin {                // 28 B, order-preserving uint encodings so that integer atomics apply
    uint32_t count;
    uint32_t lo[3], hi[3];
};
struct Split {
    uint32_t axis_plane;    // axis | plane << 2 | median << 8
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__device__ inline uint32_t f_order(float f)
{
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ inline float f_unorder(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

__device__ inline float half_area(const float* lo, const float* hi)
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    if (!(ex >= 0.0f) || !(ey >= 0.0f) || !(ez >= 0.0f)) return 0.0f;
    return ex * ey + ey * ez + ez * ex;
}

__device__ inline int bin_of(float c, float lo, float hi)
{
    if (!(hi > lo)) return 0;
    int b = (int)((c - lo) * ((float)kBins / (hi - lo)));
    return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
}

struct Counters {
    uint32_t node_count;     // BVH2 nodes allocated
    uint32_t n_active[2];    // big nodes of the current / next level
    uint32_t n_small;        // nodes queued for phase 2
    uint32_t root_bounds[12];
};

// ---------------------------------------------------------------- root
__global__ void k_root_init(Counters* ctr)
{
    if (threadIdx.x < 6) ctr->root_bounds[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;          // box
    else if (threadIdx.x < 12) ctr->root_bounds[threadIdx.x] = threadIdx.x < 9 ? 0xffffffffu : 0u;    // centroids
    if (threadIdx.x == 0) { ctr->node_count = 1; ctr->n_active[0] = 0; ctr->n_active[1] = 0; ctr->n_small = 0; }
}
__global__ void k_root_bounds(const DevBox* __restrict__ boxes, uint32_t n, Counters* ctr, uint32_t* order, uint32_t* node_of_pos)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    if (i < n) {
        order[i] = i;
        node_of_pos[i] = 0;
        for (int a = 0; a < 3; a++) {
            lo[a] = boxes[i].lo[a]; hi[a] = boxes[i].hi[a];
            clo[a] = chi[a] = 0.5f * (lo[a] + hi[a]);
        }
    }
    for (int off = 32; off > 0; off >>= 1)
        for (int a = 0; a < 3; a++) {
            lo[a] = fminf(lo[a], __shfl_down(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off));
            clo[a] = fminf(clo[a], __shfl_down(clo[a], off)); chi[a] = fmaxf(chi[a], __shfl_down(chi[a], off));
        }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) {
            atomicMin(&ctr->root_bounds[a], f_order(lo[a])); atomicMax(&ctr->root_bounds[3 + a], f_order(hi[a]));
            atomicMin(&ctr->root_bounds[6 + a], f_order(clo[a])); atomicMax(&ctr->root_bounds[9 + a], f_order(chi[a]));
        }
}
__global__ void k_root_node(uint32_t n, Counters* ctr, SNode* nodes, uint32_t* active, uint32_t* small, uint32_t* bin_slot, uint8_t* is_big, Bin* bins, uint32_t small_limit)
{
    if (threadIdx.x != 0) return;
    SNode r;
    for (int a = 0; a < 3; a++) {
        r.lo[a] = f_unorder(ctr->root_bounds[a]); r.hi[a] = f_unorder(ctr->root_bounds[3 + a]);
        r.cb[a] = ctr->root_bounds[6 + a]; r.cb[3 + a] = ctr->root_bounds[9 + a];
    }
    r.first = 0; r.count = n; r.left = kNone; r.parent = kNone;
    nodes[0] = r;
    if (n > small_limit) {
        active[0] = 0; bin_slot[0] = 0; is_big[0] = 1; ctr->n_active[0] = 1;
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        for (int k = 0; k < 3 * kBins; k++) bins[k] = e;
    } else {
        small[0] = 0; is_big[0] = 0; ctr->n_small = 1;
    }
}

// ---------------------------------------------------------------- phase 1: one level
__global__ __launch_bounds__(kBlock) void k_bin(const DevBox* __restrict__ boxes, const uint32_t* __restrict__ order, const uint32_t* __restrict__ node_of_pos,
                                               const SNode* __restrict__ nodes, const uint32_t* __restrict__ bin_slot, const uint8_t* __restrict__ is_big,
                                               Bin* bins, uint32_t n)
{
    __shared__ Bin sb[3 * kBins];
    __shared__ uint32_t s_node, s_mixed;
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t nd = p < n ? node_of_pos[p] : kNone;
    if (nd != kNone && !is_big[nd]) nd = kNone;
    if (threadIdx.x == 0) { s_node = kNone; s_mixed = 0; }
    __syncthreads();
    if (nd != kNone) atomicMin(&s_node, nd);
    __syncthreads();
    const uint32_t first_node = s_node;
    if (first_node == kNone) return; // nothing active in this workgroup
    if (nd != kNone && nd != first_node) s_mixed = 1;
    if (threadIdx.x < 3 * kBins) {
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        sb[threadIdx.x] = e;
    }
    __syncthreads();
    const bool mixed = s_mixed != 0;
    if (nd != kNone) {
        const DevBox b = boxes[order[p]];
        const SNode& node = nodes[nd];
        Bin* gb = bins + (size_t)bin_slot[nd] * 3 * kBins;
        for (int a = 0; a < 3; a++) {
            const int k = a * kBins + bin_of(0.5f * (b.lo[a] + b.hi[a]), f_unorder(node.cb[a]), f_unorder(node.cb[3 + a]));
            if (mixed) { // several nodes in this workgroup: straight to the node's bins in HBM
                atomicAdd(&gb[k].count, 1u);
                for (int c = 0; c < 3; c++) { atomicMin(&gb[k].lo[c], f_order(b.lo[c])); atomicMax(&gb[k].hi[c], f_order(b.hi[c])); }
            } else {
                atomicAdd(&sb[k].count, 1u);
                for (int c = 0; c < 3; c++) { atomicMin(&sb[k].lo[c], f_order(b.lo[c])); atomicMax(&sb[k].hi[c], f_order(b.hi[c])); }
            }
        }
    }
    if (mixed) return; // several nodes in one workgroup: the lanes went to the global bins directly
    __syncthreads();
    if (threadIdx.x < 3 * kBins && sb[threadIdx.x].count) {
        Bin* e = bins + (size_t)bin_slot[first_node] * 3 * kBins + threadIdx.x;
        atomicAdd(&e->count, sb[threadIdx.x].count);
        for (int c = 0; c < 3; c++) { atomicMin(&e->lo[c], sb[threadIdx.x].lo[c]); atomicMax(&e->hi[c], sb[threadIdx.x].hi[c]); }
    }
}

// the SAH sweep over one node's 48 bins: best (axis, plane), the two child boxes and the left count
struct SplitChoice {
    int axis, plane;
    uint32_t left_count;
    float llo[3], lhi[3], rlo[3], rhi[3];
    float cost;
};
__device__ inline void grow(float* lo, float* hi, const Bin& b)
{
    for (int c = 0; c < 3; c++) { lo[c] = fminf(lo[c], f_unorder(b.lo[c])); hi[c] = fmaxf(hi[c], f_unorder(b.hi[c])); }
}
template <typename BinPtr> __device__ inline SplitChoice sweep_axis(BinPtr bins, const int a)
{
    SplitChoice best;
    best.axis = -1; best.plane = -1; best.left_count = 0; best.cost = INFINITY;
    float right_area[kBins];
    uint32_t right_cnt[kBins];
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t c = 0;
    for (int b = kBins - 1; b > 0; b--) {
        const Bin e = bins[a * kBins + b];
        if (e.count) grow(lo, hi, e);
        c += e.count;
        right_area[b] = half_area(lo, hi);
        right_cnt[b] = c;
    }
    for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    c = 0;
    for (int b = 0; b < kBins - 1; b++) {
        const Bin e = bins[a * kBins + b];
        if (e.count) grow(lo, hi, e);
        c += e.count;
        if (c == 0 || right_cnt[b + 1] == 0) continue;
        const float cost = (float)c * half_area(lo, hi) + (float)right_cnt[b + 1] * right_area[b + 1];
        if (cost < best.cost) { best.cost = cost; best.axis = a; best.plane = b; best.left_count = c; }
    }
    if (best.axis >= 0) { // the two child boxes of the chosen plane
        for (int k = 0; k < 3; k++) { best.llo[k] = best.rlo[k] = INFINITY; best.lhi[k] = best.rhi[k] = -INFINITY; }
        for (int b = 0; b < kBins; b++) {
            const Bin e = bins[a * kBins + b];
            if (!e.count) continue;
            if (b <= best.plane) grow(best.llo, best.lhi, e);
            else grow(best.rlo, best.rhi, e);
        }
    }
    return best;
}
template <typename BinPtr> __device__ inline SplitChoice sweep_bins(BinPtr bins)
{
    SplitChoice best = sweep_axis(bins, 0);
    for (int a = 1; a < 3; a++) {
        const SplitChoice s = sweep_axis(bins, a);
        if (s.axis >= 0 && (best.axis < 0 || s.cost < best.cost)) best = s;
    }
    return best;
}

__device__ inline void init_child(SNode& c, uint32_t first, uint32_t count, const float* lo, const float* hi, uint32_t parent)
{
    for (int a = 0; a < 3; a++) { c.lo[a] = lo[a]; c.hi[a] = hi[a]; c.cb[a] = 0xffffffffu; c.cb[3 + a] = 0u; }
    c.first = first; c.count = count; c.left = kNone; c.parent = parent;
}

__global__ void k_split(const uint32_t* __restrict__ active, uint32_t level_parity, Counters* ctr, SNode* nodes, const Bin* __restrict__ bins,
                        const uint32_t* __restrict__ bin_slot, Split* splits, uint32_t* fill)
{
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= ctr->n_active[level_parity]) return;
    const uint32_t nd = active[k];
    SNode node = nodes[nd];
    const Bin* nb = bins + (size_t)bin_slot[nd] * 3 * kBins;
    SplitChoice s = sweep_bins(nb);
    const uint32_t li = atomicAdd(&ctr->node_count, 2u);
    SNode l, r;
    uint32_t code;
    if (s.axis >= 0) {
        init_child(l, node.first, s.left_count, s.llo, s.lhi, nd);
        init_child(r, node.first + s.left_count, node.count - s.left_count, s.rlo, s.rhi, nd);
        code = (uint32_t)s.axis | ((uint32_t)s.plane << 2);
    } else { // every centroid in one bin on every axis: halve the range by position (boxes: the parent's, conservatively)
        init_child(l, node.first, node.count / 2, node.lo, node.hi, nd);
        init_child(r, node.first + node.count / 2, node.count - node.count / 2, node.lo, node.hi, nd);
        code = 1u << 8;
    }
    nodes[li] = l;
    nodes[li + 1] = r;
    nodes[nd].left = li;
    splits[nd].axis_plane = code;
    fill[li] = 0;
    fill[li + 1] = 0;
}

__global__ __launch_bounds__(kBlock) void k_partition(const DevBox* __restrict__ boxes, const uint32_t* __restrict__ order_in, const uint32_t* __restrict__ nop_in,
                                                     uint32_t* order_out, uint32_t* nop_out, SNode* nodes, const Split* __restrict__ splits,
                                                     const uint8_t* __restrict__ is_big, uint32_t* fill, uint32_t n)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t nd = p < n ? nop_in[p] : kNone;
    const uint32_t prim = p < n ? order_in[p] : 0u;
    const bool act = nd != kNone && is_big[nd];
    if (p < n && !act) { // finished ranges keep their place
        order_out[p] = prim;
        nop_out[p] = nd;
    }
    // every lane stays in the kernel: the wave-level reductions below need the whole wavefront
    uint32_t child = kNone;
    float c3[3] = {0.0f, 0.0f, 0.0f};
    if (act) {
        const SNode& node = nodes[nd];
        const uint32_t code = splits[nd].axis_plane;
        const DevBox b = boxes[prim];
        for (int a = 0; a < 3; a++) c3[a] = 0.5f * (b.lo[a] + b.hi[a]);
        bool left;
        if (code >> 8) left = (p - node.first) < node.count / 2;
        else {
            const int a = (int)(code & 3u);
            left = bin_of(c3[a], f_unorder(node.cb[a]), f_unorder(node.cb[3 + a])) <= (int)((code >> 2) & 63u);
        }
        child = node.left + (left ? 0u : 1u);
    }
    const unsigned long long actmask = __ballot(act);
    if (actmask == 0ull) return;
    const int lead = __ffsll((long long)actmask) - 1;
    const uint32_t lead_node = (uint32_t)__shfl((int)nd, lead);
    const unsigned long long same = __ballot(act && nd == lead_node);
    uint32_t dest = 0;
    if (same == actmask) {
        // the common case: every active lane of the wave belongs to one node -> per child one atomic for the slots and six for
        // the centroid bounds, however the lanes interleave between the two children
        const uint32_t li = nodes[lead_node].left;
        const bool to_right = act && child != li;
        const unsigned long long rmask = __ballot(to_right), lmask = actmask & ~rmask;
        const unsigned long long mine = to_right ? rmask : lmask;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        uint32_t base = 0;
        if (act && rank == 0) base = atomicAdd(&fill[child], (uint32_t)__popcll(mine));
        const int my_lead = __ffsll((long long)mine) - 1;
        base = (uint32_t)__shfl((int)base, act ? my_lead : 0);
        dest = base + rank;
        for (int side = 0; side < 2; side++) {
            const unsigned long long m = side ? rmask : lmask;
            if (m == 0ull) continue; // wave-uniform
            const bool in = act && (to_right == (side != 0));
            float lo[3], hi[3];
            for (int a = 0; a < 3; a++) { lo[a] = in ? c3[a] : INFINITY; hi[a] = in ? c3[a] : -INFINITY; }
            for (int off = 32; off > 0; off >>= 1)
                for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], __shfl_xor(lo[a], off)); hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], off)); }
            if ((int)(threadIdx.x & 63u) == lead)
                for (int a = 0; a < 3; a++) { atomicMin(&nodes[li + side].cb[a], f_order(lo[a])); atomicMax(&nodes[li + side].cb[3 + a], f_order(hi[a])); }
        }
    } else if (act) {
        dest = atomicAdd(&fill[child], 1u);
        for (int a = 0; a < 3; a++) { atomicMin(&nodes[child].cb[a], f_order(c3[a])); atomicMax(&nodes[child].cb[3 + a], f_order(c3[a])); }
    }
    if (act) {
        dest += nodes[child].first;
        order_out[dest] = prim;
        nop_out[dest] = child;
    }
}

__global__ void k_classify(const uint32_t* __restrict__ active_in, uint32_t* active_out, uint32_t level_parity, Counters* ctr, const SNode* __restrict__ nodes,
                           uint32_t* small, uint32_t* bin_slot, uint8_t* is_big, Bin* bins, uint32_t small_limit)
{
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= 2u * ctr->n_active[level_parity]) return;
    const uint32_t parent = active_in[k >> 1];
    const uint32_t nd = nodes[parent].left + (k & 1u);
    if ((k & 1u) == 0u) is_big[parent] = 0; // the parent's range now belongs to its children
    if (nodes[nd].count > small_limit) {
        const uint32_t slot = atomicAdd(&ctr->n_active[level_parity ^ 1u], 1u);
        active_out[slot] = nd;
        bin_slot[nd] = slot;
        is_big[nd] = 1;
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        for (int b = 0; b < 3 * kBins; b++) bins[(size_t)slot * 3 * kBins + b] = e;
    } else {
        is_big[nd] = 0;
        small[atomicAdd(&ctr->n_small, 1u)] = nd;
    }
}
__global__ void k_next_level(Counters* ctr, uint32_t level_parity)
{
    if (threadIdx.x == 0) ctr->n_active[level_parity] = 0; // the level just finished; its slot counts the level after next
}

// ---------------------------------------------------------------- phase 2: one workgroup finishes one node of <= kSmall primitives
__global__ __launch_bounds__(kBlock) void k_small(const DevBox* __restrict__ boxes, const uint32_t* __restrict__ order_in, uint32_t* order_out,
                                                 const uint32_t* __restrict__ small, Counters* ctr, SNode* nodes, int max_leaf, float trav_cost)
{
    static_assert(kSmall <= kBlock, "one primitive per thread");
    __shared__ float s_lo[3][kSmall], s_hi[3][kSmall];
    __shared__ uint16_t s_perm[2][kSmall];
    __shared__ Bin s_bins[3 * kBins];
    __shared__ uint32_t s_cb[6];
    __shared__ uint32_t s_stack[3 * 16];
    __shared__ uint32_t s_fill[2];
    __shared__ SplitChoice s_axis[3];
    __shared__ SplitChoice s_choice;
    __shared__ SNode s_node;          // the node being processed (box, ids)
    __shared__ uint32_t s_decision[2]; // 0: 1 = split; 1: left child node id
    __shared__ uint32_t s_next;        // next unused node id of this workgroup's reservation
    const uint32_t tid = threadIdx.x;
    const uint32_t root = small[blockIdx.x];
    const uint32_t gfirst = nodes[root].first, gcount = nodes[root].count;
    if (tid < gcount) {
        const DevBox b = boxes[order_in[gfirst + tid]];
        for (int a = 0; a < 3; a++) { s_lo[a][tid] = b.lo[a]; s_hi[a][tid] = b.hi[a]; }
        s_perm[0][tid] = (uint16_t)tid;
    }
    if (tid == 0) {
        s_next = atomicAdd(&ctr->node_count, 2u * gcount); // a subtree over c primitives has at most 2c - 2 nodes below its root
        s_node = nodes[root];
    }
    int sp = 0;
    uint32_t first = 0, count = gcount, nid = root; // range in s_perm[0], local
    __syncthreads();
    for (;;) {
        // ---- centroid bounds and bins of [first, first + count)
        if (tid < 6) s_cb[tid] = tid < 3 ? 0xffffffffu : 0u;
        if (tid < 3 * kBins) {
            Bin e; e.count = 0;
            for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
            s_bins[tid] = e;
        }
        __syncthreads();
        const bool mine = tid < count;
        const uint32_t q = mine ? s_perm[0][first + tid] : 0u;
        float cen[3] = {0.0f, 0.0f, 0.0f};
        if (mine) {
            for (int a = 0; a < 3; a++) {
                cen[a] = 0.5f * (s_lo[a][q] + s_hi[a][q]);
                const uint32_t ce = f_order(cen[a]);
                atomicMin(&s_cb[a], ce);
                atomicMax(&s_cb[3 + a], ce);
            }
        }
        __syncthreads();
        int my_bin[3] = {0, 0, 0};
        if (mine) {
            for (int a = 0; a < 3; a++) {
                my_bin[a] = bin_of(cen[a], f_unorder(s_cb[a]), f_unorder(s_cb[3 + a]));
                Bin* e = &s_bins[a * kBins + my_bin[a]];
                atomicAdd(&e->count, 1u);
                for (int c = 0; c < 3; c++) { atomicMin(&e->lo[c], f_order(s_lo[c][q])); atomicMax(&e->hi[c], f_order(s_hi[c][q])); }
            }
        }
        __syncthreads();
        if (tid < 3) s_axis[tid] = sweep_axis(s_bins, (int)tid); // one lane per axis
        __syncthreads();
        // ---- decide (one thread): leaf, SAH split, or halve a range whose centroids coincide
        if (tid == 0) {
            SplitChoice s = s_axis[0];
            for (int a = 1; a < 3; a++)
                if (s_axis[a].axis >= 0 && (s.axis < 0 || s_axis[a].cost < s.cost)) s = s_axis[a];
            const float area = half_area(s_node.lo, s_node.hi);
            const float leaf_cost = (float)count * area;
            bool split = false;
            if (count > 1) {
                if (s.axis >= 0 && (s.cost + trav_cost * area < leaf_cost || (int)count > max_leaf)) split = true;
                else if ((int)count > max_leaf) { // coincident centroids: arbitrary halves keep leaves bounded
                    split = true;
                    s.axis = -1;
                    s.left_count = count / 2;
                    for (int a = 0; a < 3; a++) { s.llo[a] = s.rlo[a] = s_node.lo[a]; s.lhi[a] = s.rhi[a] = s_node.hi[a]; }
                }
            }
            s_choice = s;
            s_decision[0] = split ? 1u : 0u;
            if (split) {
                const uint32_t li = s_next;
                s_next += 2;
                s_decision[1] = li;
                SNode l, r;
                init_child(l, gfirst + first, s.left_count, s.llo, s.lhi, nid);
                init_child(r, gfirst + first + s.left_count, count - s.left_count, s.rlo, s.rhi, nid);
                nodes[li] = l;
                nodes[li + 1] = r;
                nodes[nid].left = li;
                // the smaller child is processed next, the larger one waits on the stack: never more than log2(kSmall) entries
                const bool left_next = s.left_count <= count - s.left_count;
                s_stack[3 * sp + 0] = left_next ? first + s.left_count : first;
                s_stack[3 * sp + 1] = left_next ? count - s.left_count : s.left_count;
                s_stack[3 * sp + 2] = left_next ? li + 1 : li;
                s_node = left_next ? l : r;
            } else if (sp > 0) {
                s_node = nodes[s_stack[3 * (sp - 1) + 2]]; // written by this thread earlier
            }
            s_fill[0] = 0;
            s_fill[1] = 0;
        }
        __syncthreads();
        if (s_decision[0] != 0u) {
            const int axis = s_choice.axis, plane = s_choice.plane;
            const uint32_t lc = s_choice.left_count;
            if (mine) {
                const bool left = axis < 0 ? tid < lc : my_bin[axis] <= plane;
                const uint32_t d = left ? atomicAdd(&s_fill[0], 1u) : lc + atomicAdd(&s_fill[1], 1u);
                s_perm[1][first + d] = (uint16_t)q;
            }
            __syncthreads();
            if (mine) s_perm[0][first + tid] = s_perm[1][first + tid];
            const uint32_t li = s_decision[1], rc = count - lc;
            sp++;
            if (lc <= rc) { count = lc; nid = li; }
            else { first = first + lc; count = rc; nid = li + 1; }
            __syncthreads();
            continue;
        }
        // ---- leaf: nodes[nid].left stays kNone; take the next range
        if (sp == 0) break;
        sp--;
        first = s_stack[3 * sp + 0];
        count = s_stack[3 * sp + 1];
        nid = s_stack[3 * sp + 2];
        __syncthreads();
    }
    __syncthreads();
    if (tid < gcount) order_out[gfirst + tid] = order_in[gfirst + s_perm[0][tid]];
}

// ---------------------------------------------------------------- BVH2 -> Node4
__global__ void k_flag_nodes(uint32_t n_nodes_cap, const Counters* __restrict__ ctr, const SNode* __restrict__ nodes, uint32_t* flag4)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes_cap) return;
    uint32_t f = 0;
    if (i < ctr->node_count && nodes[i].left != kNone) {
        uint32_t depth = 0, p = nodes[i].parent;
        while (p != kNone) { depth++; p = nodes[p].parent; }
        f = (depth & 1u) ? 0u : 1u;
    }
    flag4[i] = f;
}
__global__ void k_emit_nodes(uint32_t n_nodes_cap, const Counters* __restrict__ ctr, const SNode* __restrict__ nodes, const uint32_t* __restrict__ flag4,
                             const uint32_t* __restrict__ idx4, Node4* out_nodes, uint32_t* node_count_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes_cap) return;
    const uint32_t total = ctr->node_count;
    if (i == 0) {
        if (nodes[0].left == kNone) { // the whole tree is one leaf
            Node4 o;
            for (int k = 0; k < 4; k++) {
                o.lox[k] = o.loy[k] = o.loz[k] = INFINITY; o.hix[k] = o.hiy[k] = o.hiz[k] = -INFINITY;
                o.child[k] = kInvalidRef; o.pad[k] = 0;
            }
            if (nodes[0].count) {
                o.lox[0] = nodes[0].lo[0]; o.loy[0] = nodes[0].lo[1]; o.loz[0] = nodes[0].lo[2];
                o.hix[0] = nodes[0].hi[0]; o.hiy[0] = nodes[0].hi[1]; o.hiz[0] = nodes[0].hi[2];
                o.child[0] = make_leaf(nodes[0].first, nodes[0].count);
            }
            out_nodes[0] = o;
            if (node_count_out) *node_count_out = 1;
            return;
        }
        if (node_count_out) *node_count_out = idx4[total - 1] + flag4[total - 1];
    }
    if (i >= total || !flag4[i]) return;
    uint32_t kids[4];
    int nk = 0;
    for (uint32_t c = nodes[i].left; c < nodes[i].left + 2; c++) {
        if (nodes[c].left == kNone) kids[nk++] = c;
        else { kids[nk++] = nodes[c].left; kids[nk++] = nodes[c].left + 1; }
    }
    Node4 o;
    for (int k = 0; k < 4; k++) {
        if (k < nk) {
            const SNode& c = nodes[kids[k]];
            o.lox[k] = c.lo[0]; o.loy[k] = c.lo[1]; o.loz[k] = c.lo[2];
            o.hix[k] = c.hi[0]; o.hiy[k] = c.hi[1]; o.hiz[k] = c.hi[2];
            o.child[k] = c.left == kNone ? make_leaf(c.first, c.count) : idx4[kids[k]];
        } else {
            o.lox[k] = o.loy[k] = o.loz[k] = INFINITY; o.hix[k] = o.hiy[k] = o.hiz[k] = -INFINITY;
            o.child[k] = kInvalidRef;
        }
        o.pad[k] = 0;
    }
    out_nodes[idx4[i]] = o;
}

// ---------------------------------------------------------------- refit
__global__ void k_refit_setup(const Node4* __restrict__ nodes, uint32_t n_nodes, uint32_t* parent_slot, uint32_t* n_internal)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes) return;
    if (i == 0) parent_slot[0] = kNone;
    uint32_t cnt = 0;
    for (uint32_t k = 0; k < 4; k++) {
        const uint32_t c = nodes[i].child[k];
        if (c == kInvalidRef || (c & kLeafBit)) continue;
        parent_slot[c] = 4u * i + k;
        cnt++;
    }
    n_internal[i] = cnt;
}

// boxes of the leaf children: the padded triangle boxes of launch_triangle_boxes, from the triangles as they are now
__global__ void k_refit_leaves(Node4* nodes, uint32_t n_nodes, const rfw_rt_triangle* __restrict__ tris, const uint32_t* __restrict__ order)
{
    const uint32_t t = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t i = t >> 2, k = t & 3u;
    if (i >= n_nodes) return;
    const uint32_t c = nodes[i].child[k];
    if (c == kInvalidRef || !(c & kLeafBit)) return;
    const uint32_t first = c & kLeafFirstMask, count = ((c >> 27) & 15u) + 1u;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t j = 0; j < count; j++) {
        const float4* tp = reinterpret_cast<const float4*>(tris + order[first + j]);
        const float4 a = tp[0], b = tp[1], cc = tp[2];
        const float va[3] = {a.x, a.y, a.z}, vb[3] = {b.x, b.y, b.z}, vc[3] = {cc.x, cc.y, cc.z};
        for (int d = 0; d < 3; d++) {
            const float l = fminf(va[d], fminf(vb[d], vc[d])), h = fmaxf(va[d], fmaxf(vb[d], vc[d]));
            const float e = 1e-4f + 4e-6f * fmaxf(fabsf(l), fabsf(h));
            lo[d] = fminf(lo[d], l - e);
            hi[d] = fmaxf(hi[d], h + e);
        }
    }
    nodes[i].lox[k] = lo[0]; nodes[i].loy[k] = lo[1]; nodes[i].loz[k] = lo[2];
    nodes[i].hix[k] = hi[0]; nodes[i].hiy[k] = hi[1]; nodes[i].hiz[k] = hi[2];
}

// interior boxes, bottom-up: a thread starts at every node without interior children, writes the node's box into its parent's
// slot and carries on with the parent if it is the last of the parent's interior children to arrive
__global__ void k_refit_up(Node4* nodes, uint32_t n_nodes, const uint32_t* __restrict__ parent_slot, const uint32_t* __restrict__ n_internal, uint32_t* arrive)
{
    uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes || n_internal[i] != 0u) return;
    for (;;) {
        const uint32_t ps = parent_slot[i];
        if (ps == kNone) return; // the root has no slot to fill
        const volatile Node4* nd = nodes + i;
        float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (int k = 0; k < 4; k++) {
            if (nd->child[k] == kInvalidRef) continue;
            lo[0] = fminf(lo[0], nd->lox[k]); lo[1] = fminf(lo[1], nd->loy[k]); lo[2] = fminf(lo[2], nd->loz[k]);
            hi[0] = fmaxf(hi[0], nd->hix[k]); hi[1] = fmaxf(hi[1], nd->hiy[k]); hi[2] = fmaxf(hi[2], nd->hiz[k]);
        }
        const uint32_t p = ps >> 2, k = ps & 3u;
        nodes[p].lox[k] = lo[0]; nodes[p].loy[k] = lo[1]; nodes[p].loz[k] = lo[2];
        nodes[p].hix[k] = hi[0]; nodes[p].hiy[k] = hi[1]; nodes[p].hiz[k] = hi[2];
        __threadfence();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (atomicAdd(&arrive[p], 1u) + 1u != n_internal[p]) return;
        __threadfence();
        i = p;
    }
}

struct Layout {
    size_t ctr, nodes, order[2], nop[2], active[2], small, bin_slot, is_big, splits, fill, bins, flag4, idx4, cub, total, cub_bytes;
    uint32_t node_cap, big_cap;
};
Layout make_layout(uint32_t n)
{
    Layout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t m = n > 0 ? n : 1;
    L.node_cap = (uint32_t)(4 * m + 64); // phase 1 makes < 2 nodes per queued range, phase 2 reserves 2c ids for a range of c primitives
    L.big_cap = (uint32_t)(2 * m / small_limit_for(n) + 2); // ranges above the hand-over size that can coexist on one level
    L.ctr = take(sizeof(Counters));
    L.nodes = take((size_t)L.node_cap * sizeof(SNode));
    for (int k = 0; k < 2; k++) { L.order[k] = take(m * 4); L.nop[k] = take(m * 4); L.active[k] = take((size_t)L.big_cap * 4); }
    L.small = take(m * 4);
    L.bin_slot = take((size_t)L.node_cap * 4);
    L.is_big = take((size_t)L.node_cap);
    L.splits = take((size_t)L.node_cap * sizeof(Split));
    L.fill = take((size_t)L.node_cap * 4);
    L.bins = take((size_t)L.big_cap * 3 * kBins * sizeof(Bin));
    L.flag4 = take((size_t)L.node_cap * 4);
    L.idx4 = take((size_t)L.node_cap * 4);
    size_t cb = 0;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, cb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (int)L.node_cap);
    L.cub_bytes = cb + 256;
    L.cub = take(L.cub_bytes);
    L.total = off;
    return L;
}

inline uint32_t blocks(uint32_t n, uint32_t per = kBlock) { return (n + per - 1) / per; }

} // namespace

size_t sah_workspace_bytes(uint32_t n) { return make_layout(n).total; }

hipError_t sah_build(hipStream_t s, const DevBox* boxes, uint32_t n, void* workspace, size_t workspace_bytes, Node4* nodes_out, uint32_t* order_out,
                     uint32_t* node_count_out, int max_leaf, float trav_cost)
{
    const Layout L = make_layout(n);
    if (L.total > workspace_bytes) return hipErrorInvalidValue;
    max_leaf = max_leaf < 1 ? 1 : (max_leaf > kMaxLeafTris ? kMaxLeafTris : max_leaf);
    char* w = static_cast<char*>(workspace);
    Counters* ctr = (Counters*)(w + L.ctr);
    SNode* nodes = (SNode*)(w + L.nodes);
    uint32_t* order[2] = {(uint32_t*)(w + L.order[0]), (uint32_t*)(w + L.order[1])};
    uint32_t* nop[2] = {(uint32_t*)(w + L.nop[0]), (uint32_t*)(w + L.nop[1])};
    uint32_t* active[2] = {(uint32_t*)(w + L.active[0]), (uint32_t*)(w + L.active[1])};
    uint32_t* small = (uint32_t*)(w + L.small);
    uint32_t* bin_slot = (uint32_t*)(w + L.bin_slot);
    uint8_t* is_big = (uint8_t*)(w + L.is_big);
    Split* splits = (Split*)(w + L.splits);
    uint32_t* fill = (uint32_t*)(w + L.fill);
    Bin* bins = (Bin*)(w + L.bins);
    uint32_t* flag4 = (uint32_t*)(w + L.flag4);
    uint32_t* idx4 = (uint32_t*)(w + L.idx4);

    {   // ids a workgroup of phase 2 reserves but does not use must read as leaves nobody references: left = kNone
        const hipError_t me = hipMemsetAsync(nodes, 0xff, (size_t)L.node_cap * sizeof(SNode), s);
        if (me != hipSuccess) return me;
    }
    hipLaunchKernelGGL(k_root_init, dim3(1), dim3(64), 0, s, ctr);
    if (n) hipLaunchKernelGGL(k_root_bounds, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, n, ctr, order[0], nop[0]);
    const uint32_t small_limit = small_limit_for(n);
    hipLaunchKernelGGL(k_root_node, dim3(1), dim3(64), 0, s, n, ctr, nodes, active[0], small, bin_slot, is_big, bins, small_limit);
    static const bool dbg = getenv("RFW_SAH_DEBUG") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto msf = [](auto a, auto b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
    if (dbg) (void)hipStreamSynchronize(s);
    const auto t_start = now();
    int levels = 0;
    // phase 1: level by level while nodes above kSmall remain (the count comes back to the host once per level)
    int cur = 0;
    uint32_t n_active = n > small_limit ? 1u : 0u;
    for (int level = 0; level < kMaxLevels && n_active > 0; level++) {
        const uint32_t par = (uint32_t)(level & 1);
        hipLaunchKernelGGL(k_bin, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, order[cur], nop[cur], nodes, bin_slot, is_big, bins, n);
        hipLaunchKernelGGL(k_split, dim3(blocks(n_active, 64)), dim3(64), 0, s, active[par], par, ctr, nodes, bins, bin_slot, splits, fill);
        hipLaunchKernelGGL(k_partition, dim3(blocks(n)), dim3(kBlock), 0, s, boxes, order[cur], nop[cur], order[cur ^ 1], nop[cur ^ 1], nodes, splits, is_big, fill, n);
        hipLaunchKernelGGL(k_classify, dim3(blocks(2 * n_active, 64)), dim3(64), 0, s, active[par], active[par ^ 1], par, ctr, nodes, small, bin_slot, is_big, bins, small_limit);
        hipLaunchKernelGGL(k_next_level, dim3(1), dim3(64), 0, s, ctr, par);
        cur ^= 1;
        uint32_t next = 0;
        hipError_t e = hipMemcpyAsync(&next, &ctr->n_active[par ^ 1], 4, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return e;
        if (next > L.big_cap) return hipErrorInvalidValue;
        n_active = next;
        levels++;
        if (dbg) fprintf(stderr, "sah level %d: next active %u, %.3f ms since start\n", level, next, msf(t_start, now()));
    }
    const auto t_p1 = now();
    if (n_active > 0) return hipErrorInvalidValue; // deeper than kMaxLevels above kSmall: not a tree this builder makes
    // phase 2
    uint32_t n_small = 0;
    hipError_t e = hipMemcpyAsync(&n_small, &ctr->n_small, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return e;
    if (n_small) hipLaunchKernelGGL(k_small, dim3(n_small), dim3(kBlock), 0, s, boxes, order[cur], order_out, small, ctr, nodes, max_leaf, trav_cost);
    if (dbg) { (void)hipStreamSynchronize(s); fprintf(stderr, "sah n=%u: phase1 %.3f ms (%d levels), phase2 %.3f ms (%u small nodes)\n", n, msf(t_start, t_p1), levels, msf(t_p1, now()), n_small); }
    // BVH2 -> Node4: internal nodes at even depth, children = grandchildren
    hipLaunchKernelGGL(k_flag_nodes, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, ctr, nodes, flag4);
    size_t cub_bytes = L.cub_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(w + L.cub, cub_bytes, flag4, idx4, (int)L.node_cap, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit_nodes, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, ctr, nodes, flag4, idx4, nodes_out, node_count_out);
    return hipGetLastError();
}

void launch_refit_setup(hipStream_t s, const Node4* nodes, uint32_t n_nodes, uint32_t* parent_slot, uint32_t* n_internal)
{
    if (n_nodes) hipLaunchKernelGGL(k_refit_setup, dim3(blocks(n_nodes)), dim3(kBlock), 0, s, nodes, n_nodes, parent_slot, n_internal);
}
void launch_refit(hipStream_t s, Node4* nodes, uint32_t n_nodes, const rfw_rt_triangle* tris, const uint32_t* order, const uint32_t* parent_slot,
                  const uint32_t* n_internal, uint32_t* arrive)
{
    if (!n_nodes) return;
    (void)hipMemsetAsync(arrive, 0, (size_t)n_nodes * 4, s);
    hipLaunchKernelGGL(k_refit_leaves, dim3(blocks(4 * n_nodes)), dim3(kBlock), 0, s, nodes, n_nodes, tris, order);
    hipLaunchKernelGGL(k_refit_up, dim3(blocks(n_nodes)), dim3(kBlock), 0, s, nodes, n_nodes, parent_slot, n_internal, arrive);
}

} // namespace rfwhip
