// sah_build.hip — binned-SAH BVH construction on the device (replaces the reference's per-mesh rtbvh builds on rayon,
// backends/gpu-rt/src/lib.rs:1345-1383, and BinnedSahBuilder, :1576-1581).
//
// Top-down, in two phases, ONE host read-back per build (round 2: one per level of the upper tree; 1 M triangles 26 ms -> 4.6 ms of kernels):
//   1. nodes with more than 512 primitives (256 for small meshes), one LEVEL per round of three kernels, launched for a number of levels the
//      host derives from the primitive count (the kernels of a level without such nodes return at once):
//        bin        16 bins x 3 axes per node.  One workgroup per 256 positions; a node of this phase spans more than 256 positions, so a block
//                   meets at most two: both get bins in LDS, filled with wave-aggregated adds (meshes arrive in a spatially coherent order: most
//                   lanes of a wavefront share a bin, and 64 LDS atomics on one word take 64 turns), flushed with <= 336 atomics per node and
//                   block into one of up to 32 COPIES of the node's bins (device-scope atomics on one word serialise at ~40 ns each);
//        split      ONE WAVEFRONT per node: a row of 16 lanes is an axis, every lane loads its bin, prefix / suffix scans along the row by DPP
//                   give the boxes left and right of all 48 planes at once; arg-min by shuffles; the winner creates the children and files
//                   them (next level's list, or the queue of phase 2);
//        partition  slots and centroid bounds aggregated per workgroup in LDS (2 + 12 atomics per 256-block); on the upper levels a workgroup
//                   takes 8 blocks, counts first and takes its slots with ONE atomic per child.
//      The primitives' BOXES travel with the order (position-ordered copy, written by partition with wave-contiguous stores): bin and
//      partition read them coalesced instead of gathering 32 B per primitive through the order twice a level.
//   2. every queued range is finished by ONE WORKGROUP in LDS: stage A splits, all threads together, the sub-ranges above 64 primitives;
//      stage B lets every wavefront finish sub-ranges of <= 64 alone, wave-synchronously, with the same row-scan sweep.  (Round 2: ranges of
//      256, one lane per axis sweeping the bins, one split at a time: 12.5 of the builder's 26 ms.)
// The result is a BVH2 with multi-primitive leaves; every internal node at even depth becomes a 4-wide node whose children are
// its grandchildren (as lbvh.hip does), in the Node4 layout the traversal kernels' quantiser consumes.
// The order of primitives inside a leaf depends on atomics and is not reproducible run to run; ray results do not depend on
// the tree (padded boxes, exact-tie rule), so images are still bit-identical to the oracle.
#include "sah_build.h"
#include "env_switches.h"

#include <hipcub/hipcub.hpp>

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace rfwhip {
namespace {

constexpr int kBlock = 256;
constexpr int kBins = 16;
constexpr uint32_t kSmall = 512;   // primitives one workgroup finishes in LDS (one per thread): phase 1 hands a range over at this size ...
constexpr uint32_t kSmallFew = 256; // ... or at this size when the mesh is small: more, shorter workgroups (a 5120-triangle mesh: 30 instead of 7)
inline uint32_t small_cap_for(uint32_t n) { return n <= 32768u ? kSmallFew : kSmall; }
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
    uint32_t n_small;        // ranges queued for phase 2
    uint32_t root_bounds[12];
};

// wavefront-wide min / max by DPP (row shifts inside the 16-lane rows, then the two row broadcasts: six vector instructions, no LDS
// crossbar as __shfl would use); the result is returned to every lane
#define RFW_DPP_STEP(OP, v, ctrl) v = OP(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false)))
__device__ inline float wave_min(float v)
{
    RFW_DPP_STEP(fminf, v, 0x111); RFW_DPP_STEP(fminf, v, 0x112); RFW_DPP_STEP(fminf, v, 0x114); RFW_DPP_STEP(fminf, v, 0x118); // row_shr:1,2,4,8
    RFW_DPP_STEP(fminf, v, 0x142); RFW_DPP_STEP(fminf, v, 0x143);                                                                 // row_bcast:15, row_bcast:31
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ inline float wave_max(float v)
{
    RFW_DPP_STEP(fmaxf, v, 0x111); RFW_DPP_STEP(fmaxf, v, 0x112); RFW_DPP_STEP(fmaxf, v, 0x114); RFW_DPP_STEP(fmaxf, v, 0x118);
    RFW_DPP_STEP(fmaxf, v, 0x142); RFW_DPP_STEP(fmaxf, v, 0x143);
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
#undef RFW_DPP_STEP

// A wavefront's primitives into the 48 bins of a node.  Meshes come in a spatially coherent order, so most lanes of a wavefront share a bin
// on every axis and 64 atomics on one word would take 64 turns (LDS) or 64 trips to the memory side (HBM): the lanes of one bin are reduced
// in registers first and ONE lane adds the result; bins that only a few lanes hit take their atomics directly.
template <class BinPtr> __device__ inline void wave_add_to_bins(BinPtr bins, const bool act, const int* my_bin, const float* lo, const float* hi)
{
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long active = __ballot(act);
    if (active == 0ull) return;
    const int lead = __ffsll((long long)active) - 1;
    for (int a = 0; a < 3; a++) {
        // the bin of the first active lane: when at least 24 lanes share it they are reduced in registers and one lane adds the result
        // (upper levels, primitives still in mesh order); everybody else — every lane, when the bins are spread — takes its atomics directly
        const int b = __builtin_amdgcn_readlane(my_bin[a], lead);
        const bool sel = act && my_bin[a] == b;
        const unsigned long long m = __ballot(sel);
        const uint32_t cnt = (uint32_t)__popcll(m);
        const bool aggregate = cnt >= 24u; // wave-uniform
        if (aggregate) {
            float rl[3], rh[3];
            for (int c = 0; c < 3; c++) { rl[c] = wave_min(sel ? lo[c] : INFINITY); rh[c] = wave_max(sel ? hi[c] : -INFINITY); }
            if ((int)lane == lead) {
                auto* e = &bins[a * kBins + b];
                atomicAdd(&e->count, cnt);
                for (int c = 0; c < 3; c++) { atomicMin(&e->lo[c], f_order(rl[c])); atomicMax(&e->hi[c], f_order(rh[c])); }
            }
        }
        if (act && !(aggregate && sel)) {
            auto* e = &bins[a * kBins + my_bin[a]];
            atomicAdd(&e->count, 1u);
            for (int c = 0; c < 3; c++) { atomicMin(&e->lo[c], f_order(lo[c])); atomicMax(&e->hi[c], f_order(hi[c])); }
        }
    }
}

// ---------------------------------------------------------------- root
__global__ void k_root_init(Counters* ctr)
{
    if (threadIdx.x < 6) ctr->root_bounds[threadIdx.x] = threadIdx.x < 3 ? 0xffffffffu : 0u;          // box
    else if (threadIdx.x < 12) ctr->root_bounds[threadIdx.x] = threadIdx.x < 9 ? 0xffffffffu : 0u;    // centroids
    if (threadIdx.x == 0) { ctr->node_count = 1; ctr->n_active[0] = 0; ctr->n_active[1] = 0; ctr->n_small = 0; }
}
// bounds of all boxes and of their centroids: grid-stride, reduced per wavefront by shuffles and per workgroup in LDS, 12 atomics per workgroup
// (round 2: 12 per wavefront on the same 12 words — 1.1 ms for 720 k primitives)
__global__ __launch_bounds__(kBlock) void k_root_bounds(const DevBox* __restrict__ boxes, uint32_t n, Counters* ctr, uint32_t* order, uint32_t* node_of_pos)
{
    __shared__ uint32_t s_b[12];
    if (threadIdx.x < 12) s_b[threadIdx.x] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        order[i] = i;
        node_of_pos[i] = 0;
        for (int a = 0; a < 3; a++) {
            const float l = boxes[i].lo[a], h = boxes[i].hi[a], c = 0.5f * (l + h);
            lo[a] = fminf(lo[a], l); hi[a] = fmaxf(hi[a], h);
            clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
        }
    }
    for (int a = 0; a < 3; a++) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); clo[a] = wave_min(clo[a]); chi[a] = wave_max(chi[a]); }
    if ((threadIdx.x & 63) == 0)
        for (int a = 0; a < 3; a++) {
            atomicMin(&s_b[a], f_order(lo[a])); atomicMax(&s_b[3 + a], f_order(hi[a]));
            atomicMin(&s_b[6 + a], f_order(clo[a])); atomicMax(&s_b[9 + a], f_order(chi[a]));
        }
    __syncthreads();
    if (threadIdx.x < 12) {
        if ((threadIdx.x % 6) < 3) atomicMin(&ctr->root_bounds[threadIdx.x], s_b[threadIdx.x]);
        else atomicMax(&ctr->root_bounds[threadIdx.x], s_b[threadIdx.x]);
    }
}
// stamp[node] = the level at which the node is split (it is "active" on that level only); 255 = never (a range of phase 2, or not a node)
__global__ void k_root_node(uint32_t n, Counters* ctr, SNode* nodes, uint32_t* active, uint32_t* small, uint32_t* bin_slot, uint8_t* stamp, Bin* bins, uint32_t replicas,
                            uint32_t small_cap)
{
    if (n > small_cap) { // the root's bins (every replica)
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        for (uint32_t k = threadIdx.x; k < replicas * 3 * kBins; k += blockDim.x) bins[k] = e;
    }
    if (threadIdx.x != 0) return;
    SNode r;
    for (int a = 0; a < 3; a++) {
        r.lo[a] = f_unorder(ctr->root_bounds[a]); r.hi[a] = f_unorder(ctr->root_bounds[3 + a]);
        r.cb[a] = ctr->root_bounds[6 + a]; r.cb[3 + a] = ctr->root_bounds[9 + a];
    }
    r.first = 0; r.count = n; r.left = kNone; r.parent = kNone;
    nodes[0] = r;
    if (n > small_cap) {
        active[0] = 0; bin_slot[0] = 0; stamp[0] = 0; ctr->n_active[0] = 1;
    } else {
        small[0] = 0; ctr->n_small = 1;
    }
}

// ---------------------------------------------------------------- phase 1: one level
// One workgroup per 256 positions of the primitive order.  A node that is split on this level has more than small_cap >= 256 primitives in
// consecutive positions, so a block meets at most TWO such nodes: both get a set of bins in LDS (wave-aggregated adds, see above), and the
// block flushes what it gathered into the nodes' bins in HBM — 336 atomics per node and block at most, into copy blockIdx.x % replicas.
__global__ __launch_bounds__(kBlock) void k_bin(const DevBox* __restrict__ pbox, const uint32_t* __restrict__ node_of_pos, const SNode* __restrict__ nodes,
                                               const uint32_t* __restrict__ bin_slot, const uint8_t* __restrict__ stamp, Bin* bins, uint32_t n,
                                               Counters* ctr, uint32_t level, uint32_t replicas)
{
    // On the upper levels thousands of workgroups flush into the bins of a handful of nodes, and device-scope atomics on one word serialise
    // at the memory side (~40 ns each): every node of such a level has `replicas` copies of its bins, a workgroup adds to copy
    // blockIdx.x % replicas, k_split sums the copies.
    const uint32_t par = level & 1u;
    if (blockIdx.x == 0 && threadIdx.x == 0) ctr->n_active[par ^ 1u] = 0; // the level before this one is done with it; k_split counts the next level's nodes into it
    if (ctr->n_active[par] == 0u) return; // a level past the last one with big nodes
    __shared__ Bin sb[2][3 * kBins];
    __shared__ uint32_t s_lo, s_hi;
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    uint32_t nd = p < n ? node_of_pos[p] : kNone;
    DevBox b;
    if (p < n) b = pbox[p]; // issued before the node is known to be active: the load overlaps the dependent ones below
    if (nd != kNone && stamp[nd] != (uint8_t)level) nd = kNone;
    if (threadIdx.x == 0) { s_lo = kNone; s_hi = 0u; }
    if (threadIdx.x < 2 * 3 * kBins) {
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        sb[threadIdx.x / (3 * kBins)][threadIdx.x % (3 * kBins)] = e;
    }
    __syncthreads();
    if (nd != kNone) { atomicMin(&s_lo, nd); atomicMax(&s_hi, nd); }
    __syncthreads();
    const uint32_t node_a = s_lo, node_b = s_hi; // the (at most two) active nodes of this block, by id
    if (node_a == kNone) return; // nothing active in these 256 positions (uniform)
    int my_bin[3] = {0, 0, 0};
    if (nd != kNone) {
        const SNode& node = nodes[nd];
        for (int a = 0; a < 3; a++) my_bin[a] = bin_of(0.5f * (b.lo[a] + b.hi[a]), f_unorder(node.cb[a]), f_unorder(node.cb[3 + a]));
    }
    wave_add_to_bins(sb[0], nd == node_a, my_bin, b.lo, b.hi);
    if (node_b != node_a) wave_add_to_bins(sb[1], nd == node_b, my_bin, b.lo, b.hi); // (uniform)
    __syncthreads();
    if (threadIdx.x < 2 * 3 * kBins) {
        const uint32_t which = threadIdx.x / (3 * kBins), k = threadIdx.x % (3 * kBins);
        const Bin v = sb[which][k];
        if (v.count && (which == 0 || node_b != node_a)) {
            Bin* e = bins + ((size_t)bin_slot[which ? node_b : node_a] * replicas + blockIdx.x % replicas) * 3 * kBins + k;
            atomicAdd(&e->count, v.count);
            for (int c = 0; c < 3; c++) { atomicMin(&e->lo[c], v.lo[c]); atomicMax(&e->hi[c], v.hi[c]); }
        }
    }
}

// the SAH sweep over one node's 48 bins by 48 lanes of a wavefront (lane = 16 * axis + plane; plane 15 is no split): every lane gathers the
// boxes left and right of its plane, the cheapest lane wins (ties: the lowest lane, i.e. the order a serial sweep would find them in)
struct SplitChoice {
    int axis, plane;
    uint32_t left_count;
    float llo[3], lhi[3], rlo[3], rhi[3];
    float cost;
};
// returns the winning lane (or -1: every centroid in one bin on every axis); `mine` is this lane's own candidate.
// A row of 16 lanes is one axis: every lane loads ITS bin, an inclusive prefix scan along the row (DPP row shifts: no LDS crossbar) gives the
// box and the count left of each plane, a suffix scan shifted by one the right side — 7 loads and ~70 row operations per lane instead of a
// loop over the 16 bins of the axis (112 loads).
#define RFW_ROW_MIN(v, ctrl) v = fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false)))
#define RFW_ROW_MAX(v, ctrl) v = fmaxf(v, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(v), __float_as_int(v), ctrl, 0xf, 0xf, false)))
#define RFW_ROW_ADD(v, ctrl) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, 0xf, 0xf, false)
template <typename BinPtr> __device__ inline int sweep48(BinPtr bins, const uint32_t lane, SplitChoice& mine)
{
    mine.axis = -1; mine.plane = -1; mine.left_count = 0; mine.cost = INFINITY;
    float llo[3] = {INFINITY, INFINITY, INFINITY}, lhi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t cl = 0;
    if (lane < 48u) {
        const Bin e = bins[lane];
        if (e.count) { cl = e.count; for (int c = 0; c < 3; c++) { llo[c] = f_unorder(e.lo[c]); lhi[c] = f_unorder(e.hi[c]); } }
    }
    float rlo[3] = {llo[0], llo[1], llo[2]}, rhi[3] = {lhi[0], lhi[1], lhi[2]};
    uint32_t cr = cl;
    // inclusive prefix along the row (row_shr 1, 2, 4, 8) and inclusive suffix (row_shl 1, 2, 4, 8); lanes without a source keep their value
    for (int c = 0; c < 3; c++) {
        RFW_ROW_MIN(llo[c], 0x111); RFW_ROW_MIN(llo[c], 0x112); RFW_ROW_MIN(llo[c], 0x114); RFW_ROW_MIN(llo[c], 0x118);
        RFW_ROW_MAX(lhi[c], 0x111); RFW_ROW_MAX(lhi[c], 0x112); RFW_ROW_MAX(lhi[c], 0x114); RFW_ROW_MAX(lhi[c], 0x118);
        RFW_ROW_MIN(rlo[c], 0x101); RFW_ROW_MIN(rlo[c], 0x102); RFW_ROW_MIN(rlo[c], 0x104); RFW_ROW_MIN(rlo[c], 0x108);
        RFW_ROW_MAX(rhi[c], 0x101); RFW_ROW_MAX(rhi[c], 0x102); RFW_ROW_MAX(rhi[c], 0x104); RFW_ROW_MAX(rhi[c], 0x108);
    }
    RFW_ROW_ADD(cl, 0x111); RFW_ROW_ADD(cl, 0x112); RFW_ROW_ADD(cl, 0x114); RFW_ROW_ADD(cl, 0x118);
    RFW_ROW_ADD(cr, 0x101); RFW_ROW_ADD(cr, 0x102); RFW_ROW_ADD(cr, 0x104); RFW_ROW_ADD(cr, 0x108);
    // the right side of plane b = the suffix of bin b + 1: one more shift (the last lane of a row gets "nothing")
    for (int c = 0; c < 3; c++) {
        rlo[c] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(INFINITY), __float_as_int(rlo[c]), 0x101, 0xf, 0xf, false));
        rhi[c] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(-INFINITY), __float_as_int(rhi[c]), 0x101, 0xf, 0xf, false));
    }
    cr = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)cr, 0x101, 0xf, 0xf, false);
    for (int c = 0; c < 3; c++) { mine.llo[c] = llo[c]; mine.lhi[c] = lhi[c]; mine.rlo[c] = rlo[c]; mine.rhi[c] = rhi[c]; }
    if (lane < 48u && (lane & 15u) < (uint32_t)kBins - 1u && cl != 0u && cr != 0u) {
        mine.axis = (int)(lane >> 4); mine.plane = (int)(lane & 15u); mine.left_count = cl;
        mine.cost = (float)cl * half_area(llo, lhi) + (float)cr * half_area(rlo, rhi);
    }
    // arg-min over the wavefront: (cost, lane), lowest lane among equal costs
    float c = mine.cost;
    int who = mine.axis >= 0 ? (int)lane : 64;
    for (int off = 32; off > 0; off >>= 1) {
        const float oc = __shfl_xor(c, off);
        const int ow = __shfl_xor(who, off);
        if (oc < c || (oc == c && ow < who)) { c = oc; who = ow; }
    }
    return who < 64 ? who : -1;
}
#undef RFW_ROW_MIN
#undef RFW_ROW_MAX
#undef RFW_ROW_ADD

__device__ inline void init_child(SNode& c, uint32_t first, uint32_t count, const float* lo, const float* hi, uint32_t parent)
{
    for (int a = 0; a < 3; a++) { c.lo[a] = lo[a]; c.hi[a] = hi[a]; c.cb[a] = 0xffffffffu; c.cb[3 + a] = 0u; }
    c.first = first; c.count = count; c.left = kNone; c.parent = parent;
}

// one wavefront per active node; the winning lane creates the two children and files each of them: more than kSmall primitives -> the next
// level's list (its bins cleared by the wavefront), else the queue of phase 2
__global__ __launch_bounds__(kBlock) void k_split(const uint32_t* __restrict__ active, uint32_t* active_out, uint32_t level, Counters* ctr, SNode* nodes,
                                                 const Bin* __restrict__ bins, Bin* bins_next, uint32_t* bin_slot, uint8_t* stamp, Split* splits, uint32_t* fill, uint32_t* small, uint32_t replicas,
                                                 uint32_t replicas_next, uint32_t small_cap)
{
    __shared__ Bin s_sum[kBlock / 64][3 * kBins];
    const uint32_t par = level & 1u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, k = blockIdx.x * (kBlock / 64) + wave;
    if (k >= ctr->n_active[par]) return; // (wave-uniform; no workgroup barrier below)
    const uint32_t nd = active[k];
    if (lane < 3 * kBins) { // the node's bins: the sum of its replicas, into LDS (read 16 times each by the sweep)
        const Bin* nb = bins + (size_t)bin_slot[nd] * replicas * 3 * kBins + lane;
        Bin acc = nb[0];
        for (uint32_t r = 1; r < replicas; r++) {
            const Bin e = nb[(size_t)r * 3 * kBins];
            acc.count += e.count;
            for (int c = 0; c < 3; c++) { acc.lo[c] = e.lo[c] < acc.lo[c] ? e.lo[c] : acc.lo[c]; acc.hi[c] = e.hi[c] > acc.hi[c] ? e.hi[c] : acc.hi[c]; }
        }
        s_sum[wave][lane] = acc;
    }
    __builtin_amdgcn_wave_barrier(); // (LDS operations of one wavefront complete in order; this only keeps the compiler from moving the reads up)
    SplitChoice s;
    const int win = sweep48(s_sum[wave], lane, s);
    const uint32_t writer = win >= 0 ? (uint32_t)win : 0u; // the winning lane (lane 0 when there is no split) writes the children
    uint32_t slot_l = kNone, slot_r = kNone;
    if (lane == writer) {
        const SNode node = nodes[nd];
        const uint32_t li = atomicAdd(&ctr->node_count, 2u);
        SNode l, r;
        uint32_t code;
        if (win >= 0) {
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
        for (uint32_t c = 0; c < 2; c++) {
            const uint32_t id = li + c, cnt = c ? r.count : l.count;
            if (cnt > small_cap) {
                const uint32_t slot = atomicAdd(&ctr->n_active[par ^ 1u], 1u);
                active_out[slot] = id;
                bin_slot[id] = slot;
                stamp[id] = (uint8_t)(level + 1u);
                (c ? slot_r : slot_l) = slot;
            } else {
                small[atomicAdd(&ctr->n_small, 1u)] = id;
            }
        }
    }
    // the bins of the children that go on: cleared by 48 lanes each.  They live in the OTHER of two bin arrays (by level parity): the
    // wavefronts of this launch run in any order, and a child's slot on the next level says nothing about whose bins of THIS level it
    // would overwrite (found as a memory fault / hang with two processes sharing the device: a late wavefront read bins an early one had cleared)
    slot_l = (uint32_t)__shfl((int)slot_l, (int)writer);
    slot_r = (uint32_t)__shfl((int)slot_r, (int)writer);
    {
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        for (uint32_t q = lane; q < replicas_next * 3 * kBins; q += 64u) {
            if (slot_l != kNone) bins_next[(size_t)slot_l * replicas_next * 3 * kBins + q] = e;
            if (slot_r != kNone) bins_next[(size_t)slot_r * replicas_next * 3 * kBins + q] = e;
        }
    }
}

// Where a 256-block of positions lies inside ONE active node (the upper levels, where the atomics of thousands of wavefronts would meet on
// a handful of words) slots and centroid bounds are aggregated over the workgroup: 2 + 12 atomics per block; blocks that hold several
// nodes aggregate per wavefront, lanes of mixed wavefronts go alone.
struct PartShared {
    uint32_t node, mixed, cnt[kBlock / 64][2], base[2], cb[2][6];
};
__device__ inline void partition_block(PartShared& S, const uint32_t vblock, const DevBox* __restrict__ pbox_in, DevBox* __restrict__ pbox_out,
                                       const uint32_t* __restrict__ order_in, const uint32_t* __restrict__ nop_in, uint32_t* order_out, uint32_t* nop_out, SNode* nodes,
                                       const Split* __restrict__ splits, const uint8_t* __restrict__ stamp, uint32_t* fill, uint32_t n, uint32_t level)
{
    uint32_t& s_node = S.node; uint32_t& s_mixed = S.mixed;
    uint32_t (&s_cnt)[kBlock / 64][2] = S.cnt; uint32_t (&s_base)[2] = S.base; uint32_t (&s_cb)[2][6] = S.cb;
    __syncthreads(); // (a caller that walks several blocks: nobody is still reading the previous block's shared values)
    const uint32_t p = vblock * kBlock + threadIdx.x;
    const uint32_t nd = p < n ? nop_in[p] : kNone;
    const uint32_t prim = p < n ? order_in[p] : 0u;
    const bool act = nd != kNone && stamp[nd] == (uint8_t)level;
    if (p < n && !act) { // finished ranges keep their place (their boxes are not needed again: phase 2 reads the builder's input)
        order_out[p] = prim;
        nop_out[p] = nd;
    }
    if (threadIdx.x == 0) { s_node = kNone; s_mixed = 0; }
    if (threadIdx.x < 12) s_cb[threadIdx.x / 6][threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    __syncthreads();
    if (act) atomicMin(&s_node, nd);
    __syncthreads();
    const uint32_t block_node = s_node;
    if (block_node == kNone) return; // nothing active here (uniform)
    if (act && nd != block_node) s_mixed = 1;
    __syncthreads();
    const bool block_single = s_mixed == 0;
    // every lane stays in the kernel: the wave-level reductions below need the whole wavefront
    uint32_t child = kNone;
    float c3[3] = {0.0f, 0.0f, 0.0f};
    DevBox b;
    if (act) {
        const SNode& node = nodes[nd];
        const uint32_t code = splits[nd].axis_plane;
        b = pbox_in[p];
        for (int a = 0; a < 3; a++) c3[a] = 0.5f * (b.lo[a] + b.hi[a]);
        bool left;
        if (code >> 8) left = (p - node.first) < node.count / 2;
        else {
            const int a = (int)(code & 3u);
            left = bin_of(c3[a], f_unorder(node.cb[a]), f_unorder(node.cb[3 + a])) <= (int)((code >> 2) & 63u);
        }
        child = node.left + (left ? 0u : 1u);
    }
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long actmask = __ballot(act);
    uint32_t dest = 0;
    if (block_single) {
        const uint32_t li = nodes[block_node].left;
        const bool to_right = act && child != li;
        const unsigned long long rmask = __ballot(to_right), lmask = actmask & ~rmask;
        const unsigned long long mine = to_right ? rmask : lmask;
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        if (lane == 0) { s_cnt[wave][0] = (uint32_t)__popcll(lmask); s_cnt[wave][1] = (uint32_t)__popcll(rmask); }
        for (int side = 0; side < 2; side++) { // centroid bounds of the two children: wavefront, then workgroup (LDS), then 12 atomics per block
            const unsigned long long m = side ? rmask : lmask;
            if (m == 0ull) continue; // wave-uniform
            const bool in = act && (to_right == (side != 0));
            for (int a = 0; a < 3; a++) {
                const float lo = wave_min(in ? c3[a] : INFINITY), hi = wave_max(in ? c3[a] : -INFINITY);
                if (lane == 0) { atomicMin(&s_cb[side][a], f_order(lo)); atomicMax(&s_cb[side][3 + a], f_order(hi)); }
            }
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            uint32_t tot = 0;
            for (int w = 0; w < kBlock / 64; w++) tot += s_cnt[w][threadIdx.x];
            s_base[threadIdx.x] = tot ? atomicAdd(&fill[li + threadIdx.x], tot) : 0u;
        }
        if (threadIdx.x >= 64 && threadIdx.x < 76) {
            const uint32_t k = threadIdx.x - 64, side = k / 6, c = k % 6;
            uint32_t tot = 0;
            for (int w = 0; w < kBlock / 64; w++) tot += s_cnt[w][side];
            if (tot) {
                if (c < 3) atomicMin(&nodes[li + side].cb[c], s_cb[side][c]);
                else atomicMax(&nodes[li + side].cb[c], s_cb[side][c]);
            }
        }
        __syncthreads();
        if (act) {
            const uint32_t side = to_right ? 1u : 0u;
            uint32_t off = s_base[side];
            for (uint32_t w = 0; w < wave; w++) off += s_cnt[w][side];
            dest = off + rank;
        }
    } else {
        if (actmask == 0ull) return;
        const int lead = __ffsll((long long)actmask) - 1;
        const uint32_t lead_node = (uint32_t)__shfl((int)nd, lead);
        const unsigned long long same = __ballot(act && nd == lead_node);
        if (same == actmask) {
            // every active lane of the wave belongs to one node -> per child one atomic for the slots and six for the centroid bounds
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
                for (int a = 0; a < 3; a++) {
                    const float lo = wave_min(in ? c3[a] : INFINITY), hi = wave_max(in ? c3[a] : -INFINITY);
                    if ((int)lane == lead) { atomicMin(&nodes[li + side].cb[a], f_order(lo)); atomicMax(&nodes[li + side].cb[3 + a], f_order(hi)); }
                }
            }
        } else if (act) {
            dest = atomicAdd(&fill[child], 1u);
            for (int a = 0; a < 3; a++) { atomicMin(&nodes[child].cb[a], f_order(c3[a])); atomicMax(&nodes[child].cb[3 + a], f_order(c3[a])); }
        }
    }
    if (act) {
        dest += nodes[child].first;
        order_out[dest] = prim;
        nop_out[dest] = child;
        pbox_out[dest] = b;
    }
}

__global__ __launch_bounds__(kBlock) void k_partition(const DevBox* __restrict__ pbox_in, DevBox* __restrict__ pbox_out, const uint32_t* __restrict__ order_in,
                                                     const uint32_t* __restrict__ nop_in, uint32_t* order_out, uint32_t* nop_out, SNode* nodes,
                                                     const Split* __restrict__ splits, const uint8_t* __restrict__ stamp, uint32_t* fill, uint32_t n, uint32_t level)
{
    __shared__ PartShared S;
    partition_block(S, blockIdx.x, pbox_in, pbox_out, order_in, nop_in, order_out, nop_out, nodes, splits, stamp, fill, n, level);
}

// The same for the UPPER levels, where thousands of 256-blocks lie inside the same few nodes and their atomics (2 for the slots, 12 for the
// centroid bounds per block) meet on the same words, ~40 ns apiece at the memory side: a workgroup takes kPartChunk consecutive blocks; when they
// all lie inside ONE node that is split on this level it counts first (one pass: sides as bits in a register, counts per block and wavefront
// in LDS, centroid bounds reduced over the whole chunk), takes its slots with ONE atomic per child, and writes in a second pass (the boxes
// come from L2 then).  A chunk that crosses nodes goes block by block through the generic code.
constexpr int kPartChunk = 8;
__global__ __launch_bounds__(kBlock) void k_partition_chunk(const DevBox* __restrict__ pbox_in, DevBox* __restrict__ pbox_out, const uint32_t* __restrict__ order_in,
                                                           const uint32_t* __restrict__ nop_in, uint32_t* order_out, uint32_t* nop_out, SNode* nodes,
                                                           const Split* __restrict__ splits, const uint8_t* __restrict__ stamp, uint32_t* fill, uint32_t n, uint32_t level)
{
    __shared__ PartShared S;
    __shared__ uint32_t s_cnt[kPartChunk][kBlock / 64][2], s_tot[2], s_cbu[2][6];
    const uint32_t begin = blockIdx.x * kPartChunk * kBlock, end = begin + kPartChunk * kBlock < n ? begin + kPartChunk * kBlock : n;
    const uint32_t nd0 = nop_in[begin], nd1 = nop_in[end - 1];
    const bool one_node = nd0 == nd1 && nd0 != kNone && stamp[nd0] == (uint8_t)level; // (uniform; a node's positions are consecutive)
    if (!one_node) {
        for (uint32_t vb = begin / kBlock; vb * kBlock < end; vb++)
            partition_block(S, vb, pbox_in, pbox_out, order_in, nop_in, order_out, nop_out, nodes, splits, stamp, fill, n, level);
        return;
    }
    const SNode node = nodes[nd0];
    const uint32_t code = splits[nd0].axis_plane, li = node.left;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    if (threadIdx.x < 12) s_cbu[threadIdx.x / 6][threadIdx.x % 6] = (threadIdx.x % 6) < 3 ? 0xffffffffu : 0u;
    // ---- pass 1: sides and counts
    uint32_t right_bits = 0; // bit j: my primitive of block j goes right
    float clo[2][3], chi[2][3];
    for (int sd = 0; sd < 2; sd++) for (int a = 0; a < 3; a++) { clo[sd][a] = INFINITY; chi[sd][a] = -INFINITY; }
    for (int j = 0; j < kPartChunk; j++) {
        const uint32_t p = begin + (uint32_t)j * kBlock + threadIdx.x;
        bool in = p < end, right = false;
        if (in) {
            const DevBox b = pbox_in[p];
            float c3[3];
            for (int a = 0; a < 3; a++) c3[a] = 0.5f * (b.lo[a] + b.hi[a]);
            bool left;
            if (code >> 8) left = (p - node.first) < node.count / 2;
            else {
                const int a = (int)(code & 3u);
                left = bin_of(c3[a], f_unorder(node.cb[a]), f_unorder(node.cb[3 + a])) <= (int)((code >> 2) & 63u);
            }
            right = !left;
            const int sd = right ? 1 : 0;
            for (int a = 0; a < 3; a++) { clo[sd][a] = fminf(clo[sd][a], c3[a]); chi[sd][a] = fmaxf(chi[sd][a], c3[a]); }
        }
        const unsigned long long rm = __ballot(in && right), lm = __ballot(in && !right);
        if (right) right_bits |= 1u << j;
        if (lane == 0) { s_cnt[j][wave][0] = (uint32_t)__popcll(lm); s_cnt[j][wave][1] = (uint32_t)__popcll(rm); }
    }
    for (int sd = 0; sd < 2; sd++)
        for (int a = 0; a < 3; a++) {
            const float l = wave_min(clo[sd][a]), h = wave_max(chi[sd][a]);
            if (lane == 0 && l <= h) { atomicMin(&s_cbu[sd][a], f_order(l)); atomicMax(&s_cbu[sd][3 + a], f_order(h)); }
        }
    __syncthreads();
    if (threadIdx.x < 2) {
        uint32_t tot = 0;
        for (int j = 0; j < kPartChunk; j++)
            for (int w = 0; w < kBlock / 64; w++) tot += s_cnt[j][w][threadIdx.x];
        s_tot[threadIdx.x] = tot ? atomicAdd(&fill[li + threadIdx.x], tot) : 0u; // the chunk's slots in the child's range
    }
    if (threadIdx.x >= 64 && threadIdx.x < 76) {
        const uint32_t k = threadIdx.x - 64, sd = k / 6, c = k % 6;
        const uint32_t v = s_cbu[sd][c];
        if (c < 3) { if (v != 0xffffffffu) atomicMin(&nodes[li + sd].cb[c], v); }
        else if (v != 0u) atomicMax(&nodes[li + sd].cb[c], v);
    }
    __syncthreads();
    // ---- pass 2: every primitive to its slot (chunk order is kept on both sides)
    const uint32_t first_l = nodes[li].first, first_r = nodes[li + 1].first;
    uint32_t run[2] = {s_tot[0], s_tot[1]}; // slots taken by the blocks and wavefronts before mine
    for (int j = 0; j < kPartChunk; j++) {
        const uint32_t p = begin + (uint32_t)j * kBlock + threadIdx.x;
        const bool in = p < end, right = (right_bits >> j) & 1u;
        const unsigned long long rm = __ballot(in && right), lm = __ballot(in && !right);
        uint32_t off[2] = {run[0], run[1]};
        for (uint32_t w = 0; w < wave; w++) { off[0] += s_cnt[j][w][0]; off[1] += s_cnt[j][w][1]; }
        if (in) {
            const unsigned long long m = right ? rm : lm;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t dest = (right ? first_r + off[1] : first_l + off[0]) + rank;
            order_out[dest] = order_in[p];
            nop_out[dest] = li + (right ? 1u : 0u);
            pbox_out[dest] = pbox_in[p];
        }
        for (int w = 0; w < kBlock / 64; w++) { run[0] += s_cnt[j][w][0]; run[1] += s_cnt[j][w][1]; }
    }
}

// ---------------------------------------------------------------- phase 2: one workgroup finishes one range of <= kSmall primitives
// Boxes and the permutation of the range stay in LDS (a primitive per thread).  Two stages:
//   A  the whole workgroup splits, one after the other, the sub-ranges of MORE than kWaveRange (64) primitives: bins by LDS atomics, the same
//      48-lane sweep as k_split by wavefront 0, partition by ballots + a prefix over the wavefronts; sub-ranges of <= 64 go on a list;
//   B  every wavefront takes sub-ranges off that list and finishes each alone, wave-synchronously (no workgroup barrier): a primitive per lane,
//      its own bins, the larger child on a small stack.
// (Round 2 finished ranges of 256 with one lane per axis sweeping the bins and one split at a time: 12.5 of the builder's 26 ms; the first
// version of this round handed over at 64 primitives and paid for five more LEVELS of phase 1, ~0.5 ms each at 720 k primitives.)
constexpr uint32_t kWaveRange = 64;
constexpr int kListCap = 96;

__device__ inline void wave_sync()
{
    // LDS operations of one wavefront are executed in issue order; this only stops the compiler from moving accesses across
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <uint32_t CAP> struct SmallShared {
    static constexpr int kWaves = (int)(CAP / 64);
    float lo[3][CAP], hi[3][CAP];
    uint16_t perm[2][CAP];
    Bin bins[3 * kBins];               // stage A
    Bin wbins[kWaves][3 * kBins];      // stage B, per wavefront
    uint32_t cb[6];
    SplitChoice choice;
    SplitChoice wchoice[kWaves];
    uint32_t next_id;                  // next unused node id of this workgroup's reservation
    uint32_t wave_cnt[kWaves][2];
    // stage A's stack of ranges still above kWaveRange, and the list of ranges for stage B: first, count, node id, box
    uint32_t stack[3 * 12]; float stack_box[6 * 12];
    uint32_t list[3 * kListCap]; float list_box[6 * kListCap];
    uint32_t list_n, list_head;
    uint32_t wstack[kWaves][3 * 8]; float wstack_box[kWaves][6 * 8];
};

// one wavefront finishes the range [first, first + count) of S.perm[0] (count <= 64) whose node is nid with box nlo / nhi
template <uint32_t CAP> __device__ inline void finish_range_wave(SmallShared<CAP>& S, const uint32_t wave, const uint32_t lane, uint32_t first, uint32_t count, uint32_t nid, float* nlo, float* nhi,
                                         const uint32_t gfirst, SNode* nodes, const int max_leaf, const float trav_cost)
{
    int sp = 0;
    for (;;) {
        if (count == 2u) {
            // Two primitives: the only possible split is one from the other (along the first axis on which their centroids differ, the
            // lower one left), its cost is the two boxes' half areas — no bins, no sweep, and both children are leaves as they stand.  Half
            // of the ranges a binary tree over 64 primitives evaluates have two primitives, and this chain of dependent evaluations is what
            // k_small's time consists of (the kernel is bound by the latency of one workgroup).  Every lane computes the same.
            const uint32_t q0 = S.perm[0][first], q1 = S.perm[0][first + 1u];
            float l0[3], h0[3], l1[3], h1[3];
            bool sep = false, swap = false;
            for (int a = 0; a < 3; a++) {
                l0[a] = S.lo[a][q0]; h0[a] = S.hi[a][q0]; l1[a] = S.lo[a][q1]; h1[a] = S.hi[a][q1];
                const float c0 = 0.5f * (l0[a] + h0[a]), c1 = 0.5f * (l1[a] + h1[a]);
                if (!sep && c0 != c1) { sep = true; swap = c1 < c0; }
            }
            const float area = half_area(nlo, nhi);
            const bool split2 = (sep && half_area(l0, h0) + half_area(l1, h1) + trav_cost * area < 2.0f * area) || 2 > max_leaf;
            if (split2) {
                uint32_t li = 0;
                if (lane == 0) li = atomicAdd(&S.next_id, 2u);
                li = (uint32_t)__builtin_amdgcn_readfirstlane((int)li);
                if (lane == 0) {
                    SNode l, r;
                    init_child(l, gfirst + first, 1u, swap ? l1 : l0, swap ? h1 : h0, nid);
                    init_child(r, gfirst + first + 1u, 1u, swap ? l0 : l1, swap ? h0 : h1, nid);
                    nodes[li] = l;
                    nodes[li + 1] = r;
                    nodes[nid].left = li;
                    if (swap) { S.perm[0][first] = (uint16_t)q1; S.perm[0][first + 1u] = (uint16_t)q0; }
                }
                wave_sync();
            }
            // (a leaf, or two leaves: the next range)
            if (sp == 0) break;
            sp--;
            first = S.wstack[wave][3 * sp + 0];
            count = S.wstack[wave][3 * sp + 1];
            nid = S.wstack[wave][3 * sp + 2];
            for (int a = 0; a < 3; a++) { nlo[a] = S.wstack_box[wave][6 * sp + a]; nhi[a] = S.wstack_box[wave][6 * sp + 3 + a]; }
            wave_sync();
            continue;
        }
        // (tried: ranges of 3 and 4 primitives by an exact search over all their bipartitions, every lane alike from the boxes in LDS — the
        // same trees, and no faster than the bins: 4.13 against 4.10 ms per build)
        const bool mine = lane < count;
        const uint32_t q = mine ? S.perm[0][first + lane] : 0u;
        float cen[3] = {0.0f, 0.0f, 0.0f};
        int my_bin[3] = {0, 0, 0};
        bool split = false;
        int win = -1;
        SplitChoice s;
        if (count > 1) {
            if (lane < 3 * kBins) {
                Bin e; e.count = 0;
                for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
                S.wbins[wave][lane] = e;
            }
            float clo[3], chi[3];
            for (int a = 0; a < 3; a++) {
                cen[a] = mine ? 0.5f * (S.lo[a][q] + S.hi[a][q]) : 0.0f;
                clo[a] = wave_min(mine ? cen[a] : INFINITY);
                chi[a] = wave_max(mine ? cen[a] : -INFINITY);
            }
            wave_sync();
            {
                float blo[3] = {0.0f, 0.0f, 0.0f}, bhi[3] = {0.0f, 0.0f, 0.0f};
                if (mine)
                    for (int a = 0; a < 3; a++) { my_bin[a] = bin_of(cen[a], clo[a], chi[a]); blo[a] = S.lo[a][q]; bhi[a] = S.hi[a][q]; }
                wave_add_to_bins(S.wbins[wave], mine, my_bin, blo, bhi);
            }
            wave_sync();
            win = sweep48(S.wbins[wave], lane, s);
            if ((int)lane == (win >= 0 ? win : 0)) S.wchoice[wave] = s;
            wave_sync();
            s = S.wchoice[wave];
            // ---- decide (uniform): leaf, SAH split, or halve a range whose centroids coincide
            const float area = half_area(nlo, nhi);
            const float leaf_cost = (float)count * area;
            if (win >= 0 && (s.cost + trav_cost * area < leaf_cost || (int)count > max_leaf)) split = true;
            else if ((int)count > max_leaf) { // coincident centroids: arbitrary halves keep leaves bounded
                split = true;
                win = -1;
                s.left_count = count / 2;
                for (int a = 0; a < 3; a++) { s.llo[a] = s.rlo[a] = nlo[a]; s.lhi[a] = s.rhi[a] = nhi[a]; }
            }
        }
        if (split) {
            const uint32_t lc = s.left_count, rc = count - lc;
            uint32_t li = 0;
            if (lane == 0) li = atomicAdd(&S.next_id, 2u);
            li = (uint32_t)__builtin_amdgcn_readfirstlane((int)li);
            if (lane == 0) {
                SNode l, r;
                init_child(l, gfirst + first, lc, s.llo, s.lhi, nid);
                init_child(r, gfirst + first + lc, rc, s.rlo, s.rhi, nid);
                nodes[li] = l;
                nodes[li + 1] = r;
                nodes[nid].left = li;
            }
            // partition the lanes' primitives: left ones first, in lane order on both sides
            const bool left = mine && (win < 0 ? lane < lc : my_bin[s.axis] <= s.plane);
            const unsigned long long lm = __ballot(left), rm = __ballot(mine && !left);
            if (mine) {
                const unsigned long long m = left ? lm : rm;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                S.perm[1][first + (left ? rank : lc + rank)] = (uint16_t)q;
            }
            wave_sync();
            if (mine) S.perm[0][first + lane] = S.perm[1][first + lane];
            // the smaller child is processed next, the larger one waits on the stack: never more than log2(64) entries
            const bool left_next = lc <= rc;
            if (lane == 0) {
                S.wstack[wave][3 * sp + 0] = left_next ? first + lc : first;
                S.wstack[wave][3 * sp + 1] = left_next ? rc : lc;
                S.wstack[wave][3 * sp + 2] = left_next ? li + 1 : li;
                for (int a = 0; a < 3; a++) { S.wstack_box[wave][6 * sp + a] = left_next ? s.rlo[a] : s.llo[a]; S.wstack_box[wave][6 * sp + 3 + a] = left_next ? s.rhi[a] : s.lhi[a]; }
            }
            sp++;
            if (left_next) { count = lc; nid = li; for (int a = 0; a < 3; a++) { nlo[a] = s.llo[a]; nhi[a] = s.lhi[a]; } }
            else { first = first + lc; count = rc; nid = li + 1; for (int a = 0; a < 3; a++) { nlo[a] = s.rlo[a]; nhi[a] = s.rhi[a]; } }
            wave_sync();
            continue;
        }
        // ---- leaf: nodes[nid].left stays kNone; take the next range
        if (sp == 0) break;
        sp--;
        first = S.wstack[wave][3 * sp + 0];
        count = S.wstack[wave][3 * sp + 1];
        nid = S.wstack[wave][3 * sp + 2];
        for (int a = 0; a < 3; a++) { nlo[a] = S.wstack_box[wave][6 * sp + a]; nhi[a] = S.wstack_box[wave][6 * sp + 3 + a]; }
        wave_sync();
    }
}

template <uint32_t CAP> __global__ __launch_bounds__(CAP) void k_small(const DevBox* __restrict__ boxes, const uint32_t* __restrict__ order_in, uint32_t* order_out,
                                                 const uint32_t* __restrict__ small, Counters* ctr, SNode* nodes, int max_leaf, float trav_cost)
{
    __shared__ SmallShared<CAP> S;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (blockIdx.x >= ctr->n_small) return;
    const uint32_t root = small[blockIdx.x];
    const uint32_t gfirst = nodes[root].first, gcount = nodes[root].count;
    if (tid < gcount) {
        const DevBox b = boxes[order_in[gfirst + tid]];
        for (int a = 0; a < 3; a++) { S.lo[a][tid] = b.lo[a]; S.hi[a][tid] = b.hi[a]; }
        S.perm[0][tid] = (uint16_t)tid;
    }
    if (tid == 0) {
        S.next_id = atomicAdd(&ctr->node_count, 2u * gcount); // a subtree over c primitives has at most 2c - 2 nodes below its root
        S.list_n = 0; S.list_head = 0;
        S.stack[0] = 0; S.stack[1] = gcount; S.stack[2] = root;
        for (int a = 0; a < 3; a++) { S.stack_box[a] = nodes[root].lo[a]; S.stack_box[3 + a] = nodes[root].hi[a]; }
    }
    __syncthreads();
    // ---- stage A: the workgroup splits ranges above kWaveRange (everything below is uniform over the workgroup)
    int sp = 1;
    while (sp > 0) {
        sp--;
        uint32_t first = S.stack[3 * sp + 0], count = S.stack[3 * sp + 1], nid = S.stack[3 * sp + 2];
        float nlo[3], nhi[3];
        for (int a = 0; a < 3; a++) { nlo[a] = S.stack_box[6 * sp + a]; nhi[a] = S.stack_box[6 * sp + 3 + a]; }
        __syncthreads(); // everybody has read the entry before somebody overwrites it
        for (;;) { // this range, then its smaller children, for as long as they are above kWaveRange
            const bool list_full = S.list_n >= (uint32_t)kListCap; // (uniform: written behind barriers only)
            if (count <= kWaveRange && !list_full) {
                if ((int)count > max_leaf || count > 1) { // something to decide: a wavefront will (a single primitive is a leaf as it stands)
                    if (tid == 0) {
                        const uint32_t e = S.list_n++;
                        S.list[3 * e + 0] = first; S.list[3 * e + 1] = count; S.list[3 * e + 2] = nid;
                        for (int a = 0; a < 3; a++) { S.list_box[6 * e + a] = nlo[a]; S.list_box[6 * e + 3 + a] = nhi[a]; }
                    }
                    __syncthreads();
                }
                break;
            }
            if (count <= 1) break; // (list full and a single primitive: a leaf)
            // ---- centroid bounds and bins of [first, first + count)
            if (tid < 6) S.cb[tid] = tid < 3 ? 0xffffffffu : 0u;
            if (tid < 3 * kBins) {
                Bin e; e.count = 0;
                for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
                S.bins[tid] = e;
            }
            __syncthreads();
            const bool mine = tid < count;
            const uint32_t q = mine ? S.perm[0][first + tid] : 0u;
            float cen[3] = {0.0f, 0.0f, 0.0f};
            for (int a = 0; a < 3; a++) {
                cen[a] = mine ? 0.5f * (S.lo[a][q] + S.hi[a][q]) : 0.0f;
                const float l = wave_min(mine ? cen[a] : INFINITY), h = wave_max(mine ? cen[a] : -INFINITY);
                if (lane == 0 && l <= h) { atomicMin(&S.cb[a], f_order(l)); atomicMax(&S.cb[3 + a], f_order(h)); }
            }
            __syncthreads();
            int my_bin[3] = {0, 0, 0};
            {
                float blo[3] = {0.0f, 0.0f, 0.0f}, bhi[3] = {0.0f, 0.0f, 0.0f};
                if (mine)
                    for (int a = 0; a < 3; a++) { my_bin[a] = bin_of(cen[a], f_unorder(S.cb[a]), f_unorder(S.cb[3 + a])); blo[a] = S.lo[a][q]; bhi[a] = S.hi[a][q]; }
                wave_add_to_bins(S.bins, mine, my_bin, blo, bhi);
            }
            __syncthreads();
            if (wave == 0) {
                SplitChoice s;
                const int win = sweep48(S.bins, lane, s);
                if ((int)lane == (win >= 0 ? win : 0)) {
                    if (win < 0) s.axis = -1;
                    S.choice = s;
                }
            }
            __syncthreads();
            SplitChoice s = S.choice;
            bool split = false;
            {
                const float area = half_area(nlo, nhi);
                const float leaf_cost = (float)count * area;
                if (s.axis >= 0 && (s.cost + trav_cost * area < leaf_cost || (int)count > max_leaf)) split = true;
                else if ((int)count > max_leaf) { // coincident centroids: arbitrary halves keep leaves bounded
                    split = true;
                    s.axis = -1;
                    s.left_count = count / 2;
                    for (int a = 0; a < 3; a++) { s.llo[a] = s.rlo[a] = nlo[a]; s.lhi[a] = s.rhi[a] = nhi[a]; }
                }
            }
            if (!split) break; // a leaf (only when the list was full and the range small)
            const uint32_t lc = s.left_count, rc = count - lc, li = S.next_id;
            // partition: left ones first, in thread order on both sides (ballots per wavefront, prefix over the wavefronts)
            const bool left = mine && (s.axis < 0 ? tid < lc : my_bin[s.axis] <= s.plane);
            const unsigned long long lm = __ballot(left), rm = __ballot(mine && !left);
            if (lane == 0) { S.wave_cnt[wave][0] = (uint32_t)__popcll(lm); S.wave_cnt[wave][1] = (uint32_t)__popcll(rm); }
            __syncthreads();
            if (mine) {
                uint32_t off = 0;
                for (uint32_t w = 0; w < wave; w++) off += S.wave_cnt[w][left ? 0 : 1];
                const unsigned long long m = left ? lm : rm;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                S.perm[1][first + (left ? off + rank : lc + off + rank)] = (uint16_t)q;
            }
            if (tid == 0) {
                S.next_id = li + 2;
                SNode l, r;
                init_child(l, gfirst + first, lc, s.llo, s.lhi, nid);
                init_child(r, gfirst + first + lc, rc, s.rlo, s.rhi, nid);
                nodes[li] = l;
                nodes[li + 1] = r;
                nodes[nid].left = li;
                // the larger child waits on the stack
                const bool left_next = lc <= rc;
                S.stack[3 * sp + 0] = left_next ? first + lc : first;
                S.stack[3 * sp + 1] = left_next ? rc : lc;
                S.stack[3 * sp + 2] = left_next ? li + 1 : li;
                for (int a = 0; a < 3; a++) { S.stack_box[6 * sp + a] = left_next ? s.rlo[a] : s.llo[a]; S.stack_box[6 * sp + 3 + a] = left_next ? s.rhi[a] : s.lhi[a]; }
            }
            __syncthreads();
            if (mine) S.perm[0][first + tid] = S.perm[1][first + tid];
            sp++;
            const bool left_next = lc <= rc;
            if (left_next) { count = lc; nid = li; for (int a = 0; a < 3; a++) { nlo[a] = s.llo[a]; nhi[a] = s.lhi[a]; } }
            else { first = first + lc; count = rc; nid = li + 1; for (int a = 0; a < 3; a++) { nlo[a] = s.rlo[a]; nhi[a] = s.rhi[a]; } }
            __syncthreads();
        }
    }
    __syncthreads();
    // ---- stage B: every wavefront takes ranges off the list
    const uint32_t n_list = S.list_n;
    for (;;) {
        uint32_t e = 0;
        if (lane == 0) e = atomicAdd(&S.list_head, 1u);
        e = (uint32_t)__builtin_amdgcn_readfirstlane((int)e);
        if (e >= n_list) break;
        float nlo[3], nhi[3];
        for (int a = 0; a < 3; a++) { nlo[a] = S.list_box[6 * e + a]; nhi[a] = S.list_box[6 * e + 3 + a]; }
        finish_range_wave(S, wave, lane, S.list[3 * e + 0], S.list[3 * e + 1], S.list[3 * e + 2], nlo, nhi, gfirst, nodes, max_leaf, trav_cost);
    }
    __syncthreads();
    if (tid < gcount) order_out[gfirst + tid] = order_in[gfirst + S.perm[0][tid]];
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

// ---------------------------------------------------------------- a FOREST: many meshes in one build
// A scene of many small meshes is all launch latency when every mesh is built by itself (~30 dependent launches and a read-back each).  The
// level kernels above never cared how many nodes a level has, so the meshes of a scene are built TOGETHER: their primitives lie one mesh
// after the other in one position space, every mesh is a root (node ids 0 .. M - 1), and the levels, the workgroup phase and the emission
// run once for all of them.  Per tree: its own node region in the output (root = index 0), leaf ranges relative to the tree's first primitive.
__global__ void k_forest_init(Counters* ctr, uint32_t n_trees)
{
    if (threadIdx.x == 0) { ctr->node_count = n_trees; ctr->n_active[0] = 0; ctr->n_active[1] = 0; ctr->n_small = 0; }
}
// Bounds and centroid bounds of every tree, identity order, tree of every position: ALL positions by a fixed grid of workgroups, each over a
// contiguous chunk (round 3 gave a tree to ONE workgroup: a 720 k-primitive tree among 64 small ones — or two big meshes — kept a single CU busy
// for ~0.9 ms while 255 idled; that, not the level kernels, was why two large meshes built faster one after the other).  A thread accumulates
// for the tree of its current position and flushes (12 atomics on the tree's words) when its next position is another tree's; at the end the
// lanes of a wavefront that all hold the same tree are reduced in registers first.
constexpr uint32_t kForestBoundsBlocks = 1024;
__device__ inline uint32_t tree_of_position(const ForestTree* __restrict__ trees, uint32_t n_trees, uint32_t i)
{
    uint32_t lo = 0, hi = n_trees; // the last tree whose first position is <= i (trees lie one after the other; empty trees share a first)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (trees[mid].first <= i) lo = mid; else hi = mid;
    }
    while (lo > 0 && trees[lo].count == 0) lo--; // (an empty tree at the same offset as its successor owns nothing)
    return lo;
}
__global__ __launch_bounds__(kBlock) void k_forest_bounds(const DevBox* __restrict__ boxes, const ForestTree* __restrict__ trees, uint32_t n_trees, uint32_t n, uint32_t chunk,
                                                         uint32_t* order, uint32_t* node_of_pos, uint32_t* bounds)
{
    const uint32_t begin = blockIdx.x * chunk, end = min(n, begin + chunk);
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    uint32_t m = kNone, m_end = 0;
    auto flush = [&]() {
        uint32_t* b = bounds + (size_t)m * 12u;
        for (int a = 0; a < 3; a++) {
            atomicMin(&b[a], f_order(lo[a])); atomicMax(&b[3 + a], f_order(hi[a]));
            atomicMin(&b[6 + a], f_order(clo[a])); atomicMax(&b[9 + a], f_order(chi[a]));
            lo[a] = INFINITY; hi[a] = -INFINITY; clo[a] = INFINITY; chi[a] = -INFINITY;
        }
    };
    for (uint32_t i = begin + threadIdx.x; i < end; i += kBlock) {
        if (m == kNone || i >= m_end) {
            if (m != kNone) flush();
            m = tree_of_position(trees, n_trees, i);
            m_end = trees[m].first + trees[m].count;
        }
        order[i] = i;
        node_of_pos[i] = m;
        const DevBox bx = boxes[i];
        for (int a = 0; a < 3; a++) {
            const float l = bx.lo[a], h = bx.hi[a], c = 0.5f * (l + h);
            lo[a] = fminf(lo[a], l); hi[a] = fmaxf(hi[a], h);
            clo[a] = fminf(clo[a], c); chi[a] = fmaxf(chi[a], c);
        }
    }
    // what is left: one set of atomics per wavefront when its lanes agree on the tree (a large tree: every wavefront of the chunk), else per lane
    const unsigned long long have = __ballot(m != kNone);
    if (have == 0ull) return;
    const uint32_t lead_m = (uint32_t)__builtin_amdgcn_readlane((int)m, __ffsll((long long)have) - 1);
    if (__ballot(m != kNone && m != lead_m) == 0ull) {
        for (int a = 0; a < 3; a++) { lo[a] = wave_min(lo[a]); hi[a] = wave_max(hi[a]); clo[a] = wave_min(clo[a]); chi[a] = wave_max(chi[a]); }
        m = lead_m;
        if ((threadIdx.x & 63u) == 0u) flush();
    } else if (m != kNone) flush();
}
// one wavefront per tree: the root node from the tree's bounds, where the root goes next, and its bins on level 0
__global__ __launch_bounds__(64) void k_forest_roots(const ForestTree* __restrict__ trees, const uint32_t* __restrict__ bounds, Counters* ctr, SNode* nodes, uint32_t* active,
                                                    uint32_t* small, uint32_t* bin_slot, uint8_t* stamp, Bin* bins, uint32_t replicas, uint32_t small_cap)
{
    __shared__ uint32_t s_slot;
    const uint32_t m = blockIdx.x;
    const ForestTree t = trees[m];
    if (threadIdx.x == 0) {
        s_slot = kNone;
        const uint32_t* b = bounds + (size_t)m * 12u;
        SNode r;
        for (int a = 0; a < 3; a++) {
            r.lo[a] = f_unorder(b[a]); r.hi[a] = f_unorder(b[3 + a]);
            r.cb[a] = b[6 + a]; r.cb[3 + a] = b[9 + a];
        }
        r.first = t.first; r.count = t.count; r.left = kNone; r.parent = kNone;
        nodes[m] = r;
        if (t.count > small_cap) {
            const uint32_t slot = atomicAdd(&ctr->n_active[0], 1u);
            active[slot] = m; bin_slot[m] = slot; stamp[m] = 0;
            s_slot = slot;
        } else if (t.count) {
            small[atomicAdd(&ctr->n_small, 1u)] = m;
        }
    }
    __syncthreads();
    if (s_slot != kNone) { // the root's bins on level 0 (every replica)
        Bin e; e.count = 0;
        for (int a = 0; a < 3; a++) { e.lo[a] = 0xffffffffu; e.hi[a] = 0u; }
        for (uint32_t k = threadIdx.x; k < replicas * 3 * kBins; k += 64u) bins[(size_t)s_slot * replicas * 3 * kBins + k] = e;
    }
}
__global__ void k_forest_bounds_init(uint32_t* bounds, uint32_t n_words)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n_words) bounds[i] = (i % 6u) < 3u ? 0xffffffffu : 0u;
}
// even-depth interior nodes become wide nodes; every node learns its tree (= the id of its root)
__global__ void k_flag_forest(uint32_t n_nodes_cap, const Counters* __restrict__ ctr, const SNode* __restrict__ nodes, uint32_t* flag4, uint32_t* tree_of)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes_cap) return;
    uint32_t f = 0, root = kNone;
    if (i < ctr->node_count) {
        uint32_t depth = 0, r = i, p = nodes[i].parent;
        while (p != kNone) { depth++; r = p; p = nodes[p].parent; }
        root = r;
        if (nodes[i].left != kNone) f = (depth & 1u) ? 0u : 1u;
    }
    flag4[i] = f;
    tree_of[i] = root;
}
// a wide node's index inside its tree's region: the root is 0, the others take the next free one (any order will do: parents look them up)
__global__ void k_index_forest(uint32_t n_nodes_cap, const Counters* __restrict__ ctr, const SNode* __restrict__ nodes, const uint32_t* __restrict__ flag4,
                               const uint32_t* __restrict__ tree_of, uint32_t* idx4, uint32_t* tree_nodes)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    const bool wide = i < n_nodes_cap && i < ctr->node_count && flag4[i] != 0u;
    const bool is_root = wide && nodes[i].parent == kNone;
    if (is_root) idx4[i] = 0u;
    // one atomic per wavefront and tree (a scene of two big meshes would otherwise send every wide node to one of two words)
    const uint32_t t = wide ? tree_of[i] : kNone;
    bool todo = wide && !is_root;
    while (__ballot(todo) != 0ull) {
        const unsigned long long pending = __ballot(todo);
        const uint32_t lead_tree = (uint32_t)__shfl((int)t, __ffsll((long long)pending) - 1);
        const unsigned long long mine = __ballot(todo && t == lead_tree);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mine >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mine, 0u));
        uint32_t base = 0;
        if (todo && t == lead_tree && rank == 0u) base = atomicAdd(&tree_nodes[lead_tree], (uint32_t)__popcll(mine));
        base = (uint32_t)__shfl((int)base, __ffsll((long long)mine) - 1);
        if (todo && t == lead_tree) { idx4[i] = 1u + base + rank; todo = false; }
    }
}
__global__ void k_emit_forest(uint32_t n_nodes_cap, uint32_t n_trees, const Counters* __restrict__ ctr, const SNode* __restrict__ nodes, const uint32_t* __restrict__ flag4,
                              const uint32_t* __restrict__ idx4, const uint32_t* __restrict__ tree_of, const ForestTree* __restrict__ trees,
                              const uint32_t* __restrict__ tree_nodes, Node4* out_base, uint32_t* node_counts_out)
{
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_nodes_cap) return;
    const uint32_t total = ctr->node_count;
    if (i < n_trees) { // a root: its tree's node count; a tree that is one leaf (or empty) is one wide node with one (or no) child
        if (nodes[i].left == kNone) {
            Node4 o;
            for (int k = 0; k < 4; k++) {
                o.lox[k] = o.loy[k] = o.loz[k] = INFINITY; o.hix[k] = o.hiy[k] = o.hiz[k] = -INFINITY;
                o.child[k] = kInvalidRef; o.pad[k] = 0;
            }
            if (nodes[i].count) {
                o.lox[0] = nodes[i].lo[0]; o.loy[0] = nodes[i].lo[1]; o.loz[0] = nodes[i].lo[2];
                o.hix[0] = nodes[i].hi[0]; o.hiy[0] = nodes[i].hi[1]; o.hiz[0] = nodes[i].hi[2];
                o.child[0] = make_leaf(0u, nodes[i].count);
            }
            out_base[trees[i].node_base] = o;
            if (node_counts_out) node_counts_out[i] = 1;
            return;
        }
        if (node_counts_out) node_counts_out[i] = tree_nodes[i] + 1u;
    }
    if (i >= total || !flag4[i]) return;
    const ForestTree t = trees[tree_of[i]];
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
            o.child[k] = c.left == kNone ? make_leaf(c.first - t.first, c.count) : idx4[kids[k]];
        } else {
            o.lox[k] = o.loy[k] = o.loz[k] = INFINITY; o.hix[k] = o.hiy[k] = o.hiz[k] = -INFINITY;
            o.child[k] = kInvalidRef;
        }
        o.pad[k] = 0;
    }
    out_base[t.node_base + idx4[i]] = o;
}
// the primitive order holds positions of the whole forest: make every tree's part relative to the tree's first primitive
__global__ void k_forest_relative_order(uint32_t* order, uint32_t n, const ForestTree* __restrict__ trees, uint32_t n_trees)
{
    const uint32_t p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    uint32_t lo = 0, hi = n_trees; // the last tree with first <= p (trees are sorted by first, ranges are disjoint and cover [0, n))
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (trees[mid].first <= p) lo = mid; else hi = mid;
    }
    order[p] -= trees[lo].first;
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
    size_t ctr, nodes, order[2], nop[2], pbox[2], active[2], small, bin_slot, stamp, splits, fill, bins[2], flag4, idx4, cub, total, cub_bytes;
    uint32_t node_cap, big_cap;
};
Layout make_layout(uint32_t n)
{
    Layout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t m = n > 0 ? n : 1;
    L.node_cap = (uint32_t)(4 * m + 64); // phase 1 makes < 2 nodes per queued range, phase 2 reserves 2c ids for a range of c primitives
    L.big_cap = (uint32_t)(2 * m / small_cap_for(n) + 2); // ranges above the hand-over size that can coexist on one level
    L.ctr = take(sizeof(Counters));
    L.nodes = take((size_t)L.node_cap * sizeof(SNode));
    for (int k = 0; k < 2; k++) { L.order[k] = take(m * 4); L.nop[k] = take(m * 4); L.pbox[k] = take(m * sizeof(DevBox)); L.active[k] = take((size_t)L.big_cap * 4); }
    L.small = take((m + 1) * 4);
    L.bin_slot = take((size_t)L.node_cap * 4);
    L.stamp = take((size_t)L.node_cap);
    L.splits = take((size_t)L.node_cap * sizeof(Split));
    L.fill = take((size_t)L.node_cap * 4);
    for (int k = 0; k < 2; k++) L.bins[k] = take(((size_t)L.big_cap + 512) * 3 * kBins * sizeof(Bin)); // (nodes of a level) x (replicas of that level) <= max(big_cap, 256 + ...); one array per level parity
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
    uint8_t* stamp = (uint8_t*)(w + L.stamp);
    DevBox* pbox[2] = {(DevBox*)(w + L.pbox[0]), (DevBox*)(w + L.pbox[1])};
    Split* splits = (Split*)(w + L.splits);
    uint32_t* fill = (uint32_t*)(w + L.fill);
    Bin* bins[2] = {(Bin*)(w + L.bins[0]), (Bin*)(w + L.bins[1])};
    uint32_t* flag4 = (uint32_t*)(w + L.flag4);
    uint32_t* idx4 = (uint32_t*)(w + L.idx4);

    {   // ids a wavefront of phase 2 reserves but does not use must read as leaves nobody references: left = kNone
        hipError_t me = hipMemsetAsync(nodes, 0xff, (size_t)L.node_cap * sizeof(SNode), s);
        if (me == hipSuccess) me = hipMemsetAsync(stamp, 0xff, (size_t)L.node_cap, s);
        if (me != hipSuccess) return me;
    }
    hipLaunchKernelGGL(k_root_init, dim3(1), dim3(64), 0, s, ctr);
    if (n) hipLaunchKernelGGL(k_root_bounds, dim3(std::min(blocks(n), 1024u)), dim3(kBlock), 0, s, boxes, n, ctr, order[0], nop[0]);
    // upper bound of the big nodes of a level, and the replicas of their bins on that level (so that ~8 workgroups of k_bin share a copy)
    const uint32_t small_cap = small_cap_for(n);
    auto level_ub = [&](int l) { return (uint32_t)std::min<uint64_t>(l < 31 ? (1ull << l) : (1ull << 31), (uint64_t)n / small_cap + 1); };
    const uint32_t bin_groups = blocks(n); // workgroups of k_bin
    auto level_replicas = [&](int l) { return std::max(1u, std::min(32u, bin_groups / (8u * level_ub(l)))); };
    hipLaunchKernelGGL(k_root_node, dim3(1), dim3(64), 0, s, n, ctr, nodes, active[0], small, bin_slot, stamp, bins[0], level_replicas(0), small_cap);
    const bool dbg = env_switches().build_trace;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto msf = [](auto a, auto b) { return std::chrono::duration<float, std::milli>(b - a).count(); };
    if (dbg) (void)hipStreamSynchronize(s);
    const auto t_start = now();
    // phase 1: level by level.  The host does not know how many big nodes a level has: it launches for an upper bound (level l has at most
    // 2^l nodes, and at most n / kSmall of them are big) and for a depth estimated from n; the kernels of a level without big nodes return at
    // once.  One read-back after the estimated depth tells whether big nodes are left (a very uneven tree) and how many ranges phase 2 has.
    int cur = 0, level = 0;
    auto run_level = [&]() {
        const uint32_t par = (uint32_t)(level & 1);
        const uint32_t ub = level_ub(level), reps = level_replicas(level), reps_next = level_replicas(level + 1);
        const DevBox* pin = level == 0 ? boxes : pbox[cur]; // level 0: the order is the identity, the input boxes ARE position-ordered
        hipLaunchKernelGGL(k_bin, dim3(bin_groups), dim3(kBlock), 0, s, pin, nop[cur], nodes, bin_slot, stamp, bins[par], n, ctr, (uint32_t)level, reps);
        hipLaunchKernelGGL(k_split, dim3(blocks(ub, kBlock / 64)), dim3(kBlock), 0, s, active[par], active[par ^ 1], (uint32_t)level, ctr, nodes, bins[par], bins[par ^ 1], bin_slot, stamp, splits, fill,
                           small, reps, reps_next, small_cap);
        if (ub * 64u <= blocks(n)) // few nodes, each across many blocks: chunks of blocks take their slots together
            hipLaunchKernelGGL(k_partition_chunk, dim3(blocks(n, kPartChunk * kBlock)), dim3(kBlock), 0, s, pin, pbox[cur ^ 1], order[cur], nop[cur], order[cur ^ 1], nop[cur ^ 1],
                               nodes, splits, stamp, fill, n, (uint32_t)level);
        else
            hipLaunchKernelGGL(k_partition, dim3(blocks(n)), dim3(kBlock), 0, s, pin, pbox[cur ^ 1], order[cur], nop[cur], order[cur ^ 1], nop[cur ^ 1], nodes, splits, stamp, fill, n,
                               (uint32_t)level);
        cur ^= 1;
        level++;
    };
    uint32_t counts[4] = {0, 0, 0, 0}; // node_count, n_active[0], n_active[1], n_small
    auto read_counts = [&]() -> hipError_t {
        hipError_t e = hipMemcpyAsync(counts, ctr, sizeof(counts), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        return e;
    };
    if (n > small_cap) {
        int expect = n > 65536u ? 8 : 2; // log2(n / kSmall) levels for an even tree, some more for an uneven one (a small mesh is launch-bound: fewer spare levels, look sooner)
        for (uint32_t v = n / small_cap; v > 1; v >>= 1) expect++;
        while (level < expect && level < kMaxLevels) run_level();
    }
    hipError_t e = read_counts();
    if (e != hipSuccess) return e;
    while (counts[1 + (level & 1)] > 0 && level < kMaxLevels) { // (rare) big nodes are left: a few levels more, and look again
        if (counts[1 + (level & 1)] > L.big_cap) return hipErrorInvalidValue;
        for (int k = 0; k < 4 && level < kMaxLevels; k++) run_level();
        if ((e = read_counts()) != hipSuccess) return e;
    }
    const auto t_p1 = now();
    if (counts[1 + (level & 1)] > 0) return hipErrorInvalidValue; // deeper than kMaxLevels above kSmall: not a tree this builder makes
    // phase 2
    const uint32_t n_small = counts[3];
    if (n_small && small_cap == kSmall) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_small<kSmall>), dim3(n_small), dim3(kSmall), 0, s, boxes, order[cur], order_out, small, ctr, nodes, max_leaf, trav_cost);
    else if (n_small) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_small<kSmallFew>), dim3(n_small), dim3(kSmallFew), 0, s, boxes, order[cur], order_out, small, ctr, nodes, max_leaf, trav_cost);
    if (dbg) { (void)hipStreamSynchronize(s); fprintf(stderr, "sah n=%u: phase1 %.3f ms (%d levels), phase2 %.3f ms (%u small ranges)\n", n, msf(t_start, t_p1), level, msf(t_p1, now()), n_small); }
    // BVH2 -> Node4: internal nodes at even depth, children = grandchildren
    hipLaunchKernelGGL(k_flag_nodes, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, ctr, nodes, flag4);
    size_t cub_bytes = L.cub_bytes;
    e = hipcub::DeviceScan::ExclusiveSum(w + L.cub, cub_bytes, flag4, idx4, (int)L.node_cap, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_emit_nodes, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, ctr, nodes, flag4, idx4, nodes_out, node_count_out);
    return hipGetLastError();
}

size_t sah_forest_workspace_bytes(uint32_t n, uint32_t n_trees)
{
    return make_layout(n + n_trees).total + align_up((size_t)(4 * ((size_t)n + n_trees) + 64 + n_trees) * 4, 256) + align_up((size_t)n_trees * 4, 256) +
           align_up((size_t)n_trees * 12 * 4, 256); // (+ tree_of, tree_nodes, and 12 words of bounds per tree)
}

hipError_t sah_build_forest(hipStream_t s, const DevBox* boxes, uint32_t n, const ForestTree* trees, uint32_t n_trees, uint32_t largest_tree, void* workspace,
                            size_t workspace_bytes, Node4* nodes_base, uint32_t* order_out, uint32_t* node_counts_out, int max_leaf, float trav_cost)
{
    if (n == 0 || n_trees == 0) return hipErrorInvalidValue;
    // the single-tree layout for n + M primitives has room for the M extra roots everywhere a per-node array is sized from n
    const Layout L = make_layout(n + n_trees);
    const size_t tree_of_off = L.total, tree_nodes_off = tree_of_off + align_up((size_t)L.node_cap * 4, 256);
    if (sah_forest_workspace_bytes(n, n_trees) > workspace_bytes) return hipErrorInvalidValue;
    max_leaf = max_leaf < 1 ? 1 : (max_leaf > kMaxLeafTris ? kMaxLeafTris : max_leaf);
    char* w = static_cast<char*>(workspace);
    Counters* ctr = (Counters*)(w + L.ctr);
    SNode* nodes = (SNode*)(w + L.nodes);
    uint32_t* order[2] = {(uint32_t*)(w + L.order[0]), (uint32_t*)(w + L.order[1])};
    uint32_t* nop[2] = {(uint32_t*)(w + L.nop[0]), (uint32_t*)(w + L.nop[1])};
    uint32_t* active[2] = {(uint32_t*)(w + L.active[0]), (uint32_t*)(w + L.active[1])};
    uint32_t* small = (uint32_t*)(w + L.small);
    uint32_t* bin_slot = (uint32_t*)(w + L.bin_slot);
    uint8_t* stamp = (uint8_t*)(w + L.stamp);
    DevBox* pbox[2] = {(DevBox*)(w + L.pbox[0]), (DevBox*)(w + L.pbox[1])};
    Split* splits = (Split*)(w + L.splits);
    uint32_t* fill = (uint32_t*)(w + L.fill);
    Bin* bins[2] = {(Bin*)(w + L.bins[0]), (Bin*)(w + L.bins[1])};
    uint32_t* flag4 = (uint32_t*)(w + L.flag4);
    uint32_t* idx4 = (uint32_t*)(w + L.idx4);
    uint32_t* tree_of = (uint32_t*)(w + tree_of_off);
    uint32_t* tree_nodes = (uint32_t*)(w + tree_nodes_off);
    uint32_t* tree_bounds = (uint32_t*)(w + tree_nodes_off + align_up((size_t)n_trees * 4, 256));
    hipError_t e = hipMemsetAsync(nodes, 0xff, (size_t)L.node_cap * sizeof(SNode), s);
    if (e == hipSuccess) e = hipMemsetAsync(stamp, 0xff, (size_t)L.node_cap, s);
    if (e == hipSuccess) e = hipMemsetAsync(tree_nodes, 0, (size_t)n_trees * 4, s);
    if (e != hipSuccess) return e;
    const uint32_t small_cap = small_cap_for(n);
    // big nodes of level l: at most 2^l per tree, and at most n / small_cap altogether
    auto level_ub = [&](int l) { return (uint32_t)std::min<uint64_t>(l < 20 ? ((uint64_t)n_trees << l) : ~0ull >> 1, (uint64_t)n / small_cap + 1); };
    const uint32_t bin_groups = blocks(n);
    auto level_replicas = [&](int l) { return std::max(1u, std::min(32u, bin_groups / (8u * level_ub(l)))); };
    hipLaunchKernelGGL(k_forest_init, dim3(1), dim3(64), 0, s, ctr, n_trees);
    hipLaunchKernelGGL(k_forest_bounds_init, dim3(blocks(n_trees * 12u)), dim3(kBlock), 0, s, tree_bounds, n_trees * 12u);
    {
        const uint32_t chunk = (uint32_t)align_up(((size_t)n + kForestBoundsBlocks - 1) / kForestBoundsBlocks, kBlock);
        hipLaunchKernelGGL(k_forest_bounds, dim3(blocks(n, chunk)), dim3(kBlock), 0, s, boxes, trees, n_trees, n, chunk, order[0], nop[0], tree_bounds);
    }
    hipLaunchKernelGGL(k_forest_roots, dim3(n_trees), dim3(64), 0, s, trees, tree_bounds, ctr, nodes, active[0], small, bin_slot, stamp, bins[0], level_replicas(0), small_cap);
    int cur = 0, level = 0;
    auto run_level = [&]() {
        const uint32_t par = (uint32_t)(level & 1);
        const uint32_t ub = level_ub(level), reps = level_replicas(level), reps_next = level_replicas(level + 1);
        const DevBox* pin = level == 0 ? boxes : pbox[cur];
        hipLaunchKernelGGL(k_bin, dim3(bin_groups), dim3(kBlock), 0, s, pin, nop[cur], nodes, bin_slot, stamp, bins[par], n, ctr, (uint32_t)level, reps);
        hipLaunchKernelGGL(k_split, dim3(blocks(ub, kBlock / 64)), dim3(kBlock), 0, s, active[par], active[par ^ 1], (uint32_t)level, ctr, nodes, bins[par], bins[par ^ 1], bin_slot, stamp, splits, fill,
                           small, reps, reps_next, small_cap);
        if ((largest_tree >> std::min(level, 31)) >= 64u * kBlock) // the largest tree's nodes still span many blocks: chunks of blocks take their slots together
            hipLaunchKernelGGL(k_partition_chunk, dim3(blocks(n, kPartChunk * kBlock)), dim3(kBlock), 0, s, pin, pbox[cur ^ 1], order[cur], nop[cur], order[cur ^ 1], nop[cur ^ 1],
                               nodes, splits, stamp, fill, n, (uint32_t)level);
        else
            hipLaunchKernelGGL(k_partition, dim3(blocks(n)), dim3(kBlock), 0, s, pin, pbox[cur ^ 1], order[cur], nop[cur], order[cur ^ 1], nop[cur ^ 1], nodes, splits, stamp, fill, n,
                               (uint32_t)level);
        cur ^= 1;
        level++;
    };
    uint32_t counts[4] = {0, 0, 0, 0};
    auto read_counts = [&]() -> hipError_t {
        hipError_t e2 = hipMemcpyAsync(counts, ctr, sizeof(counts), hipMemcpyDeviceToHost, s);
        if (e2 == hipSuccess) e2 = hipStreamSynchronize(s);
        return e2;
    };
    if (largest_tree > small_cap) {
        int expect = largest_tree > 65536u ? 8 : 3;
        for (uint32_t v = largest_tree / small_cap; v > 1; v >>= 1) expect++;
        while (level < expect && level < kMaxLevels) run_level();
    }
    if ((e = read_counts()) != hipSuccess) return e;
    while (counts[1 + (level & 1)] > 0 && level < kMaxLevels) {
        if (counts[1 + (level & 1)] > L.big_cap) return hipErrorInvalidValue;
        for (int k = 0; k < 4 && level < kMaxLevels; k++) run_level();
        if ((e = read_counts()) != hipSuccess) return e;
    }
    if (counts[1 + (level & 1)] > 0) return hipErrorInvalidValue;
    const uint32_t n_small = counts[3];
    if (n_small && small_cap == kSmall) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_small<kSmall>), dim3(n_small), dim3(kSmall), 0, s, boxes, order[cur], order_out, small, ctr, nodes, max_leaf, trav_cost);
    else if (n_small) hipLaunchKernelGGL(HIP_KERNEL_NAME(k_small<kSmallFew>), dim3(n_small), dim3(kSmallFew), 0, s, boxes, order[cur], order_out, small, ctr, nodes, max_leaf, trav_cost);
    hipLaunchKernelGGL(k_flag_forest, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, ctr, nodes, flag4, tree_of);
    hipLaunchKernelGGL(k_index_forest, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, ctr, nodes, flag4, tree_of, idx4, tree_nodes);
    hipLaunchKernelGGL(k_emit_forest, dim3(blocks(L.node_cap)), dim3(kBlock), 0, s, L.node_cap, n_trees, ctr, nodes, flag4, idx4, tree_of, trees, tree_nodes, nodes_base, node_counts_out);
    return hipGetLastError();
}
void launch_forest_relative_order(hipStream_t s, uint32_t* order, uint32_t n, const ForestTree* trees, uint32_t n_trees)
{
    if (n && n_trees) hipLaunchKernelGGL(k_forest_relative_order, dim3(blocks(n)), dim3(kBlock), 0, s, order, n, trees, n_trees);
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
