// traverse.h — two-level (TLAS of instances -> per-mesh BLAS) BVH4 traversal, one ray per lane.
//
// Replaces intersect_top_mbvh/intersect_mbvh of backends/gpu-rt/shaders/ray_gen.comp:202-250,310-362
// (closest hit) and ray_shadow.comp:83-132,191-243 (any hit).  Same results, different machine:
//   * one loop, one stack: TLAS and BLAS entries share a short per-lane stack held in LDS
//     (lane-interleaved, conflict-free), spilling to HBM only past kStackLds entries — the reference
//     keeps two 32-entry private arrays per thread in scratch memory (ray_gen.comp:204,312);
//   * leaves hold 48-B triangle packets in leaf order — no prim-index indirection and no 176-B
//     RTTriangle gather in the leaf loop (ray_gen.comp:230-233);
//   * nodes are 64 B (child boxes quantised to 8 bits per plane, conservatively), 4 loads per visit instead of 7;
//   * node boxes are padded at build time, so the slab test is conservative with respect to the
//     Moeller-Trumbore arithmetic and the answer is independent of the tree.
// The per-triangle arithmetic is intersection.glsl:1-38 / 40-70 operation for operation.
#pragma once
#include "device_math.h"
#include "device_types.h"

// What ships (every alternative measured in rounds 1-5 is a row of EXPERIMENTS.md and a patch under experiments/traversal_variants/):
//   * a lane reads the copy of the tree made for ITS ray's octant (device_types.h, make_octant_node): the planes it enters and leaves a box
//     through are in fixed words (no per-plane selects), the children are stored front to back along the octant's diagonal, so the hit ones
//     go on the stack in the stored order — no keys, no sorting network;
//   * the slab test is ONE compare per child, min(tf, t) >= max(tn, 0) (degenerate directions are turned away at the entry);
//   * the leaf's triangle test has no early outs (traverse_leaf.inc);
//   * an instance whose inverse matrix is exactly the identity is entered without the matrix product;
//   * the world-space ray is parked in LDS above the lane's stack column (6 words) and fetched back when an instance is entered or left;
//   * pops of the any-hit loop read LDS with ds_read_b32 (the closest-hit loop keeps the flat load the compiler merges the two stack homes into);
//   * the loop's SHAPE per kind of ray: any hit and both streaming kinds FLAT (traverse_flat.inc), the plain closest hit NESTED (traverse_nested.inc).
namespace rfwhip {

constexpr int kTraceBlock = 64;   // threads per workgroup of the trace kernels (one wavefront)
constexpr int kStackLds = 16;     // stack entries per lane kept in LDS
constexpr int kStackLdsAny = 12;   // the same for any hit (its ray is parked in LDS too: 18 rows = 4.5 KB per wavefront; measured: no spills at 8 waves per SIMD)
constexpr int kStackSpill = 48;   // further entries per lane in HBM (rarely touched)
// any hit: a lane that holds a leaf waits for the next even trip of the loop, which batches the triangle tests (round 1; periods 1 / 3 / 4 and
// waiting for 8 / 16 / 24 lanes at leaves were measured in round 5: EXPERIMENTS.md)
constexpr uint32_t kAnyLeafPeriod = 2;
constexpr int kTraceLdsRows = kStackLds + 6, kTraceLdsRowsAny = kStackLdsAny + 6; // LDS words per lane of a closest-hit / an any-hit trace kernel: stack + parked ray

struct SceneView {
    const Node4Q* tlas_nodes;
    const uint32_t* tlas_prims; // instance ids in TLAS leaf order
    const InstanceXform* instances;
    const Node4Q* blas_nodes;
    // the per-octant copies of both trees (make_octant_node: entry / exit planes picked, children front to back), copy `oct` of node i at
    // [oct * stride + i]: what the traversal below reads
    const Node4Q* tlas_oct;
    const Node4Q* blas_oct;
    uint32_t tlas_oct_stride, blas_oct_stride;
    const TriPacket* tri_packets;
    uint32_t* spill;            // kStackSpill x spill_stride
    uint32_t spill_stride;
    uint32_t spill_rows;        // rows of `spill` a lane may use (kStackSpill; a test option lowers it to exercise the overflow path)
    uint32_t* overflow_flag;    // pinned host word (mapped): set when a ray's stack would exceed LDS + spill rows
    QueueCounters* counters;
};

typedef float v2f __attribute__((ext_vector_type(2)));

// 1 / direction for the slab test only (boxes are padded: an ulp does not matter; the triangle test keeps IEEE division)
RFW_DI f3 slab_inv(const f3 d)
{
    return mk3(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
}

RFW_DI uint32_t octant_bits(const f3 inv) { return (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u); }
// node array of a space for a ray with reciprocal direction `inv`: the TLAS, or the BLAS region `node_base` of an instance
RFW_DI const Node4Q* space_nodes(const SceneView& sc, const bool blas, const uint32_t node_base, const f3 inv)
{
    const uint32_t oct = octant_bits(inv);
    return blas ? sc.blas_oct + (size_t)oct * sc.blas_oct_stride + node_base : sc.tlas_oct + (size_t)oct * sc.tlas_oct_stride;
}

struct TravCounters {
    uint32_t nodes, tris, insts;
    // COUNT mode: times this lane was the first active lane of a node test / triangle test; summed over a wavefront = how often the
    // wavefront executed that code (lane utilisation of the node test = nodes / (64 * wave_nodes))
    uint32_t wave_nodes = 0, wave_tris = 0;
    uint32_t wave_uniform = 0; // node-test executions whose active lanes all visit one node (a scalar fetch would serve them)
};

// Closest hit (ANY_HIT = false): on return t/hu/hv/hit_inst/hit_tri describe the nearest accepted hit, ties resolved
// to the lowest (instance, triangle) id.  Any hit (ANY_HIT = true): returns true as soon as one triangle has
// t_min < t' < t.
// FAR_FIRST (any hit only): hit children are visited in order of DECREASING EXIT distance (key = -exit instead of entry) — the search for an
// occluder starts at the far end of the ray.  The triangle tests, and therefore the answer, are the same in either order.
template <bool ANY_HIT, bool COUNT, bool FAR_FIRST = false>
RFW_DI bool traverse(const SceneView& sc, const f3 O, const f3 D, const float t_min, float& t, float& hu, float& hv, int32_t& hit_inst,
                     int32_t& hit_tri, uint32_t* lds_stack, const uint32_t lane_slot, const uint32_t spill_base, TravCounters& tc)
{
    constexpr int kStack = ANY_HIT ? kStackLdsAny : kStackLds; // LDS stack rows of this kernel flavour
    f3 o = O, d = D;
    // the world-space ray is parked in LDS (6 words above the lane's stack column) and fetched back when an instance is entered or left, so
    // it does not occupy six registers through the whole loop (the closest-hit kernels spill at 6 waves per SIMD, the any-hit ones at 8)
    {
        uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        park[0] = fbits(O.x); park[kTraceBlock] = fbits(O.y); park[2 * kTraceBlock] = fbits(O.z);
        park[3 * kTraceBlock] = fbits(D.x); park[4 * kTraceBlock] = fbits(D.y); park[5 * kTraceBlock] = fbits(D.z);
    }
    auto world_o = [&]() -> f3 {
        const uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        return mk3(bitsf(park[0]), bitsf(park[kTraceBlock]), bitsf(park[2 * kTraceBlock]));
    };
    auto world_d = [&]() -> f3 {
        const uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        return mk3(bitsf(park[3 * kTraceBlock]), bitsf(park[4 * kTraceBlock]), bitsf(park[5 * kTraceBlock]));
    };
    // the single compare of the slab test takes max(tn, 0), and min / max IGNORE NaNs: a direction of exact zeros (or with a NaN) would enter
    // every box.  Such a ray hits nothing
    if (!((D.x == D.x) && (D.y == D.y) && (D.z == D.z) && (D.x != 0.0f || D.y != 0.0f || D.z != 0.0f))) return false;
    f3 inv = slab_inv(d);
    int sp = 0;
    int blas_sp = -1;          // stack height at BLAS entry; -1 = currently in the TLAS
    int32_t cur_inst = -1;
    uint32_t tri_base = 0;
    uint32_t cur = 0;          // TLAS root (interior ref 0)
    const Node4Q* nodes = space_nodes(sc, false, 0u, inv); // node array of the current space (TLAS, or the entered instance's BLAS)

    // pops of the any-hit kernel read LDS with ds_read_b32 (measured +0.7 % frame rate); the closest-hit kernels keep the flat load the
    // compiler builds, because every other formulation of their pop tried (ds_read, top of stack in a register) changed their loop nest
    // for the worse (k_primary 0.335 -> 0.39 ms)
    constexpr bool kDsPop = ANY_HIT;
    // A lane's column of the HBM spill rows = spill_base (wave-uniform: lane 0's column) + its lane.  Formed where it is used, behind a
    // barrier the optimiser cannot look through: hoisted out of the loop the 64-bit address of the column is a register pair that lives
    // through the whole traversal — at 8 waves per SIMD it went to scratch memory — for a path almost no ray takes.
    auto spill_slot = [&]() -> uint32_t {
        uint32_t l = lane_slot;
        if (ANY_HIT) asm volatile("" : "+v"(l)); // (closest hit has the registers: hoisted it is 4 ... 12 % faster, measured)
        return spill_base + l;
    };
    auto push = [&](uint32_t v) {
        if (sp < kStack) lds_stack[sp * kTraceBlock + lane_slot] = v;
        else if (sp < kStack + (int)sc.spill_rows) sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot()] = v;
        else {
            // overflow (a tree deeper than LDS + spill rows): the entry is dropped and sp does NOT advance, so pop() never indexes
            // past the spill rows — the ray finishes deterministically on what it has (possibly missing a hit), and the host reports
            // RFW_HIP_E_STATE from the next render / read / query (the flag lives in pinned host memory: no read-back needed)
            *sc.overflow_flag = 1u;
            return;
        }
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        sp--;
        if (kDsPop) {
            // The LDS read is unconditional (row clamped) and the HBM spill row overrides it in a branch that is almost never taken.  Written
            // as `sp < kStack ? lds[...] : spill[...]` the compiler selects between the two ADDRESSES and issues one flat_load_dword — a
            // vector-memory instruction through the texture-address unit, for every pop, although the entry is in LDS
            uint32_t v = lds_stack[(sp < kStack ? sp : kStack - 1) * kTraceBlock + lane_slot];
            if (__builtin_expect(sp >= kStack, 0)) {
                v = sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot()];
                asm volatile("" : "+v"(v)); // the merged value is not a load: keeps the compiler from folding both loads into one flat load again
            }
            return v;
        }
        if (sp < kStack) return lds_stack[sp * kTraceBlock + lane_slot];
        return sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot()];
    };

    // (the loop is included once per kind of ray so that each has the shape that measured best for it)
    if constexpr (ANY_HIT) {
#define RFW_TRAV_TOP
#define RFW_TRAV_OCCLUDED occ_ = 1u; next_ = kDoneRef; break;
#define RFW_TRAV_FLAT_WHILE __ballot(cur != kDoneRef) != 0ull
#define RFW_TRAV_LEAF_TRIP ((iteration % kAnyLeafPeriod) == 0u)
#define RFW_TRAV_TLAS_TRIP true
#include "traverse_flat.inc"
#undef RFW_TRAV_TOP
#undef RFW_TRAV_OCCLUDED
#undef RFW_TRAV_FLAT_WHILE
#undef RFW_TRAV_LEAF_TRIP
#undef RFW_TRAV_TLAS_TRIP
        return occ_ != 0u;
    } else {
#define RFW_TRAV_OCCLUDED
#define RFW_TRAV_EXHAUSTED break;
#include "traverse_nested.inc"
#undef RFW_TRAV_OCCLUDED
#undef RFW_TRAV_EXHAUSTED
        return false;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Streaming flavour.  The trace kernels are bound by vector-instruction issue, and an instruction costs the same with 10 active lanes as
// with 64: with one ray per lane a wavefront lasts as long as its longest ray (measured: the lanes of a shadow-ray wavefront are busy
// 67 % of its life, those of a bounce's extension rays 51 %).  Here a wavefront owns a RUN of consecutive queue entries instead, and
// whenever `refill` of its lanes are idle they commit their results and take the run's next entries — the tail is paid once per run, not
// once per 64 rays, and the rays of a wavefront still come from one neighbourhood of the queue (no atomics, no global cursor: the
// assignment of rays to wavefronts is static, the assignment to LANES is not, and no result depends on either).
//   st.more()                 wave-uniform: the run has entries left
//   st.fetch(idle, O, D, t)   the calling lanes (all idle) take the next entries; false for a lane that gets none (Stream::kTMin: the queue's t_min)
//   st.commit(occluded, t, hu, hv, hit_inst, hit_tri)   the calling lane's finished ray
template <bool ANY_HIT, bool COUNT, bool FAR_FIRST, class Stream>
RFW_DI void traverse_stream(const SceneView& sc, Stream& st, const uint32_t refill, const uint32_t leaf_gate, uint32_t* lds_stack, const uint32_t lane_slot, const uint32_t spill_base,
                            TravCounters& tc)
{
    constexpr int kStack = ANY_HIT ? kStackLdsAny : kStackLds;
    f3 o = mk3(0.0f), d = o;
    const float t_min = Stream::kTMin; // (the same for every ray of a queue: not a register per lane)
    float t = 0.0f, hu = 0.0f, hv = 0.0f;
    int32_t hit_inst = -1, hit_tri = -1;
    auto world_o = [&]() -> f3 {
        const uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        return mk3(bitsf(park[0]), bitsf(park[kTraceBlock]), bitsf(park[2 * kTraceBlock]));
    };
    auto world_d = [&]() -> f3 {
        const uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        return mk3(bitsf(park[3 * kTraceBlock]), bitsf(park[4 * kTraceBlock]), bitsf(park[5 * kTraceBlock]));
    };
    f3 inv = mk3(0.0f);
    int sp = 0;
    int blas_sp = -1;
    int32_t cur_inst = -1;
    uint32_t tri_base = 0;
    uint32_t cur = 0xfffffffdu; // kIdleRef (traverse_flat.inc): the lane holds no ray yet
    const Node4Q* nodes = sc.tlas_nodes; // (set with every ray a lane takes)
    constexpr bool kDsPop = ANY_HIT;
    // A lane's column of the HBM spill rows = spill_base (wave-uniform: lane 0's column) + its lane.  Formed where it is used, behind a
    // barrier the optimiser cannot look through: hoisted out of the loop the 64-bit address of the column is a register pair that lives
    // through the whole traversal — at 8 waves per SIMD it went to scratch memory — for a path almost no ray takes.
    auto spill_slot = [&]() -> uint32_t {
        uint32_t l = lane_slot;
        if (ANY_HIT) asm volatile("" : "+v"(l)); // (closest hit has the registers: hoisted it is 4 ... 12 % faster, measured)
        return spill_base + l;
    };
    auto push = [&](uint32_t v) {
        if (sp < kStack) lds_stack[sp * kTraceBlock + lane_slot] = v;
        else if (sp < kStack + (int)sc.spill_rows) sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot()] = v;
        else {
            *sc.overflow_flag = 1u;
            return;
        }
        sp++;
    };
    auto pop = [&]() -> uint32_t {
        sp--;
        if (kDsPop) {
            uint32_t v = lds_stack[(sp < kStack ? sp : kStack - 1) * kTraceBlock + lane_slot];
            if (__builtin_expect(sp >= kStack, 0)) {
                v = sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot()];
                asm volatile("" : "+v"(v));
            }
            return v;
        }
        if (sp < kStack) return lds_stack[sp * kTraceBlock + lane_slot];
        return sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot()];
    };
    const uint64_t everyone = __ballot(1);
    // Leaves wait for company: the packet loop (and the entry into an instance) costs a wavefront the same with 3 active lanes as with 40, and
    // with refilled lanes at every depth some lane holds a leaf in almost every trip.  A lane at a leaf sits out until `leaf_gate` lanes
    // are, or until no lane has a node to test.
    bool do_leaves = true;
// The supply of rays at the top of every trip of the flat loop (traverse_flat.inc): a lane's state is `cur` alone — kIdleRef: no ray,
// kDoneRef: a finished ray waiting to be committed, anything else: at work — no flags that live across the loop's
// branches (each would be an SGPR pair merged with three scalar instructions at every join)
#define RFW_STREAM_TOP_FLAT                                                                                                           \
    if (iteration > (1u << 20)) { *sc.overflow_flag = 1u; break; }                                                                    \
    {                                                                                                                                 \
        const bool have_ = (cur + 3u) > 1u; /* neither kIdleRef nor kDoneRef */                                                       \
        const uint64_t idle = __ballot(!have_);                                                                                       \
        if (idle != 0ull) {                                                                                                           \
            const bool more = st.more();                                                                                              \
            if (!more && idle == everyone) break;                                                                                     \
            if (more && (uint32_t)__popcll(idle) >= refill) {                                                                         \
                if (!have_) {                                                                                                         \
                    if (cur == kDoneRef) st.commit(occ_ != 0u, t, hu, hv, hit_inst, hit_tri);                                         \
                    const bool got_ = st.fetch(idle, o, d, t);                                                                        \
                    cur = kIdleRef;                                                                                                   \
                    if (got_) {                                                                                                       \
                        {   /* the fetched ray lands in the traversal's own registers and is parked at once */                         \
                            uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;                                            \
                            park[0] = fbits(o.x); park[kTraceBlock] = fbits(o.y); park[2 * kTraceBlock] = fbits(o.z);                 \
                            park[3 * kTraceBlock] = fbits(d.x); park[4 * kTraceBlock] = fbits(d.y); park[5 * kTraceBlock] = fbits(d.z); \
                        }                                                                                                             \
                        inv = slab_inv(d);                                                                                            \
                        sp = 0; blas_sp = -1; cur_inst = -1; tri_base = 0; nodes = space_nodes(sc, false, 0u, inv);                   \
                        hu = 0.0f; hv = 0.0f; hit_inst = -1; hit_tri = -1; occ_ = 0u;                                                 \
                        /* a degenerate direction hits nothing (see traverse()): finished before it starts */                          \
                        const bool fine_ = ((d.x == d.x) && (d.y == d.y) && (d.z == d.z) && (d.x != 0.0f || d.y != 0.0f || d.z != 0.0f)); \
                        cur = fine_ ? 0u : kDoneRef;                                                                                  \
                    }                                                                                                                 \
                }                                                                                                                     \
                st.advance(idle);                                                                                                     \
            }                                                                                                                         \
        }                                                                                                                             \
    }                                                                                                                                 \
    {                                                                                                                                 \
        const bool have_ = (cur + 3u) > 1u;                                                                                           \
        const uint64_t busy = __ballot(have_), at_leaf = __ballot(have_ & ((int32_t)cur < -3));                                       \
        do_leaves = at_leaf == busy || (uint32_t)__popcll(at_leaf) >= leaf_gate;                                                      \
    }
#define RFW_TRAV_TOP RFW_STREAM_TOP_FLAT
#define RFW_TRAV_FLAT_WHILE true
#define RFW_TRAV_LEAF_TRIP do_leaves
#define RFW_TRAV_TLAS_TRIP do_leaves
    if constexpr (ANY_HIT) {
#define RFW_TRAV_OCCLUDED occ_ = 1u; next_ = kDoneRef; break;
#include "traverse_flat.inc"
#undef RFW_TRAV_OCCLUDED
        if (cur == kDoneRef) st.commit(occ_ != 0u, t, hu, hv, hit_inst, hit_tri);
    } else {
#define RFW_TRAV_OCCLUDED
#include "traverse_flat.inc"
#undef RFW_TRAV_OCCLUDED
        if (cur == kDoneRef) st.commit(false, t, hu, hv, hit_inst, hit_tri);
    }
#undef RFW_TRAV_TOP
#undef RFW_TRAV_FLAT_WHILE
#undef RFW_TRAV_LEAF_TRIP
#undef RFW_TRAV_TLAS_TRIP
#undef RFW_STREAM_TOP_FLAT
}

} // namespace rfwhip
