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

#ifndef RFW_ANY_PARK
#define RFW_ANY_PARK 1 // any hit parks its ray too, over a 12-row stack (measured: no spills at 8 waves per SIMD, +0.6 %)
#endif
#ifndef RFW_POP_DS_READ
#define RFW_POP_DS_READ 1 // bit 0: any hit, bit 1: closest hit — pops read LDS with ds_read_b32 instead of the flat load the compiler merges the two stack homes into
#endif

#ifndef RFW_SCALAR_NODES
#define RFW_SCALAR_NODES 0
#endif
#ifndef RFW_RAY_IN_LDS
#define RFW_RAY_IN_LDS 1 // closest hit parks the world-space ray in LDS (measured: no spills at 6 waves per SIMD, +0.9 %)
#endif
namespace rfwhip {

constexpr int kTraceBlock = 64;   // threads per workgroup of the trace kernels (one wavefront)
constexpr int kStackLds = 16;     // stack entries per lane kept in LDS
constexpr int kStackLdsAny = 12;  // the same for any hit when its ray is parked in LDS too (RFW_ANY_PARK): 18 rows = 4.5 KB per wavefront
constexpr int kStackSpill = 48;   // further entries per lane in HBM (rarely touched)

struct SceneView {
    const Node4Q* tlas_nodes;
    const uint32_t* tlas_prims; // instance ids in TLAS leaf order
    const InstanceXform* instances;
    const Node4Q* blas_nodes;
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
                     int32_t& hit_tri, uint32_t* lds_stack, const uint32_t lane_slot, const uint32_t spill_slot, TravCounters& tc)
{
    constexpr int kStack = (ANY_HIT && RFW_ANY_PARK) ? kStackLdsAny : kStackLds; // LDS stack rows of this kernel flavour
    constexpr bool kPark = RFW_RAY_IN_LDS && (!ANY_HIT || RFW_ANY_PARK);
    f3 o = O, d = D;
#if RFW_RAY_IN_LDS
    // closest hit: the world-space ray is parked in LDS (6 words above the lane's stack column) and fetched back when an instance is
    // entered or left, so it does not occupy six registers through the whole loop (the closest-hit kernels spill at 6 waves per SIMD)
    if (kPark) {
        uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        park[0] = fbits(O.x); park[kTraceBlock] = fbits(O.y); park[2 * kTraceBlock] = fbits(O.z);
        park[3 * kTraceBlock] = fbits(D.x); park[4 * kTraceBlock] = fbits(D.y); park[5 * kTraceBlock] = fbits(D.z);
    }
    auto world_o = [&]() -> f3 {
        if (!kPark) return O;
        const uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        return mk3(bitsf(park[0]), bitsf(park[kTraceBlock]), bitsf(park[2 * kTraceBlock]));
    };
    auto world_d = [&]() -> f3 {
        if (!kPark) return D;
        const uint32_t* park = lds_stack + kStack * kTraceBlock + lane_slot;
        return mk3(bitsf(park[3 * kTraceBlock]), bitsf(park[4 * kTraceBlock]), bitsf(park[5 * kTraceBlock]));
    };
#else
    auto world_o = [&]() -> f3 { return O; };
    auto world_d = [&]() -> f3 { return D; };
#endif
    f3 inv = slab_inv(d);
    int sp = 0;
    int blas_sp = -1;          // stack height at BLAS entry; -1 = currently in the TLAS
    int32_t cur_inst = -1;
    uint32_t tri_base = 0;
    uint32_t cur = 0;          // TLAS root (interior ref 0)
    const Node4Q* nodes = sc.tlas_nodes; // node array of the current space (TLAS, or the entered instance's BLAS)

    // pops of the any-hit kernel read LDS with ds_read_b32 (measured +0.7 % frame rate); the closest-hit kernels keep the flat load the
    // compiler builds, because every other formulation of their pop tried (ds_read, top of stack in a register) changed their loop nest
    // for the worse (k_primary 0.335 -> 0.39 ms)
    constexpr bool kDsPop = ((RFW_POP_DS_READ >> (ANY_HIT ? 0 : 1)) & 1) != 0;
    auto push = [&](uint32_t v) {
        if (sp < kStack) lds_stack[sp * kTraceBlock + lane_slot] = v;
        else if (sp < kStack + (int)sc.spill_rows) sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot] = v;
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
                v = sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot];
                asm volatile("" : "+v"(v)); // the merged value is not a load: keeps the compiler from folding both loads into one flat load again
            }
            return v;
        }
        if (sp < kStack) return lds_stack[sp * kTraceBlock + lane_slot];
        return sc.spill[(size_t)(sp - kStack) * sc.spill_stride + spill_slot];
    };

    uint32_t iteration = 0;
    for (;;) {
        iteration++;
        if (!(cur & kLeafBit)) {
            // ---- interior node: 4-wide slab test on the quantised child boxes (64 B = 4 dwordx4 per lane)
            const uint4* np = reinterpret_cast<const uint4*>(nodes + cur);
#if RFW_SCALAR_NODES
            // experiment: when every active lane of the wavefront visits the SAME node (68 % of the primary rays' node tests, 20 % of the
            // shadow rays'), fetch it once through the scalar cache instead of 4 vector loads through the texture-address unit
            uint4 w0, w1, w2, ch;
            {
                typedef uint32_t su4 __attribute__((ext_vector_type(4)));
                typedef const su4 __attribute__((address_space(4))) * scalar_ptr;
                const uintptr_t mine = (uintptr_t)np;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mine), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mine >> 32));
                const uintptr_t first = ((uintptr_t)hi << 32) | lo;
                if (__ballot(mine != first) == 0ull) {
                    scalar_ptr sp = (scalar_ptr)first;
                    const su4 a = sp[0], b = sp[1], c = sp[2], d = sp[3];
                    w0 = make_uint4(a.x, a.y, a.z, a.w); w1 = make_uint4(b.x, b.y, b.z, b.w); w2 = make_uint4(c.x, c.y, c.z, c.w); ch = make_uint4(d.x, d.y, d.z, d.w);
                } else {
                    w0 = np[0]; w1 = np[1]; w2 = np[2]; ch = np[3];
                }
            }
#else
            const uint4 w0 = np[0], w1 = np[1], w2 = np[2], ch = np[3];
#endif
            if (COUNT) {
                tc.nodes++;
                if (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u) {
                    tc.wave_nodes++;
                }
                {
                    const uint64_t first_ptr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)np >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)np);
                    const bool all_same = __ballot((uintptr_t)np == first_ptr) == __ballot(1);
                    if (all_same && __builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u) tc.wave_uniform++;
                }
            }
            // plane = origin + q * scale  =>  t = q * (scale * inv) + (origin - o) * inv : one cvt + one fma per plane
            const float Ax = bitsf(w0.w) * inv.x, Ay = bitsf(w2.z) * inv.y, Az = bitsf(w2.w) * inv.z;
            const float Bx = (bitsf(w0.x) - o.x) * inv.x, By = (bitsf(w0.y) - o.y) * inv.y, Bz = (bitsf(w0.z) - o.z) * inv.z;
            int32_t key[4];
            bool hit[4];
            uint32_t nhit = 0;
            // the ray's direction signs pick the near and the far plane of each axis, so a child costs 6 conversions, 3 packed
            // FMAs (near, far share scale and offset), one max3 and one min3.  A NaN (0 * inf on an axis-parallel ray) is ignored
            // by max3 / min3 and only drops that axis' constraint: conservative.
            const bool mx = inv.x < 0.0f, my = inv.y < 0.0f, mz = inv.z < 0.0f;
            const uint32_t nxw = mx ? w1.w : w1.x, fxw = mx ? w1.x : w1.w;
            const uint32_t nyw = my ? w2.x : w1.y, fyw = my ? w1.y : w2.x;
            const uint32_t nzw = mz ? w2.y : w1.z, fzw = mz ? w1.z : w2.y;
            const v2f Ax2 = {Ax, Ax}, Ay2 = {Ay, Ay}, Az2 = {Az, Az}, Bx2 = {Bx, Bx}, By2 = {By, By}, Bz2 = {Bz, Bz};
#define RFW_SLAB(i, CH)                                                                                                               \
    {                                                                                                                                 \
        const v2f qx = {(float)((nxw >> (8 * i)) & 0xffu), (float)((fxw >> (8 * i)) & 0xffu)};                                        \
        const v2f qy = {(float)((nyw >> (8 * i)) & 0xffu), (float)((fyw >> (8 * i)) & 0xffu)};                                        \
        const v2f qz = {(float)((nzw >> (8 * i)) & 0xffu), (float)((fzw >> (8 * i)) & 0xffu)};                                        \
        const v2f tx = __builtin_elementwise_fma(qx, Ax2, Bx2), ty = __builtin_elementwise_fma(qy, Ay2, By2),                         \
                  tz = __builtin_elementwise_fma(qz, Az2, Bz2);                                                                       \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), tz.x);                                                          \
        const float tf = __builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y);                                                          \
        const bool h = (tf >= tn) & (tn <= t) & (tf >= 0.0f) & (CH != kInvalidRef);                                                   \
        nhit += h ? 1u : 0u;                                                                                                          \
        hit[i] = h;                                                                                                                   \
        key[i] = (int32_t)((fbits((ANY_HIT && FAR_FIRST) ? -tf : tn) & 0xfffffffcu) | (uint32_t)i); /* slot index in the 2 LSBs: equal distances go lower slot first */                                  \
    }
            RFW_SLAB(0, ch.x)
            RFW_SLAB(1, ch.y)
            RFW_SLAB(2, ch.z)
            RFW_SLAB(3, ch.w)
#undef RFW_SLAB
            if (nhit == 0) {
                cur = kInvalidRef;
            } else {
                // one hit child (the common case near the leaves): it is the next node, no ordering needed
                cur = hit[0] ? ch.x : (hit[1] ? ch.y : (hit[2] ? ch.z : ch.w));
                if (ANY_HIT && nhit > 1 && __ballot(nhit > 2u) == 0ull) {
                    // any hit only (measured: shadow -3.7 %; the same path costs the closest-hit kernels +4…7 %, spills or not): every lane of the
                    // wavefront that has several hits has exactly two, so one compare orders them (the common case below the top of the
                    // tree); `cur` already holds the hit child of the lower slot
                    const uint32_t second = hit[3] ? ch.w : (hit[2] ? ch.z : ch.y);
                    const int32_t k_first = hit[0] ? key[0] : (hit[1] ? key[1] : key[2]);
                    const int32_t k_second = hit[3] ? key[3] : (hit[2] ? key[2] : key[1]);
                    const bool swap = bitsf((uint32_t)k_second) < bitsf((uint32_t)k_first); // the float order the any-hit sort uses
                    const uint32_t far = swap ? cur : second;
                    cur = swap ? second : cur;
                    if (sp < kStack) { lds_stack[sp * kTraceBlock + lane_slot] = far; sp++; }
                    else push(far);
                } else if (nhit > 1) {
                    // (key, child) PAIRS go through the sorting network: one compare and four selects per exchange, and the sorted children
                    // are simply there afterwards — no child look-up by index (measured against keys-with-index + look-up: closest hit
                    // -3.9 %, any hit +-0).  Misses carry the largest key and sink to the end; keys are the entry distances: compared as floats
                    // by the any-hit kernel, as integers by the closest-hit kernels (float order for the non-negative ones, some fixed order
                    // among the boxes the ray starts inside of) — each flavour measured faster in its kernel.
                    uint32_t c0 = ch.x, c1 = ch.y, c2 = ch.z, c3 = ch.w;
                    for (int i = 0; i < 4; i++) key[i] = hit[i] ? key[i] : (int32_t)0x7f7fffff; // FLT_MAX: the largest key in either order
#define RFW_PSWAP(ka, ca, kb, cb)                                                                                                     \
    {                                                                                                                                 \
        const bool s_ = ANY_HIT ? (bitsf((uint32_t)kb) < bitsf((uint32_t)ka)) : (kb < ka);                                           \
        const int32_t kl_ = s_ ? kb : ka, kh_ = s_ ? ka : kb;                                                                         \
        const uint32_t cl_ = s_ ? cb : ca, ch_ = s_ ? ca : cb;                                                                        \
        ka = kl_; kb = kh_; ca = cl_; cb = ch_;                                                                                       \
    }
                    RFW_PSWAP(key[0], c0, key[1], c1)
                    RFW_PSWAP(key[2], c2, key[3], c3)
                    RFW_PSWAP(key[0], c0, key[2], c2)
                    RFW_PSWAP(key[1], c1, key[3], c3)
                    RFW_PSWAP(key[1], c1, key[2], c2)
#undef RFW_PSWAP
                    cur = c0;
                    const int extra = (int)nhit - 1;
                    if (sp + 3 <= kStack) {
                        // far children go on first: child j (1..3 in sorted order) lands in slot sp + extra - j.  Straight LDS writes,
                        // no per-push capacity branches; with fewer than 4 hits the third write lands above the new top (never read)
                        lds_stack[(sp + extra - 1) * kTraceBlock + lane_slot] = c1;
                        if (nhit > 2) {
                            lds_stack[(sp + extra - 2) * kTraceBlock + lane_slot] = c2;
                            lds_stack[(sp + (3 <= extra ? 0 : 2)) * kTraceBlock + lane_slot] = c3;
                        }
                        sp += extra;
                    } else {
                        if (nhit > 3) push(c3);
                        if (nhit > 2) push(c2);
                        push(c1);
                    }
                }
                continue;
            }
        } else if (blas_sp >= 0) {
            // Any-hit: the compiled loop takes one stack entry per lane per iteration in lock step, and only ~7 of 64 lanes hold a leaf
            // at any one iteration (measured).  Leaves wait for the next even iteration, which batches their tests.
            if (ANY_HIT && (iteration & 1u) != 0u) continue;
            // ---- BLAS leaf: Moeller-Trumbore over the packets (intersection.glsl:1-38 / 40-70)
            const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
            const float4* tp = reinterpret_cast<const float4*>(sc.tri_packets + tri_base + first);
            for (uint32_t k = 0; k < count; k++) {
                const float4 p0 = tp[3 * k], p1 = tp[3 * k + 1], p2 = tp[3 * k + 2];
                if (COUNT) {
                    tc.tris++;
                    if (__builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u) tc.wave_tris++;
                }
                const f3 v0 = mk3(p0.x, p0.y, p0.z), edge1 = mk3(p1.x, p1.y, p1.z), edge2 = mk3(p2.x, p2.y, p2.z);
                const f3 h = cross(d, edge2);
                const float a = dot(edge1, h);
                if (a > -0.0001f && a < 0.0001f) continue;
                const float f = 1.0f / a;
                const f3 s = o - v0;
                const float u = f * dot(s, h);
                if (u < 0.0f || u > 1.0f) continue;
                const f3 q = cross(s, edge1);
                const float v = f * dot(d, q);
                if (v < 0.0f || (u + v) > 1.0f) continue;
                const float tt = f * dot(edge2, q);
                if (ANY_HIT) {
                    if (tt > t_min && tt < t) return true;
                } else {
                    const int32_t prim = (int32_t)fbits(p0.w);
                    const bool lower = (cur_inst < hit_inst) || (cur_inst == hit_inst && prim < hit_tri);
                    if (tt > t_min && (tt < t || (tt == t && hit_inst >= 0 && lower))) {
                        t = tt;
                        hu = u * p1.w;
                        hv = v * p1.w;
                        hit_inst = cur_inst;
                        hit_tri = prim;
                    }
                }
            }
            cur = kInvalidRef;
        } else {
            // ---- TLAS leaf: enter the first instance, keep the rest of the list on the stack
            const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
            if (count > 1) push(make_leaf(first + 1, count - 1));
            const uint32_t gid = sc.tlas_prims[first];
            const float4* ip = reinterpret_cast<const float4*>(sc.instances + gid);
            const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
            const uint4 meta = *reinterpret_cast<const uint4*>(ip + 3);
            if (COUNT) tc.insts++;
            // ray into object space with the inverse instance matrix; direction NOT renormalised (ray_gen.comp:340-341)
            o = xform_rows(r0, r1, r2, world_o(), 1.0f);
            d = xform_rows(r0, r1, r2, world_d(), 0.0f);
            inv = slab_inv(d);
            tri_base = meta.y;
            cur_inst = (int32_t)gid;
            nodes = sc.blas_nodes + meta.x;
            blas_sp = sp;
            cur = 0;
            continue;
        }
        // ---- next entry
        if (blas_sp >= 0 && sp == blas_sp) { // BLAS exhausted: back to world space
            blas_sp = -1;
            o = world_o();
            d = world_d();
            inv = slab_inv(d);
            nodes = sc.tlas_nodes;
        }
        if (sp == 0) break;
        cur = pop();
    }
    return false;
}

} // namespace rfwhip
