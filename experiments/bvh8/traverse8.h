// traverse8.h — two-level (TLAS of instances -> per-mesh BLAS) traversal of the 8-wide compressed nodes of bvh8.h, one ray per lane.
//
// Replaces intersect_top_mbvh / intersect_mbvh of backends/gpu-rt/shaders/ray_gen.comp:202-250,310-362 (closest hit) and
// ray_shadow.comp:83-132,191-243 (any hit) — same results (the per-triangle arithmetic is intersection.glsl:1-38 / 40-70 operation for
// operation, exact ties go to the lowest (instance, triangle) id), different machine:
//   * a traversal step is a DEPENDENT chain (address -> 5 loads of 16 B -> 8 slab tests -> next address) and the kernels are bound by that
//     latency, not by a unit: 8 children per step instead of 4 makes the chain ~35-45 % shorter;
//   * no ordering arithmetic: children sit in octant slots, the ray's direction signs say in which order it wants them (slot ^ octant), and the
//     hit children of a node are ONE stack entry (first interior child, hit mask in visiting order, interior mask) — the reference sorts four
//     distances per node and keeps two 32-entry arrays per thread in scratch memory (ray_gen.comp:204,312);
//   * the leaves of a node are tested when the node is visited: their triangles are consecutive 48-B packets, one bit each in a per-lane
//     mask, so a lane's triangle tests of one node run back to back;
//   * one loop, one stack for both levels: a TLAS leaf (instances) becomes a stack entry, entering it switches the lane's ray to object
//     space and remembers the stack height to switch back.
#pragma once
#include "bvh8.h"
#include "device_math.h"
#include "device_types.h"

#ifndef RFW_TRI_BATCH
#define RFW_TRI_BATCH 16 // lanes of a wavefront that must hold pending triangles before the triangle test runs (one triangle per lane)
#endif

namespace rfwhip {

constexpr int kTraceBlock = 64;    // threads per workgroup of the trace kernels (one wavefront)
constexpr int kStackLds = 10;      // stack ENTRIES (two words each) per lane kept in LDS, closest hit: 20 + 6 rows = 6.5 KB per wavefront
constexpr int kStackLdsAny = 7;    // the same for any hit: 14 + 6 rows = 5 KB per wavefront (8 waves per SIMD fit the 160 KB)
constexpr int kStackSpill = 48;    // further WORDS per lane in HBM (24 entries; rarely touched)
constexpr int kParkRows = 6;       // the world-space ray, parked in LDS above the stack while an instance is traversed

struct SceneView {
    const Node8* tlas_nodes;
    const uint32_t* tlas_prims; // instance ids in the order the TLAS' leaf slots address them
    const InstanceXform* instances;
    const Node8* blas_nodes;
    const TriPacket* tri_packets;
    uint32_t* spill;            // kStackSpill x spill_stride
    uint32_t spill_stride;
    uint32_t spill_rows;        // rows (words) of `spill` a lane may use
    uint32_t stack_entries;     // LDS stack entries a lane may use (>= the kernel's own count normally; a test option lowers it to exercise the spill and overflow paths)
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
    uint32_t wave_uniform = 0; // node-test executions whose active lanes all visit one node
};

RFW_DI bool first_active_lane()
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(__ballot(1) >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)__ballot(1), 0u)) == 0u;
}

// the 8 bits of x with bit j moved to bit j ^ k (k = 0..7): the hit mask in slot order -> in the order this ray visits the slots
RFW_DI uint32_t xor_permute8(uint32_t x, const uint32_t k)
{
    const uint32_t a = ((x & 0x55u) << 1) | ((x >> 1) & 0x55u);
    x = (k & 1u) ? a : x;
    const uint32_t b = ((x & 0x33u) << 2) | ((x >> 2) & 0x33u);
    x = (k & 2u) ? b : x;
    const uint32_t c = ((x & 0x0fu) << 4) | (x >> 4);
    x = (k & 4u) ? c : x;
    return x;
}

// Closest hit (ANY_HIT = false): on return t / hu / hv / hit_inst / hit_tri describe the nearest accepted hit, ties resolved to the lowest
// (instance, triangle) id.  Any hit (ANY_HIT = true): returns true as soon as one triangle has t_min < t' < t.
// far_first (any hit): the hit children of a node are visited in the REVERSE octant order — the search for an occluder starts at the far
// end of the ray.  The triangle tests, and therefore the answer, are the same in either order.
template <bool ANY_HIT, bool COUNT>
RFW_DI bool traverse(const SceneView& sc, const f3 O, const f3 D, const float t_min, float& t, float& hu, float& hv, int32_t& hit_inst, int32_t& hit_tri,
                     uint32_t* lds_stack, const uint32_t lane_slot, const uint32_t spill_slot, TravCounters& tc, const bool far_first = false)
{
    constexpr int kStackRows = ANY_HIT ? kStackLdsAny : kStackLds; // LDS stack entries of this kernel flavour
    const int kStack = (int)sc.stack_entries < kStackRows ? (int)sc.stack_entries : kStackRows; // wave-uniform
    uint32_t* const park = lds_stack + 2 * kStackRows * kTraceBlock + lane_slot;
    park[0] = fbits(O.x); park[kTraceBlock] = fbits(O.y); park[2 * kTraceBlock] = fbits(O.z);
    park[3 * kTraceBlock] = fbits(D.x); park[4 * kTraceBlock] = fbits(D.y); park[5 * kTraceBlock] = fbits(D.z);
    auto world_o = [&]() -> f3 { return mk3(bitsf(park[0]), bitsf(park[kTraceBlock]), bitsf(park[2 * kTraceBlock])); };
    auto world_d = [&]() -> f3 { return mk3(bitsf(park[3 * kTraceBlock]), bitsf(park[4 * kTraceBlock]), bitsf(park[5 * kTraceBlock])); };

    f3 o = O, d = D;
    f3 inv = slab_inv(d);
    // slot = (leading zeros of the hit byte) ^ kk: near first kk = octant of the direction (bit a: d_a < 0), far first its complement
    const uint32_t flip = far_first ? 7u : 0u;
    uint32_t kk = ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)) ^ flip;
    int sp = 0;                // stack height in entries
    int blas_sp = -1;          // stack height at BLAS entry; -1 = currently in the TLAS
    int32_t cur_inst = -1;
    uint32_t tri_base = 0;     // first packet of the entered instance's mesh
    const Node8* nodes = sc.tlas_nodes;
    // the current group: first interior child of the node it came from | hit children still to visit (bits 31..24, in visiting order) and the
    // node's interior mask (bits 7..0).  Start: the root as the only child of a pseudo node
    uint32_t g_base = 0u, g_bits = 0x80000000u;

    auto push = [&](const uint32_t a, const uint32_t b) {
        if (sp < kStack) {
            lds_stack[(2 * sp) * kTraceBlock + lane_slot] = a;
            lds_stack[(2 * sp + 1) * kTraceBlock + lane_slot] = b;
        } else if (2 * (sp - kStack) + 1 < (int)sc.spill_rows) {
            sc.spill[(size_t)(2 * (sp - kStack)) * sc.spill_stride + spill_slot] = a;
            sc.spill[(size_t)(2 * (sp - kStack) + 1) * sc.spill_stride + spill_slot] = b;
        } else {
            // overflow (a tree deeper than LDS + spill rows): the entry is dropped and sp does NOT advance, so a pop never indexes past the
            // spill rows — the ray finishes deterministically on what it has (possibly missing a hit), and the host reports RFW_HIP_E_STATE
            // from the next render / read / query (the flag lives in pinned host memory: no read-back needed)
            *sc.overflow_flag = 1u;
            return;
        }
        sp++;
    };
    auto pop = [&](uint32_t& a, uint32_t& b) {
        sp--;
        if (__builtin_expect(sp >= kStack, 0)) {
            a = sc.spill[(size_t)(2 * (sp - kStack)) * sc.spill_stride + spill_slot];
            b = sc.spill[(size_t)(2 * (sp - kStack) + 1) * sc.spill_stride + spill_slot];
        } else {
            a = lds_stack[(2 * sp) * kTraceBlock + lane_slot];
            b = lds_stack[(2 * sp + 1) * kTraceBlock + lane_slot];
        }
    };

    // Pending triangles: the leaves of a node are NOT tested when the node is visited (a wavefront would run the triangle test with the few
    // lanes that happen to have hit a leaf in this very step: measured 0.18 lane utilisation) but remembered — packet base + one bit per
    // triangle — and tested ONE per lane and loop iteration whenever at least RFW_TRI_BATCH lanes of the wavefront hold some (or a lane
    // cannot go on without: it is about to leave the instance the triangles belong to, or has nothing else left).
    uint32_t p_base = 0u, p_mask = 0u;
    auto test_pending_triangle = [&]() -> bool { // tests the lowest pending triangle of this lane; true = any-hit found
        const uint32_t k = (uint32_t)__builtin_ctz(p_mask);
        p_mask &= p_mask - 1u;
        const float4* tp = reinterpret_cast<const float4*>(sc.tri_packets + p_base + k);
        const float4 p0 = tp[0], p1 = tp[1], p2 = tp[2];
        if (COUNT) {
            tc.tris++;
            if (first_active_lane()) tc.wave_tris++;
        }
        const f3 v0 = mk3(p0.x, p0.y, p0.z), edge1 = mk3(p1.x, p1.y, p1.z), edge2 = mk3(p2.x, p2.y, p2.z);
        const f3 h = cross(d, edge2);
        const float a = dot(edge1, h);
        if (a > -0.0001f && a < 0.0001f) return false;
        const float f = 1.0f / a;
        const f3 s = o - v0;
        const float u = f * dot(s, h);
        if (u < 0.0f || u > 1.0f) return false;
        const f3 q = cross(s, edge1);
        const float v = f * dot(d, q);
        if (v < 0.0f || (u + v) > 1.0f) return false;
        const float tt = f * dot(edge2, q);
        if (ANY_HIT) return tt > t_min && tt < t;
        const int32_t prim = (int32_t)fbits(p0.w);
        const bool lower = (cur_inst < hit_inst) || (cur_inst == hit_inst && prim < hit_tri);
        if (tt > t_min && (tt < t || (tt == t && hit_inst >= 0 && lower))) {
            t = tt;
            hu = u * p1.w;
            hv = v * p1.w;
            hit_inst = cur_inst;
            hit_tri = prim;
        }
        return false;
    };

    for (;;) {
        // ---- next node: the first remaining child of the current group, or of the group on top of the stack.  A lane whose pending
        // triangles belong to the instance it is about to leave (or that has nothing else left) stalls here until they are tested.
        bool has_node = false;
        for (;;) {
            if ((g_bits >> 24) != 0u) { has_node = true; break; }
            if (blas_sp >= 0 && sp == blas_sp) { // BLAS exhausted: back to world space
                if (p_mask != 0u) break;
                blas_sp = -1;
                o = world_o();
                d = world_d();
                inv = slab_inv(d);
                kk = ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)) ^ flip;
                nodes = sc.tlas_nodes;
            }
            if (sp == 0) return false; // (no triangle is pending outside an instance)
            uint32_t a, b;
            pop(a, b);
            if ((b >> 24) != 0u) { g_base = a; g_bits = b; has_node = true; break; }
            // ---- a TLAS leaf's instances (first, count): enter the first, keep the rest
            if (b > 1u) push(a + 1u, b - 1u);
            const uint32_t gid = sc.tlas_prims[a];
            const float4* ip = reinterpret_cast<const float4*>(sc.instances + gid);
            const float4 r0 = ip[0], r1 = ip[1], r2 = ip[2];
            const uint4 meta = *reinterpret_cast<const uint4*>(ip + 3);
            if (COUNT) tc.insts++;
            // ray into object space with the inverse instance matrix; direction NOT renormalised (ray_gen.comp:340-341)
            o = xform_rows(r0, r1, r2, world_o(), 1.0f);
            d = xform_rows(r0, r1, r2, world_d(), 0.0f);
            inv = slab_inv(d);
            kk = ((d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u)) ^ flip;
            tri_base = meta.y;
            cur_inst = (int32_t)gid;
            nodes = sc.blas_nodes + meta.x;
            blas_sp = sp;
            g_base = 0u;
            g_bits = 0x80000000u;
            has_node = true;
            break;
        }
        if (has_node) {
        const uint32_t lz = (uint32_t)__builtin_clz(g_bits); // 0..7
        g_bits &= ~(0x80000000u >> lz);
        const uint32_t slot = lz ^ kk;
        const uint32_t node_index = g_base + (uint32_t)__builtin_popcount(g_bits & ((1u << slot) - 1u) & 0xffu);

        // ---- the node: 8 slab tests on the quantised child boxes (80 B = 5 dwordx4 per lane)
        const uint4* np = reinterpret_cast<const uint4*>(nodes + node_index);
        const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3], w4 = np[4];
        if (COUNT) {
            tc.nodes++;
            if (first_active_lane()) tc.wave_nodes++;
            const uint64_t first_ptr = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)((uintptr_t)np >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uintptr_t)np);
            const bool all_same = __ballot((uintptr_t)np == first_ptr) == __ballot(1);
            if (all_same && first_active_lane()) tc.wave_uniform++;
        }
        const uint32_t imask = w0.w >> 24;
        // plane = origin + q * scale  =>  t = q * (scale * inv) + (origin - o) * inv : one conversion + one fma per plane
        const float Ax = bitsf((w0.w & 0xffu) << 23) * inv.x, Ay = bitsf(((w0.w >> 8) & 0xffu) << 23) * inv.y, Az = bitsf(((w0.w >> 16) & 0xffu) << 23) * inv.z;
        const float Bx = (bitsf(w0.x) - o.x) * inv.x, By = (bitsf(w0.y) - o.y) * inv.y, Bz = (bitsf(w0.z) - o.z) * inv.z;
        // the ray's direction signs pick the near and the far plane of each axis: a child costs 6 conversions, 3 packed FMAs (near and far
        // share scale and offset), one max3, one min3.  A NaN (0 * inf on an axis-parallel ray) is ignored by max3 / min3 and only drops that
        // axis' constraint: conservative.
        const bool mx = inv.x < 0.0f, my = inv.y < 0.0f, mz = inv.z < 0.0f;
        const uint32_t nx0 = mx ? w3.z : w2.x, nx1 = mx ? w3.w : w2.y, fx0 = mx ? w2.x : w3.z, fx1 = mx ? w2.y : w3.w;
        const uint32_t ny0 = my ? w4.x : w2.z, ny1 = my ? w4.y : w2.w, fy0 = my ? w2.z : w4.x, fy1 = my ? w2.w : w4.y;
        const uint32_t nz0 = mz ? w4.z : w3.x, nz1 = mz ? w4.w : w3.y, fz0 = mz ? w3.x : w4.z, fz1 = mz ? w3.y : w4.w;
        const v2f Ax2 = {Ax, Ax}, Ay2 = {Ay, Ay}, Az2 = {Az, Az}, Bx2 = {Bx, Bx}, By2 = {By, By}, Bz2 = {Bz, Bz};
        uint32_t hm = 0u; // hit mask, slot order
#define RFW_SLAB8(NXW, FXW, NYW, FYW, NZW, FZW, i, s)                                                                                  \
    {                                                                                                                                 \
        const v2f qx = {(float)((NXW >> (8 * i)) & 0xffu), (float)((FXW >> (8 * i)) & 0xffu)};                                        \
        const v2f qy = {(float)((NYW >> (8 * i)) & 0xffu), (float)((FYW >> (8 * i)) & 0xffu)};                                        \
        const v2f qz = {(float)((NZW >> (8 * i)) & 0xffu), (float)((FZW >> (8 * i)) & 0xffu)};                                        \
        const v2f tx = __builtin_elementwise_fma(qx, Ax2, Bx2), ty = __builtin_elementwise_fma(qy, Ay2, By2),                         \
                  tz = __builtin_elementwise_fma(qz, Az2, Bz2);                                                                       \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), tz.x);                                                          \
        const float tf = __builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y);                                                          \
        const bool h = __builtin_fmaxf(tn, 0.0f) <= __builtin_fminf(tf, t);                                                           \
        hm |= h ? (1u << s) : 0u;                                                                                                     \
    }
        RFW_SLAB8(nx0, fx0, ny0, fy0, nz0, fz0, 0, 0)
        RFW_SLAB8(nx0, fx0, ny0, fy0, nz0, fz0, 1, 1)
        RFW_SLAB8(nx0, fx0, ny0, fy0, nz0, fz0, 2, 2)
        RFW_SLAB8(nx0, fx0, ny0, fy0, nz0, fz0, 3, 3)
        RFW_SLAB8(nx1, fx1, ny1, fy1, nz1, fz1, 0, 4)
        RFW_SLAB8(nx1, fx1, ny1, fy1, nz1, fz1, 1, 5)
        RFW_SLAB8(nx1, fx1, ny1, fy1, nz1, fz1, 2, 6)
        RFW_SLAB8(nx1, fx1, ny1, fy1, nz1, fz1, 3, 7)
#undef RFW_SLAB8
        // the hit interior children as the next group, in this ray's visiting order: bit 31 - j <- slot j ^ kk
        const uint32_t inner = xor_permute8(hm & imask, kk ^ 7u);
        uint32_t leaves = hm & ~imask;
        bool tlas_leaves = false;
        if (leaves != 0u) {
            if (blas_sp >= 0) {
                // ---- BLAS leaves: one bit per triangle of the hit leaf slots (byte s of the meta words: count << 5 | offset)
                uint32_t tmask = 0u;
                do {
                    const uint32_t s = (uint32_t)__builtin_ctz(leaves);
                    leaves &= leaves - 1u;
                    const uint32_t m = ((s & 4u) ? w1.w : w1.z) >> (8u * (s & 3u)) & 0xffu;
                    tmask |= ((1u << (m >> 5)) - 1u) << (m & 31u); // an empty slot (m = 0: reached only by degenerate rays) adds nothing
                } while (leaves != 0u);
                if (tmask != 0u) {
                    while (p_mask != 0u) // the lane still holds triangles of an earlier node: they go first (rare: they are tested every few steps)
                        if (test_pending_triangle()) return true;
                    p_base = tri_base + w1.y;
                    p_mask = tmask;
                }
            } else {
                // ---- TLAS leaves: the group's remaining children wait below the instances of this node (each leaf slot one entry:
                // first index into tlas_prims, count)
                if ((g_bits >> 24) != 0u) push(g_base, g_bits);
                g_bits = 0u;
                if (inner != 0u) push(w1.x, (inner << 24) | imask);
                do {
                    const uint32_t s = (uint32_t)__builtin_ctz(leaves);
                    leaves &= leaves - 1u;
                    const uint32_t m = ((s & 4u) ? w1.w : w1.z) >> (8u * (s & 3u)) & 0xffu;
                    if (m != 0u) push(w1.y + (m & 31u), m >> 5);
                } while (leaves != 0u);
                tlas_leaves = true;
            }
        }
        if (!tlas_leaves && inner != 0u) {
            if ((g_bits >> 24) != 0u) push(g_base, g_bits); // the rest of the current group waits
            g_base = w1.x;
            g_bits = (inner << 24) | imask;
        }
        }
        // ---- pending triangles: one per lane when enough lanes hold some, or when a lane cannot go on without
        const bool want = p_mask != 0u;
        const unsigned long long wanting = __ballot(want);
        if (wanting != 0ull && (__popcll(wanting) >= RFW_TRI_BATCH || __ballot(want && !has_node) != 0ull)) {
            if (want && test_pending_triangle()) return true;
        }
    }
}

} // namespace rfwhip
