// traverse_packet.h — the same two-level BVH4 traversal as traverse.h for COHERENT wavefronts: one shared stack, one node at a time.
//
// Replaces intersect_top_mbvh / intersect_mbvh of backends/gpu-rt/shaders/ray_gen.comp:202-250,310-362 (closest hit) and
// ray_shadow.comp:83-132,191-243 (any hit) for the camera rays of an 8x8-pixel block and for the shadow rays they spawn towards one light.
//
// Why (round 4; tools/probes/packet_model.cpp on the product's own tree): the 64 rays of such a wavefront visit almost the same nodes —
// the UNION of the nodes they visit is 27 per wavefront where the one-ray-per-lane loop of traverse.h needs 25 trips for 20 visits per
// lane — and the trace kernels are bound by vector-instruction issue.  A node visited by the whole wavefront at once
//   * is fetched ONCE through the scalar cache (s_load_dwordx16 x 2 of the 128-B float node) instead of 4 vector loads per lane,
//   * is tested with its planes in scalar registers: 6 v_fma + v_max3 + v_min3 + 3 v_cmp per child (the compares write lane masks
//     straight into scalar registers), no byte conversions, no per-lane selects — the direction signs are wave-uniform inside a group —
//   * is ordered on the scalar unit (sorting network over (key, child, lane mask) in SGPRs) — no per-lane sort, no LDS stack:
//     the shared stack lives in three vector registers, entry k in lane k (v_writelane / v_readlane with a scalar index).
// Every stack entry carries the mask of the lanes whose ray hit that child's box; a popped entry runs with exec = its mask.  Control flow
// is scalar throughout (no divergence apart from the triangle test's early outs).
//
// Results are those of traverse.h bit for bit: the boxes are the same conservative boxes (de-quantised from the 64-B nodes), the triangle
// test is the same operation sequence (intersection.glsl:1-38 / 40-70), and closest-hit ties go to the lowest (instance, triangle) id, so
// the answer does not depend on which nodes are visited in which order.
#pragma once
#include "traverse.h"

namespace rfwhip {

typedef uint32_t su4 __attribute__((ext_vector_type(4)));
typedef uint32_t su16 __attribute__((ext_vector_type(16)));
// loads through these pointers are scalar loads (constant address space) when the address is wave-uniform
typedef const su4 __attribute__((address_space(4))) * scalar_ptr4;
typedef const su16 __attribute__((address_space(4))) * scalar_ptr16;
typedef const uint32_t __attribute__((address_space(4))) * scalar_ptr1;

constexpr uint32_t kPacketStack = 64; // one entry per lane of the three stack registers

RFW_DI uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
// lane `lane` of `reg` = value (both wave-uniform); v_writelane_b32 may read ONE scalar register besides M0 on gfx9, hence the move
RFW_DI void lane_write(uint32_t& reg, const uint32_t value, const uint32_t lane)
{
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(value), "s"(lane) : "m0");
}
RFW_DI uint32_t lane_read(const uint32_t reg, const uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)reg, (int)lane); }
RFW_DI uint64_t ballot64(const bool p) { return __builtin_amdgcn_ballot_w64(p); }
RFW_DI bool in_mask(const uint64_t m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
RFW_DI uint32_t first_lane(const uint64_t m) { return (uint32_t)__builtin_ctzll(m); }

// Per-lane ray of the current space for the slab test: t_plane = plane * inv + b, with b = -o * inv nudged outwards (near planes towards
// -inf, far planes towards +inf) by 2^-21 |b| — the rounding of b against the exact (plane - o) * inv, four times over.  A NaN (an
// axis-parallel ray: 0 * inf, inf - inf) is ignored by max3 / min3 and only drops that axis' constraint: conservative.
struct SlabRay {
    f3 inv, bn, bf;
};
RFW_DI SlabRay slab_ray(const f3 o, const f3 d)
{
    SlabRay r;
    r.inv = slab_inv(d);
    const f3 b = mk3(-(o.x * r.inv.x), -(o.y * r.inv.y), -(o.z * r.inv.z));
    const f3 e = mk3(__builtin_fabsf(b.x) * 4.76837158e-7f, __builtin_fabsf(b.y) * 4.76837158e-7f, __builtin_fabsf(b.z) * 4.76837158e-7f);
    r.bn = b - e;
    r.bf = b + e;
    return r;
}
RFW_DI uint32_t octant_of(const f3 inv) { return (inv.x < 0.0f ? 1u : 0u) | (inv.y < 0.0f ? 2u : 0u) | (inv.z < 0.0f ? 4u : 0u); }

// The 4-wide slab test of one octant (bit a of OCT: the direction's component a is negative, so the near plane of axis a is the upper one).
// Every lane of the wavefront runs it (an instruction costs the same with 3 active lanes as with 64, and a lane mask assigned inside a
// divergent region would not be wave-uniform to the compiler); lanes outside the entry's mask are dropped from the votes.
// a: lox[4] hix[4] loy[4] hiy[4]   b: loz[4] hiz[4] child[4] pad[4]  (the 128-B float node, in scalar registers)
template <int OCT, bool KEY_FAR>
RFW_DI void slab4(const su16 a, const su16 b, const SlabRay& r, const float t, const uint64_t mask, uint64_t (&m)[4], float (&keyv)[4])
{
    constexpr int NX = (OCT & 1) ? 4 : 0, FX = (OCT & 1) ? 0 : 4, NY = (OCT & 2) ? 12 : 8, FY = (OCT & 2) ? 8 : 12, NZ = (OCT & 4) ? 4 : 0, FZ = (OCT & 4) ? 0 : 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(bitsf(a[NX + i]), r.inv.x, r.bn.x), __builtin_fmaf(bitsf(a[NY + i]), r.inv.y, r.bn.y)),
                                         __builtin_fmaf(bitsf(b[NZ + i]), r.inv.z, r.bn.z));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(bitsf(a[FX + i]), r.inv.x, r.bf.x), __builtin_fmaf(bitsf(a[FY + i]), r.inv.y, r.bf.y)),
                                         __builtin_fmaf(bitsf(b[FZ + i]), r.inv.z, r.bf.z));
        m[i] = ballot64(tf >= tn) & ballot64(tn <= t) & ballot64(tf >= 0.0f) & mask; // three compares straight into scalar registers
        keyv[i] = KEY_FAR ? tf : tn;
    }
}

// ANY_HIT = false: t / hu / hv / hit_inst / hit_tri of every active lane describe its nearest accepted hit.  ANY_HIT = true: `occluded` per lane.
// Called by ALL 64 lanes of the wavefront in uniform control flow; `active` says which of them carry a ray.
template <bool ANY_HIT, bool COUNT, bool FAR_FIRST = false>
RFW_DI void traverse_packet(const SceneView& sc, const Node4* __restrict__ tlas_wide, const Node4* __restrict__ blas_wide, const bool active, const f3 O,
                            const f3 D, const float t_min, float& t, float& hu, float& hv, int32_t& hit_inst, int32_t& hit_tri, bool& occluded,
                            TravCounters& tc)
{
    uint32_t st_ref = 0u, st_mlo = 0u, st_mhi = 0u; // the shared stack: entry k = lane k of these three registers
    occluded = false;
    uint64_t todo = ballot64(active);
    const uint32_t lane_id = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const SlabRay world = slab_ray(O, D);
    const uint32_t world_oct = octant_of(world.inv);
    // the wavefront's rays by world-space octant: inside a group the near / far plane of every axis is a wave-uniform choice
    while (todo != 0ull) {
        const uint32_t goct = lane_read(world_oct, first_lane(todo));
        const uint64_t group = ballot64(world_oct == goct) & todo;
        todo &= ~group;
        uint64_t live = group; // any hit: lanes still looking for an occluder
        f3 o = O, d = D;
        SlabRay r = world;
        uint32_t sgn = goct;                 // direction signs of the current space (bit a: axis a negative)
        uint32_t sp = 0u;
        int32_t blas_sp = -1;                // stack height at BLAS entry; -1 = in the TLAS
        int32_t cur_inst = -1;
        uint32_t tri_base = 0u;
        const Node4* nodes = tlas_wide;
        uint32_t cur = 0u;                   // TLAS root
        uint64_t mask = group;
        auto push = [&](const uint32_t ref, const uint64_t m) {
            if (sp >= kPacketStack) { *sc.overflow_flag = 1u; return; } // dropped: the affected rays finish on what they have; the host reports RFW_HIP_E_STATE
            lane_write(st_ref, ref, sp);
            lane_write(st_mlo, (uint32_t)m, sp);
            lane_write(st_mhi, (uint32_t)(m >> 32), sp);
            sp++;
        };
        for (;;) {
            if (cur == kInvalidRef) {
                if (blas_sp >= 0 && (int32_t)sp == blas_sp) { // BLAS exhausted: the whole group back to world space
                    blas_sp = -1;
                    o = O; d = D; r = world;
                    sgn = goct;
                    nodes = tlas_wide;
                }
                if (sp == 0u) break;
                sp--;
                cur = lane_read(st_ref, sp);
                mask = (uint64_t)lane_read(st_mlo, sp) | ((uint64_t)lane_read(st_mhi, sp) << 32);
                if (ANY_HIT) {
                    mask &= live;
                    if (mask == 0ull) { cur = kInvalidRef; continue; }
                }
            }
            if (!(cur & kLeafBit)) {
                // ---- interior node: the whole packet tests the 4 child boxes; planes in scalar registers
                const scalar_ptr16 np = (scalar_ptr16)(uintptr_t)(nodes + cur);
                const su16 a = np[0], b = np[1]; // a: lox[4] hix[4] loy[4] hiy[4]   b: loz[4] hiz[4] child[4] pad[4]
                // near / far plane of every axis: a wave-uniform choice by the group's direction signs.  The test is compiled once per octant
                // (the planes are then plain scalar operands of the FMAs: no selects at all) and dispatched with a scalar branch.
                uint64_t m[4];
                float keyv[4];
                switch (sgn) {
                case 0: slab4<0, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                case 1: slab4<1, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                case 2: slab4<2, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                case 3: slab4<3, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                case 4: slab4<4, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                case 5: slab4<5, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                case 6: slab4<6, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                default: slab4<7, ANY_HIT && FAR_FIRST>(a, b, r, t, mask, m, keyv); break;
                }
                if (COUNT && in_mask(mask)) tc.nodes++;
                if (COUNT && lane_id == first_lane(mask)) { tc.wave_nodes++; tc.wave_uniform++; }
                // (key, child, lanes) of the hit children through the 5-exchange network on the scalar unit; the key is the entry distance (far
                // first: the exit distance, descending) of the first lane that hit the child, the slot index in its two low bits as tie-break
                int32_t key[4];
                uint32_t c[4] = {b[8], b[9], b[10], b[11]};
                uint32_t nhit = 0u;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    if (c[i] == kInvalidRef) m[i] = 0ull;
                    const int32_t kb = (int32_t)lane_read(fbits(keyv[i]), first_lane(m[i] | 0x8000000000000000ull));
                    const int32_t pos = kb < 0 ? 0 : kb; // a box the ray starts inside of: distance 0
                    const int32_t ordered = (int32_t)((((ANY_HIT && FAR_FIRST) ? (0x7fffffff - pos) : pos) & ~3) | i);
                    key[i] = m[i] != 0ull ? ordered : 0x7fffffff;
                    nhit += m[i] != 0ull ? 1u : 0u;
                }
                if (nhit == 0u) { cur = kInvalidRef; continue; }
                if (nhit > 1u) {
#define RFW_SSWAP(x, y)                                                                                                               \
    {                                                                                                                                 \
        const bool s_ = key[y] < key[x];                                                                                              \
        const int32_t kl_ = s_ ? key[y] : key[x], kh_ = s_ ? key[x] : key[y];                                                         \
        const uint32_t cl_ = s_ ? c[y] : c[x], ch_ = s_ ? c[x] : c[y];                                                                \
        const uint64_t ml_ = s_ ? m[y] : m[x], mh_ = s_ ? m[x] : m[y];                                                                \
        key[x] = kl_; key[y] = kh_; c[x] = cl_; c[y] = ch_; m[x] = ml_; m[y] = mh_;                                                   \
    }
                    RFW_SSWAP(0, 1)
                    RFW_SSWAP(2, 3)
                    RFW_SSWAP(0, 2)
                    RFW_SSWAP(1, 3)
                    RFW_SSWAP(1, 2)
#undef RFW_SSWAP
                    if (nhit > 3u) push(c[3], m[3]);
                    if (nhit > 2u) push(c[2], m[2]);
                    push(c[1], m[1]);
                    cur = c[0];
                    mask = m[0];
                } else {
                    // one hit child: no ordering
                    cur = m[0] ? c[0] : (m[1] ? c[1] : (m[2] ? c[2] : c[3]));
                    mask = m[0] | m[1] | m[2] | m[3];
                }
                continue;
            }
            const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
            if (blas_sp >= 0) {
                // ---- BLAS leaf: Moeller-Trumbore over the packets (intersection.glsl:1-38 / 40-70), triangle data in scalar registers
                const scalar_ptr4 tp = (scalar_ptr4)(uintptr_t)(sc.tri_packets + tri_base + first);
                if (COUNT && lane_id == first_lane(mask)) tc.wave_tris += count;
                if (in_mask(mask)) {
                    for (uint32_t k = 0; k < count; k++) {
                        const su4 q0 = tp[3 * k], q1 = tp[3 * k + 1], q2 = tp[3 * k + 2];
                        if (COUNT) tc.tris++;
                        const f3 v0 = mk3(bitsf(q0.x), bitsf(q0.y), bitsf(q0.z)), edge1 = mk3(bitsf(q1.x), bitsf(q1.y), bitsf(q1.z)), edge2 = mk3(bitsf(q2.x), bitsf(q2.y), bitsf(q2.z));
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
                            if (tt > t_min && tt < t) occluded = true;
                        } else {
                            const int32_t prim = (int32_t)q0.w;
                            const bool lower = (cur_inst < hit_inst) || (cur_inst == hit_inst && prim < hit_tri);
                            if (tt > t_min && (tt < t || (tt == t && hit_inst >= 0 && lower))) {
                                t = tt;
                                hu = u * bitsf(q1.w);
                                hv = v * bitsf(q1.w);
                                hit_inst = cur_inst;
                                hit_tri = prim;
                            }
                        }
                    }
                }
                if (ANY_HIT) {
                    live &= ~ballot64(occluded);
                    if (live == 0ull) break; // every ray of the group has its occluder
                }
                cur = kInvalidRef;
                continue;
            }
            // ---- TLAS leaf: enter the first instance of the list with the entry's lanes; the rest of the list stays on the stack
            if (count > 1u) push(make_leaf(first + 1u, count - 1u), mask);
            const uint32_t gid = *((scalar_ptr1)(uintptr_t)(sc.tlas_prims + first));
            const su16 xf = *((scalar_ptr16)(uintptr_t)(sc.instances + gid));
            uint32_t obj_oct = 0u;
            if (in_mask(mask)) {
                if (COUNT) tc.insts++;
                // ray into object space with the inverse instance matrix; direction NOT renormalised (ray_gen.comp:340-341)
                const float4 r0 = make_float4(bitsf(xf[0]), bitsf(xf[1]), bitsf(xf[2]), bitsf(xf[3]));
                const float4 r1 = make_float4(bitsf(xf[4]), bitsf(xf[5]), bitsf(xf[6]), bitsf(xf[7]));
                const float4 r2 = make_float4(bitsf(xf[8]), bitsf(xf[9]), bitsf(xf[10]), bitsf(xf[11]));
                o = xform_rows(r0, r1, r2, O, 1.0f);
                d = xform_rows(r0, r1, r2, D, 0.0f);
                r = slab_ray(o, d);
                obj_oct = octant_of(r.inv);
            }
            // lanes whose object-space direction signs differ from the first lane's come back for this instance later (an entry of their own)
            const uint32_t ooct = lane_read(obj_oct, first_lane(mask));
            const uint64_t same = ballot64(obj_oct == ooct) & mask;
            if (same != mask) push(make_leaf(first, 1u), mask & ~same);
            sgn = ooct;
            nodes = blas_wide + xf[12];
            tri_base = xf[13];
            cur_inst = (int32_t)gid;
            blas_sp = (int32_t)sp;
            cur = 0u;
            mask = same;
        }
    }
}

} // namespace rfwhip
