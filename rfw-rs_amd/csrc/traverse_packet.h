// traverse_packet.h — the same two-level BVH4 traversal as traverse.h for COHERENT wavefronts: one shared stack, one node at a time.
//
// Replaces intersect_top_mbvh / intersect_mbvh of backends/gpu-rt/shaders/ray_gen.comp:202-250,310-362 (closest hit) and
// ray_shadow.comp:83-132,191-243 (any hit) for the camera rays of an 8x8-pixel block and for the shadow rays they spawn towards one light.
//
// Why (round 4; tools/probes/packet_model.cpp on the product's own tree): the 64 rays of such a wavefront visit almost the same nodes —
// the UNION of the nodes they visit is 27 per wavefront where the one-ray-per-lane loop of traverse.h needs 25 trips for 20 visits per
// lane — and the trace kernels are bound by instruction issue.  A node visited by the whole wavefront at once
//   * is fetched ONCE through the scalar cache (s_load_dwordx16 x 2 of the 128-B float node) instead of 4 vector loads per lane,
//   * is tested with its planes as scalar operands: 6 v_fma + max3 + max + min3 + min + ONE v_cmp per child (the compare writes the lanes
//     that hit straight into a scalar register pair); no byte conversions, no per-lane selects: the direction signs are wave-uniform
//     inside a group, and the tree exists in eight copies, one per octant, with the near / far planes already chosen,
//   * needs no ordering at run time: the children of an octant's copy are stored front to back along the octant's diagonal — no per-lane
//     sort, no LDS stack: the shared stack is ONE vector register, entry k in lane k (v_writelane / v_readlane with a scalar index).
// A child is visited when ANY ray of the packet hits its box; rays that miss it run along (an instruction costs the same with 3 active
// lanes as with 64) and fail the children's tests.  Control flow is scalar throughout, apart from the triangle test's early outs.
//
// Results are those of traverse.h bit for bit: the boxes are conservative (the boxes the 64-B nodes encode, as floats), the triangle test
// is the same operation sequence (intersection.glsl:1-38 / 40-70), and closest-hit ties go to the lowest (instance, triangle) id, so the
// answer does not depend on which nodes are visited, by whom, in which order.
#pragma once
#include "traverse.h"

namespace rfwhip {

typedef uint32_t su4 __attribute__((ext_vector_type(4)));
typedef uint32_t su16 __attribute__((ext_vector_type(16)));
// loads through these pointers are scalar loads (constant address space) when the address is wave-uniform
typedef const su4 __attribute__((address_space(4))) * scalar_ptr4;
typedef const su16 __attribute__((address_space(4))) * scalar_ptr16;
typedef const uint32_t __attribute__((address_space(4))) * scalar_ptr1;

constexpr uint32_t kPacketStack = 64; // one entry per lane of the stack register

// lane `lane` of `reg` = value (both wave-uniform).  v_writelane_b32 may read ONE scalar register besides M0 on gfx9, so the lane index has
// to travel through M0 — and the COMPILER puts it there: this is the LLVM intrinsic itself (this clang has no __builtin_amdgcn_writelane;
// the declaration binds the intrinsic by name, as the HIP headers do for theirs).  Round 4's inline asm wrote M0 behind the compiler's back.
extern "C" __device__ int rfw_llvm_writelane(int value, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
RFW_DI void lane_write(uint32_t& reg, const uint32_t value, const uint32_t lane)
{
    reg = (uint32_t)rfw_llvm_writelane((int)value, (int)lane, (int)reg);
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

// The 4-wide slab test.  The node is the copy made for the packet's octant (PacketNode below): near / far planes already chosen by the
// direction signs, children sorted front to back along the octant's diagonal.  Every lane of the wavefront runs the test; `packet` drops the
// lanes that are not part of the packet from the votes.  An empty slot holds the box (+inf, -inf): its entry distance is +inf for every
// ray, so it needs no test of its own.
// a: nx[4] ny[4] nz[4] fx[4]   b: fy[4] fz[4] child[4] pad[4]  (the 128-B node, in scalar registers)
// m[i] = lanes whose ray enters child i before t (6 fma + max3 + max + min3 + min + 1 compare per child)
template <bool MASK> RFW_DI void slab4(const su16 a, const su16 b, const SlabRay& r, const float t, const uint64_t packet, uint64_t (&m)[4])
{
#pragma unroll
    for (int i = 0; i < 4; i++) {
        // tf >= tn && tn <= t  <=>  min(tf, t) >= tn; a direction of exact zeros makes tn NaN on every axis and fails the compare
        const float tn = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaf(bitsf(a[i]), r.inv.x, r.bn.x), __builtin_fmaf(bitsf(a[4 + i]), r.inv.y, r.bn.y)),
                                         __builtin_fmaf(bitsf(a[8 + i]), r.inv.z, r.bn.z));
        const float tf = __builtin_fminf(__builtin_fminf(__builtin_fmaf(bitsf(a[12 + i]), r.inv.x, r.bf.x), __builtin_fmaf(bitsf(b[i]), r.inv.y, r.bf.y)),
                                         __builtin_fmaf(bitsf(b[4 + i]), r.inv.z, r.bf.z));
        // min(tf, t) >= max(tn, 0)  <=>  min(tf, t) >= tn  &&  tf >= 0  (t >= 0): one vector max instead of a compare and a scalar AND
        m[i] = ballot64(__builtin_fminf(tf, t) >= __builtin_fmaxf(tn, 0.0f));
        if (MASK) m[i] &= packet;
    }
}

// ANY_HIT = false: t / hu / hv / hit_inst / hit_tri of every active lane describe its nearest accepted hit.  ANY_HIT = true: `occluded` per lane.
// Called by ALL 64 lanes of the wavefront in uniform control flow; `active` says which of them carry a ray.
template <bool ANY_HIT, bool COUNT, bool FAR_FIRST = false>
RFW_DI void traverse_packet(const SceneView& sc, const PacketNode* __restrict__ tlas_wide, const uint32_t tlas_stride, const PacketNode* __restrict__ blas_wide,
                            const uint32_t blas_stride, const bool active, const f3 O, const f3 D, const float t_min, float& t, float& hu, float& hv, int32_t& hit_inst, int32_t& hit_tri, bool& occluded,
                            TravCounters& tc)
{
    uint32_t stack = 0u; // the shared stack: entry k = lane k of this register
    occluded = false;
    // a direction of exact zeros or with a NaN makes every plane distance NaN, and NaNs are IGNORED by min / max: with the single compare of
    // slab4 such a ray would enter every box.  It hits nothing (as in traverse()): it is not part of any packet
    const bool sane = (D.x == D.x) && (D.y == D.y) && (D.z == D.z) && (D.x != 0.0f || D.y != 0.0f || D.z != 0.0f);
    uint64_t todo = ballot64(active && sane);
    const uint32_t lane_id = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t world_oct = octant_of(slab_inv(D));
    if (COUNT && active) tc.nodes++; // the root
    // the wavefront's rays by world-space octant: inside a group the near / far plane of every axis is a wave-uniform choice
    while (todo != 0ull) {
        const uint32_t goct = lane_read(world_oct, first_lane(todo));
        const uint64_t group = ballot64(world_oct == goct) & todo;
        todo &= ~group;
        uint64_t live = group;               // lanes whose votes count (any hit: minus the rays that have found their occluder)
        uint64_t later = 0ull;               // lanes that wait to enter the current instance with another object-space octant
        uint32_t sp = 0u;
        uint32_t blas_sp = 0u;               // stack height at BLAS entry
        bool in_blas = false;
        uint32_t inst_first = 0u;            // the instance being traversed: its slot in the TLAS leaf list
        uint32_t cur = 0u;                   // TLAS root
        // ---- one pass of this loop per SPACE the packet is in: the rays' coordinates change only here, never inside the traversal loop
        for (;;) {
            f3 o = O, d = D;
            uint32_t sgn = goct;             // direction signs of the current space (bit a: axis a negative)
            const PacketNode* nodes = tlas_wide + (size_t)goct * tlas_stride; // the copy of the tree made for this octant
            uint32_t tri_base = 0u;
            int32_t cur_inst = -1;
            uint64_t packet = live;
            if (in_blas) {
                // enter the instance with the waiting lanes whose object-space octant is the first such lane's; the others wait in `later`
                const uint32_t gid = *((scalar_ptr1)(uintptr_t)(sc.tlas_prims + inst_first));
                const su16 xf = *((scalar_ptr16)(uintptr_t)(sc.instances + gid));
                const float4 r0 = make_float4(bitsf(xf[0]), bitsf(xf[1]), bitsf(xf[2]), bitsf(xf[3]));
                const float4 r1 = make_float4(bitsf(xf[4]), bitsf(xf[5]), bitsf(xf[6]), bitsf(xf[7]));
                const float4 r2 = make_float4(bitsf(xf[8]), bitsf(xf[9]), bitsf(xf[10]), bitsf(xf[11]));
                // ray into object space with the inverse instance matrix; direction NOT renormalised (ray_gen.comp:340-341)
                if (xf[14] & kInstanceIdentity) { // (wave-uniform: traverse_flat.inc)
                    o = mk3(O.x + 0.0f, O.y + 0.0f, O.z + 0.0f);
                    d = mk3(D.x + 0.0f, D.y + 0.0f, D.z + 0.0f);
                } else {
                    o = xform_rows(r0, r1, r2, O, 1.0f);
                    d = xform_rows(r0, r1, r2, D, 0.0f);
                }
                const uint32_t obj_oct = octant_of(slab_inv(d));
                const uint64_t want = later != 0ull ? later : live;
                sgn = lane_read(obj_oct, first_lane(want));
                packet = ballot64(obj_oct == sgn) & want;
                later = want & ~packet;
                nodes = blas_wide + (size_t)sgn * blas_stride + xf[12];
                tri_base = xf[13];
                cur_inst = (int32_t)gid;
                if (COUNT && in_mask(packet)) { tc.insts++; tc.nodes++; }
                cur = 0u;
            }
            SlabRay r = slab_ray(o, d);
            constexpr bool kNanMask = !ANY_HIT; // closest hit: the lanes outside the packet carry a ray that fails every compare instead of being masked out of every vote (any hit: the packet shrinks inside the loop as rays find their occluders)
            // a lane outside the packet leaves every box at -inf, before it could enter it: min(-inf, t) >= max(tn, 0) is false whatever tn is.
            // Its reciprocal direction is +-1 with the OCTANT's signs, so that the far plane of an EMPTY slot — -inf on an axis the octant
            // travels up, +inf on one it travels down — times it is -inf as well: with +1 throughout, (+inf) * 1 + (-inf) was a NaN, min / max
            // ignore NaNs, and such lanes voted for the empty slots of the octants with a negative axis (harmless — the pop skips
            // kInvalidRef — but every such vote was a wasted push on the shared stack)
            if (kNanMask && !in_mask(packet)) {
                r.inv = mk3((sgn & 1u) ? -1.0f : 1.0f, (sgn & 2u) ? -1.0f : 1.0f, (sgn & 4u) ? -1.0f : 1.0f);
                r.bf = mk3(-INFINITY);
            }
            const uint32_t lead = first_lane(packet); // (COUNT: the lane that counts the wavefront-level steps)
            bool leave = false, finished = false;
            const uint32_t floor_sp = in_blas ? blas_sp : 0u; // stack height at which this space is exhausted
            // next entry off the shared stack (kInvalidRef: the space is exhausted); an empty slot that was pushed (a NaN corner of the slab
            // test) comes back as kInvalidRef and is skipped
            auto pop = [&]() {
                cur = kInvalidRef;
                if (sp != floor_sp) {
                    sp--;
                    cur = lane_read(stack, sp);
                    while (__builtin_expect(cur == kInvalidRef && sp != floor_sp, 0)) { // (an empty slot somebody voted for: a NaN corner of the slab test; lanes outside the packet no longer produce one)
                        sp--;
                        cur = lane_read(stack, sp);
                    }
                }
            };
            if (cur == kInvalidRef) pop(); // back in world space behind an instance: carry on with what the stack holds
            // ---- the traversal loop of one space
            for (;;) {
                // -- the hot loop: interior nodes until a leaf comes off the stack (nothing about the rays' hits changes in here)
                while (cur != kInvalidRef && !(cur & kLeafBit)) {
                    // the whole packet tests the 4 child boxes; planes in scalar registers
                    const scalar_ptr16 np = (scalar_ptr16)(uintptr_t)(nodes + cur);
                    const su16 a = np[0], b = np[1];
                    uint64_t m[4];
                    slab4<!kNanMask>(a, b, r, t, packet, m);
                    const uint32_t c[4] = {b[8], b[9], b[10], b[11]};
                    if (COUNT) {
                        // per lane: the nodes and triangles a traversal of its own would go on to visit (the children whose boxes IT hits)
#pragma unroll
                        for (int i = 0; i < 4; i++)
                            if (in_mask(m[i])) {
                                if (c[i] & kLeafBit) tc.tris += in_blas ? ((c[i] >> 27) & 15u) + 1u : 0u;
                                else tc.nodes++;
                            }
                        if (lane_id == lead) { tc.wave_nodes++; tc.wave_uniform++; }
                    }
                    // The children are stored front to back for this octant: the hit ones go on the stack in reverse (far first: in forward) order,
                    // without a branch — every child is written to the top slot and the slot only advances past the hit ones — and the
                    // nearest comes straight back off the top.
                    // (a stack that would not take four more entries: flagged — the host reports RFW_HIP_E_STATE — and the top entries are
                    // overwritten; the pushes themselves stay unconditional: as the else-branch of this test they made the compiler keep two
                    // copies of the stack register and move one into the other on every trip)
                    if (__builtin_expect(sp + 4u > kPacketStack, 0)) { *sc.overflow_flag = 1u; sp = kPacketStack - 4u; }
                    // ... except the one that would come straight back: when the first child of the visiting order is hit it IS the next node
#pragma unroll
                    for (int j = 0; j < 3; j++) {
                        const int i = (ANY_HIT && FAR_FIRST) ? j : 3 - j;
                        lane_write(stack, c[i], sp);
                        asm("s_cmp_lg_u64 %1, 0\n\ts_addc_u32 %0, %0, 0" : "+s"(sp) : "s"(m[i]) : "scc"); // sp += (m[i] != 0), on the scalar unit (the compiler converts the bool on the vector unit)
                    }
                    constexpr int kFirst = (ANY_HIT && FAR_FIRST) ? 3 : 0;
                    if (m[kFirst] != 0ull) cur = c[kFirst];
                    else pop();
                }
                if (cur == kInvalidRef) { // the space is exhausted
                    if (in_blas) leave = true;
                    else finished = true;
                    break;
                }
                if (!in_blas) break; // a TLAS leaf: the space changes
                // ---- BLAS leaf: Moeller-Trumbore over the packets (intersection.glsl:1-38 / 40-70), triangle data in scalar registers
                const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
                const scalar_ptr4 tp = (scalar_ptr4)(uintptr_t)(sc.tri_packets + tri_base + first);
                if (COUNT && lane_id == lead) tc.wave_tris += count;
                if (in_mask(packet)) {
                    for (uint32_t k = 0; k < count; k++) {
                        const su4 q0 = tp[3 * k], q1 = tp[3 * k + 1], q2 = tp[3 * k + 2];
                        const f3 v0 = mk3(bitsf(q0.x), bitsf(q0.y), bitsf(q0.z)), edge1 = mk3(bitsf(q1.x), bitsf(q1.y), bitsf(q1.z)), edge2 = mk3(bitsf(q2.x), bitsf(q2.y), bitsf(q2.z));
                        // (without the early outs: traverse_leaf.inc)
                        const f3 h = cross(d, edge2);
                        const float a = dot(edge1, h);
                        const float f = 1.0f / a;
                        const f3 s = o - v0;
                        const float u = f * dot(s, h);
                        const f3 q = cross(s, edge1);
                        const float v = f * dot(d, q);
                        const float tt = f * dot(edge2, q);
                        const bool ok = !((a > -0.0001f) & (a < 0.0001f)) & !((u < 0.0f) | (u > 1.0f)) & !((v < 0.0f) | ((u + v) > 1.0f));
                        if (ANY_HIT) {
                            occluded = occluded | (ok & (tt > t_min) & (tt < t));
                        } else {
                            const int32_t prim = (int32_t)q0.w;
                            const bool lower = (cur_inst < hit_inst) | ((cur_inst == hit_inst) & (prim < hit_tri));
                            const bool take = ok & (tt > t_min) & ((tt < t) | ((tt == t) & (hit_inst >= 0) & lower));
                            t = take ? tt : t;
                            hu = take ? u * bitsf(q1.w) : hu;
                            hv = take ? v * bitsf(q1.w) : hv;
                            hit_inst = take ? cur_inst : hit_inst;
                            hit_tri = take ? prim : hit_tri;
                        }
                    }
                }
                if (ANY_HIT) {
                    const uint64_t found = ballot64(occluded);
                    live &= ~found;
                    packet &= ~found;
                    if (packet == 0ull) { leave = true; break; } // every ray of this packet has its occluder
                }
                pop();
            }
            if (finished) break;
            if (leave) {
                // out of the instance (exhausted, or nobody left to look): unwind to the entry height; lanes of another octant enter next
                sp = blas_sp;
                cur = kInvalidRef;
                if (ANY_HIT) later &= live;
                if (later == 0ull) in_blas = false;
                if (ANY_HIT && live == 0ull) break;
                continue;
            }
            // ---- TLAS leaf `cur`: its first instance is entered next, the rest of the list stays on the stack
            {
                const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
                if (count > 1u) {
                    if (sp >= kPacketStack) *sc.overflow_flag = 1u;
                    else { lane_write(stack, make_leaf(first + 1u, count - 1u), sp); sp++; }
                }
                inst_first = first;
                in_blas = true;
                later = 0ull;
                blas_sp = sp;
            }
        }
    }
}

} // namespace rfwhip
