// bvh8.h — the 8-wide compressed node the traversal kernels read, and the collapse of a 4-wide tree into it.
//
// Replaces MBVH::construct of backends/gpu-rt/src/lib.rs:1581 / :1411 (rtbvh's 4-wide collapse, 128-B nodes, structs.glsl:56-65) as the
// LAST step of every builder: binned SAH on the device or on the host and LBVH all emit the f32 4-wide Node4 tree (device_types.h);
// this header turns that tree into 80-byte 8-wide nodes (Ylitie, Karras, Laine 2017, "Efficient incoherent ray traversal on GPUs through
// compressed wide BVHs"):
//   * a node stands for up to 8 children; a ray visits ~12 such nodes where it visited ~22 four-wide ones — the traversal is a chain of
//     DEPENDENT steps (address -> loads -> slab tests -> next address) and the kernels are latency-bound, so fewer, fatter steps win;
//   * children sit in OCTANT slots: slot s holds a child lying towards corner s of the node (bit a of s set = the + side of axis a), so the
//     order in which a ray wants its hit children is `slot ^ octant(ray)` — no distances kept, no sorting network;
//   * the interior children of a node are stored CONSECUTIVELY (child k of the node = child_base + number of interior slots below it), so
//     the whole set of hit children is ONE stack entry (child_base, hit mask, interior mask) instead of up to three node references;
//   * the triangles (TLAS: instances) of a node's leaf children are consecutive too: leaf slot = 5-bit offset from tri_base + count (1..4).
// Child boxes are quantised to 8 bits per plane relative to the node's origin with a power-of-two scale per axis (floor / ceil: the
// decoded box encloses the f32 box, which is already padded for the Moeller-Trumbore rounding — DESIGN.md §2), so results do not depend
// on the tree and stay bit-identical to the oracle's.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "device_types.h"

#define RFW_HD_INLINE inline __host__ __device__

namespace rfwhip {

struct Node8 {               // 80 B = 5 dwordx4 per lane
    float ox, oy, oz;        // w0.xyz  origin (lower corner of all children)
    uint32_t exps_imask;     // w0.w    biased exponent bytes of the scales of x, y, z | interior mask << 24 (bit s: slot s is an interior node)
    uint32_t child_base;     // w1.x    first interior child, relative to the tree's first node
    uint32_t tri_base;       // w1.y    first primitive of the node's leaf children, relative to the tree's first primitive
    uint32_t meta[2];        // w1.zw   byte s: leaf slot -> count << 5 | offset (count 1..4 primitives at tri_base + offset); 0 = interior or empty
    uint32_t qlo[3][2];      // w2.xyzw = x, y; w3.xy = z: byte s of the axis' two words = lower plane of slot s
    uint32_t qhi[3][2];      // w3.zw = x; w4.xyzw = y, z
};
static_assert(sizeof(Node8) == 80, "Node8 is five dwordx4");
constexpr int kMaxLeaf8 = 4;         // primitives per leaf slot (a fatter leaf of the 4-wide tree takes several slots)
constexpr uint32_t kNode8Words = 20; // dwords per node

// one child of a wide node while it is being put together
struct Slot8 {
    float lo[3], hi[3];
    uint32_t ref; // interior: node index in the 4-wide tree; leaf: kLeafBit | (count - 1) << 27 | first, count <= kMaxLeaf8
};

RFW_HD_INLINE float slot_half_area(const Slot8& s)
{
    const float ex = s.hi[0] - s.lo[0], ey = s.hi[1] - s.lo[1], ez = s.hi[2] - s.lo[2];
    return ex * ey + ey * ez + ez * ex;
}

// Primitives below 4-wide node `node` when they are at most `limit` (else limit + 1), and the first of them: every builder hands a subtree a
// CONTIGUOUS range of the primitive order, so (first, count) names the whole subtree.  Bounded work: such a subtree has at most `limit` leaves.
RFW_HD_INLINE uint32_t small_subtree(const Node4* tree, uint32_t node, uint32_t limit, uint32_t& first)
{
    uint32_t stack[8];
    int sp = 0;
    uint32_t total = 0;
    first = 0xffffffffu;
    stack[sp++] = node;
    while (sp > 0) {
        const Node4& n = tree[stack[--sp]];
        for (int i = 0; i < 4; i++) {
            const uint32_t c = n.child[i];
            if (c == kInvalidRef) continue;
            if (c & kLeafBit) {
                const uint32_t f = c & kLeafFirstMask;
                total += ((c >> 27) & 15u) + 1u;
                first = f < first ? f : first;
            } else {
                total += 1u; // an interior child holds at least one primitive: lets the walk stop early
                if (total > limit || sp >= 8) return limit + 1u;
                total -= 1u;
                stack[sp++] = c;
            }
            if (total > limit) return limit + 1u;
        }
    }
    return total;
}

// children of 4-wide node `n` as slots (a leaf of more than kMaxLeaf8 primitives is cut into several slots with the leaf's box; an interior
// child whose whole subtree holds at most `merge` primitives becomes ONE leaf slot: a triangle test is cheaper than a dependent node step);
// returns how many, or -1 when they would not fit into `room`
RFW_HD_INLINE int node4_slots(const Node4* tree, const Node4& n, Slot8* out, int room, uint32_t merge)
{
    int k = 0;
    for (int i = 0; i < 4; i++) {
        uint32_t c = n.child[i];
        if (c == kInvalidRef) continue;
        Slot8 s;
        s.lo[0] = n.lox[i]; s.lo[1] = n.loy[i]; s.lo[2] = n.loz[i];
        s.hi[0] = n.hix[i]; s.hi[1] = n.hiy[i]; s.hi[2] = n.hiz[i];
        if (!(c & kLeafBit) && merge > 1u) {
            uint32_t first;
            const uint32_t total = small_subtree(tree, c, merge, first);
            if (total >= 1u && total <= merge) c = make_leaf(first, total);
        }
        if (c & kLeafBit) {
            uint32_t first = c & kLeafFirstMask, count = ((c >> 27) & 15u) + 1u;
            while (count > 0u) {
                const uint32_t take = count > (uint32_t)kMaxLeaf8 ? (uint32_t)kMaxLeaf8 : count;
                if (k >= room) return -1;
                s.ref = make_leaf(first, take);
                out[k++] = s;
                first += take;
                count -= take;
            }
        } else {
            if (k >= room) return -1;
            s.ref = c;
            out[k++] = s;
        }
    }
    return k;
}

// The up-to-8 children of the wide node rooted at 4-wide node `root`: the root's children, then repeatedly the interior child with the
// largest surface area replaced by ITS children while they fit (largest first: the child a ray is most likely to enter anyway).
RFW_HD_INLINE int gather_wide8(const Node4* tree, uint32_t root, Slot8* slots, uint32_t merge)
{
    int ns = node4_slots(tree, tree[root], slots, 8, merge);
    if (ns < 0) ns = 0; // cannot happen: 4 leaves of <= 8 primitives are 8 slots
    bool closed[8] = {false, false, false, false, false, false, false, false}; // interior children whose own children did not fit
    for (;;) {
        int best = -1;
        float best_area = -1.0f;
        for (int i = 0; i < ns; i++) {
            if ((slots[i].ref & kLeafBit) || closed[i]) continue;
            const float a = slot_half_area(slots[i]);
            if (a > best_area) { best_area = a; best = i; }
        }
        if (best < 0) break;
        Slot8 kids[8];
        const int nk = node4_slots(tree, tree[slots[best].ref], kids, 8 - (ns - 1), merge);
        if (nk < 0) { closed[best] = true; continue; }
        if (nk == 0) { // an interior node without children (not produced by the builders): drop the slot
            slots[best] = slots[ns - 1]; closed[best] = closed[ns - 1]; ns--;
            continue;
        }
        slots[best] = kids[0];
        closed[best] = false;
        for (int k = 1; k < nk; k++) { slots[ns] = kids[k]; closed[ns] = false; ns++; }
    }
    return ns;
}

// Octant slots: position p (0..7) gets the child that lies most towards corner p of the node (bit a of p: the + side of axis a), greedily
// by decreasing score = sum over the axes of +-(child centre - node centre).  pos_of[i] = position of gathered child i.
RFW_HD_INLINE void assign_octants(const Slot8* slots, int ns, int* pos_of)
{
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int i = 0; i < ns; i++)
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], slots[i].lo[a]); hi[a] = fmaxf(hi[a], slots[i].hi[a]); }
    float d[8][3];
    for (int i = 0; i < ns; i++)
        for (int a = 0; a < 3; a++) d[i][a] = 0.5f * (slots[i].lo[a] + slots[i].hi[a]) - 0.5f * (lo[a] + hi[a]);
    bool child_done[8] = {false, false, false, false, false, false, false, false}, pos_used[8] = {false, false, false, false, false, false, false, false};
    for (int round = 0; round < ns; round++) {
        int bi = -1, bp = -1;
        float best = -INFINITY;
        for (int i = 0; i < ns; i++) {
            if (child_done[i]) continue;
            for (int p = 0; p < 8; p++) {
                if (pos_used[p]) continue;
                const float sc = ((p & 1) ? d[i][0] : -d[i][0]) + ((p & 2) ? d[i][1] : -d[i][1]) + ((p & 4) ? d[i][2] : -d[i][2]);
                if (sc > best || bi < 0) { best = sc; bi = i; bp = p; }
            }
        }
        child_done[bi] = true;
        pos_used[bp] = true;
        pos_of[bi] = bp;
    }
}

// smallest power of two s (as a biased exponent byte, 1..254) with 255 * s >= extent
RFW_HD_INLINE uint32_t scale_exponent(float extent)
{
    const float want = extent * (1.0f / 255.0f);
    uint32_t e = (rfw_f2bits(want) >> 23) & 0xffu;
    if ((rfw_f2bits(want) & 0x007fffffu) != 0u) e += 1u;
    if (e < 1u) e = 1u;
    if (e > 254u) e = 254u;
    if (255.0f * rfw_bits2f(e << 23) < extent && e < 254u) e += 1u;
    return e;
}

// Encodes the node: by_pos[p] = child at octant position p (nullptr = empty), child_base / tri_base as allocated by the caller.  Leaf
// positions get their offsets in position order; `leaf_offset[p]` returns them (the caller copies the primitives there).
RFW_HD_INLINE Node8 encode_node8(const Slot8* const* by_pos, uint32_t child_base, uint32_t tri_base, uint32_t* leaf_offset)
{
    Node8 n;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int p = 0; p < 8; p++) {
        if (!by_pos[p]) continue;
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], by_pos[p]->lo[a]); hi[a] = fmaxf(hi[a], by_pos[p]->hi[a]); }
    }
    uint32_t e[3];
    float scale[3];
    for (int a = 0; a < 3; a++) {
        if (!(hi[a] >= lo[a])) { lo[a] = 0.0f; hi[a] = 0.0f; } // node without children
        e[a] = scale_exponent(hi[a] - lo[a]);
        scale[a] = rfw_bits2f(e[a] << 23);
    }
    n.ox = lo[0]; n.oy = lo[1]; n.oz = lo[2];
    uint32_t imask = 0u, off = 0u;
    n.meta[0] = n.meta[1] = 0u;
    for (int a = 0; a < 3; a++) { n.qlo[a][0] = n.qlo[a][1] = 0u; n.qhi[a][0] = n.qhi[a][1] = 0u; }
    for (int p = 0; p < 8; p++) {
        const int w = p >> 2, sh = 8 * (p & 3);
        leaf_offset[p] = 0u;
        if (!by_pos[p]) { // empty: inverted box, never hit
            for (int a = 0; a < 3; a++) n.qlo[a][w] |= 255u << sh;
            continue;
        }
        const Slot8& s = *by_pos[p];
        if (s.ref & kLeafBit) {
            const uint32_t count = ((s.ref >> 27) & 15u) + 1u;
            n.meta[w] |= ((count << 5) | off) << sh;
            leaf_offset[p] = off;
            off += count;
        } else {
            imask |= 1u << p;
        }
        for (int a = 0; a < 3; a++) {
            const float inv = 1.0f / scale[a];
            float fl = floorf((s.lo[a] - lo[a]) * inv), fh = ceilf((s.hi[a] - lo[a]) * inv);
            fl = fl < 0.0f ? 0.0f : (fl > 255.0f ? 255.0f : fl);
            fh = fh < 0.0f ? 0.0f : (fh > 255.0f ? 255.0f : fh);
            uint32_t ql = (uint32_t)fl, qh = (uint32_t)fh;
            while (ql > 0u && lo[a] + (float)ql * scale[a] > s.lo[a]) ql--; // decoded planes must enclose the f32 box
            while (qh < 255u && lo[a] + (float)qh * scale[a] < s.hi[a]) qh++;
            n.qlo[a][w] |= ql << sh;
            n.qhi[a][w] |= qh << sh;
        }
    }
    n.exps_imask = e[0] | (e[1] << 8) | (e[2] << 16) | (imask << 24);
    n.child_base = child_base;
    n.tri_base = tri_base;
    return n;
}

// decoded box of slot p (tests, validators, the CPU model of the traversal)
RFW_HD_INLINE void decode_slot8(const Node8& n, int p, float* lo, float* hi)
{
    const float o[3] = {n.ox, n.oy, n.oz};
    const int w = p >> 2, sh = 8 * (p & 3);
    for (int a = 0; a < 3; a++) {
        const float s = rfw_bits2f(((n.exps_imask >> (8 * a)) & 0xffu) << 23);
        lo[a] = o[a] + (float)((n.qlo[a][w] >> sh) & 0xffu) * s;
        hi[a] = o[a] + (float)((n.qhi[a][w] >> sh) & 0xffu) * s;
    }
}

// ---------------------------------------------------------------- host collapse (HOST_SAH builder, host TLAS, tests, probes)
struct HostBvh8 {
    std::vector<Node8> nodes;         // node 0 = root
    std::vector<uint32_t> prim_order; // primitive ids in the order the nodes' leaf slots address them
};

// tree4 / order4: a 4-wide tree as the builders emit it (node 0 = root, leaf refs index into order4)
inline void collapse_bvh8_host(const Node4* tree4, const uint32_t* order4, uint32_t n_prims, HostBvh8& out, uint32_t merge = kMaxLeaf8)
{
    out.nodes.clear();
    out.prim_order.clear();
    out.prim_order.reserve(n_prims);
    struct Job { uint32_t n4, n8; };
    std::vector<Job> queue;
    out.nodes.emplace_back();
    queue.push_back({0u, 0u});
    for (size_t q = 0; q < queue.size(); q++) {
        const Job j = queue[q];
        Slot8 slots[8];
        const int ns = gather_wide8(tree4, j.n4, slots, merge);
        int pos_of[8];
        assign_octants(slots, ns, pos_of);
        const Slot8* by_pos[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        for (int i = 0; i < ns; i++) by_pos[pos_of[i]] = &slots[i];
        const uint32_t child_base = (uint32_t)out.nodes.size(), tri_base = (uint32_t)out.prim_order.size();
        uint32_t leaf_offset[8];
        const Node8 n = encode_node8(by_pos, child_base, tri_base, leaf_offset);
        for (int p = 0; p < 8; p++) {
            if (!by_pos[p]) continue;
            const uint32_t ref = by_pos[p]->ref;
            if (ref & kLeafBit) {
                const uint32_t first = ref & kLeafFirstMask, count = ((ref >> 27) & 15u) + 1u;
                for (uint32_t k = 0; k < count; k++) out.prim_order.push_back(order4[first + k]);
            } else {
                queue.push_back({ref, (uint32_t)out.nodes.size()});
                out.nodes.emplace_back();
            }
        }
        out.nodes[j.n8] = n;
    }
}

} // namespace rfwhip
