/*
 * oracle/bvh.h — TEST INFRASTRUCTURE (CPU oracle).  Not part of the product.
 *
 * BVH construction as the reference performs it on the CPU through the crate
 * `rtbvh` ("0.6", crates/rfw-backend/Cargo.toml:17; source NOT vendored under
 * /root/reference and absent from this machine — SURVEY.md §8c):
 *   - `BinnedSahBuilder::new(aabbs, centers).build()`  (backends/gpu-rt/src/lib.rs:1576-1580)
 *   - `MBVH::construct(&bvh)`                           (backends/gpu-rt/src/lib.rs:1581)
 * Because the crate's source is unavailable, this file restates the PUBLISHED
 * algorithms those calls name — binned SAH (Wald 2007, "On fast Construction of
 * SAH-based Bounding Volume Hierarchies") and the BVH2 -> 4-wide collapse
 * (Wald/Benthin/Boulos 2008; Dammertz 2008) — and produces exactly the data
 * layout the reference's kernels consume:
 *   BVHNode  32 B  backends/gpu-rt/shaders/structs.glsl:45-54
 *   MBVHNode 128 B backends/gpu-rt/shaders/structs.glsl:56-65
 * with the leaf/interior conventions of ray_gen.comp:160-175, 213-246
 * (SURVEY.md Appendix C).  Closest-hit / any-hit results do not depend on the
 * tree topology (see the tie rule in oracle.cpp), so builder details such as
 * the bin count are free parameters, not parity-relevant constants.
 */
#ifndef ORACLE_BVH_H
#define ORACLE_BVH_H

#include <algorithm>
#include <cstdint>
#include <vector>

namespace orc {

struct Box {
    float mn[3], mx[3];
    void reset() { for (int i = 0; i < 3; i++) { mn[i] = 1e34f; mx[i] = -1e34f; } }
    void grow(const float* p) { for (int i = 0; i < 3; i++) { if (p[i] < mn[i]) mn[i] = p[i]; if (p[i] > mx[i]) mx[i] = p[i]; } }
    void grow(const Box& b) { for (int i = 0; i < 3; i++) { if (b.mn[i] < mn[i]) mn[i] = b.mn[i]; if (b.mx[i] > mx[i]) mx[i] = b.mx[i]; } }
    float half_area() const
    {
        float ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
        if (ex < 0.0f || ey < 0.0f || ez < 0.0f) return 0.0f;
        return ex * ey + ey * ez + ez * ex;
    }
};

// structs.glsl:45-54
struct BVHNode {
    float bmin_x, bmin_y, bmin_z;
    float bmax_x, bmax_y, bmax_z;
    int32_t left_first;
    int32_t count; // >= 0: leaf over prim_indices[left_first .. left_first+count); < 0: children left_first, left_first+1
};

// structs.glsl:56-65
struct MBVHNode {
    float min_x[4], max_x[4], min_y[4], max_y[4], min_z[4], max_z[4];
    int32_t children[4]; // < 0: empty slot; leaf: first prim index; interior: node index
    int32_t counts[4];   // >= 0: leaf prim count; < 0: interior
};

struct BVH {
    std::vector<BVHNode> nodes;
    std::vector<uint32_t> prim_indices;
};

struct MBVH {
    std::vector<MBVHNode> nodes;
};

static const int kBins = 16;

// Binned-SAH top-down build over primitive boxes + centroids.
inline void build_binned_sah(const std::vector<Box>& boxes, const std::vector<float>& centers /* 3 per prim */, BVH& out)
{
    const uint32_t n = (uint32_t)boxes.size();
    out.nodes.clear();
    out.prim_indices.resize(n);
    for (uint32_t i = 0; i < n; i++) out.prim_indices[i] = i;
    out.nodes.reserve(n ? 2 * n : 1);
    BVHNode root{};
    Box rb; rb.reset();
    for (uint32_t i = 0; i < n; i++) rb.grow(boxes[i]);
    root.bmin_x = rb.mn[0]; root.bmin_y = rb.mn[1]; root.bmin_z = rb.mn[2];
    root.bmax_x = rb.mx[0]; root.bmax_y = rb.mx[1]; root.bmax_z = rb.mx[2];
    root.left_first = 0; root.count = (int32_t)n;
    out.nodes.push_back(root);
    if (n <= 1) return;

    std::vector<uint32_t> stack;
    stack.push_back(0);
    while (!stack.empty()) {
        const uint32_t ni = stack.back();
        stack.pop_back();
        const int32_t first = out.nodes[ni].left_first;
        const int32_t count = out.nodes[ni].count;
        if (count <= 1) continue;

        Box nb;
        nb.mn[0] = out.nodes[ni].bmin_x; nb.mn[1] = out.nodes[ni].bmin_y; nb.mn[2] = out.nodes[ni].bmin_z;
        nb.mx[0] = out.nodes[ni].bmax_x; nb.mx[1] = out.nodes[ni].bmax_y; nb.mx[2] = out.nodes[ni].bmax_z;
        Box cb; cb.reset();
        for (int32_t i = 0; i < count; i++) cb.grow(&centers[3 * out.prim_indices[first + i]]);

        float best_cost = 1e34f;
        int best_axis = -1, best_plane = -1;
        for (int a = 0; a < 3; a++) {
            const float lo = cb.mn[a], hi = cb.mx[a];
            if (!(hi > lo)) continue;
            const float scale = (float)kBins / (hi - lo);
            Box bin_box[kBins];
            uint32_t bin_cnt[kBins];
            for (int b = 0; b < kBins; b++) { bin_box[b].reset(); bin_cnt[b] = 0; }
            for (int32_t i = 0; i < count; i++) {
                const uint32_t p = out.prim_indices[first + i];
                int b = (int)((centers[3 * p + a] - lo) * scale);
                if (b >= kBins) b = kBins - 1;
                if (b < 0) b = 0;
                bin_cnt[b]++;
                bin_box[b].grow(boxes[p]);
            }
            float left_area[kBins - 1], right_area[kBins - 1];
            uint32_t left_cnt[kBins - 1], right_cnt[kBins - 1];
            Box lb, rbx; lb.reset(); rbx.reset();
            uint32_t lc = 0, rc = 0;
            for (int b = 0; b < kBins - 1; b++) {
                lc += bin_cnt[b]; lb.grow(bin_box[b]);
                left_cnt[b] = lc; left_area[b] = lb.half_area();
                rc += bin_cnt[kBins - 1 - b]; rbx.grow(bin_box[kBins - 1 - b]);
                right_cnt[kBins - 2 - b] = rc; right_area[kBins - 2 - b] = rbx.half_area();
            }
            for (int b = 0; b < kBins - 1; b++) {
                if (left_cnt[b] == 0 || right_cnt[b] == 0) continue;
                const float cost = (float)left_cnt[b] * left_area[b] + (float)right_cnt[b] * right_area[b];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = b; }
            }
        }

        const float leaf_cost = (float)count * nb.half_area();
        int32_t mid = -1;
        if (best_axis >= 0 && (best_cost < leaf_cost || count > 8)) {
            const float lo = cb.mn[best_axis], hi = cb.mx[best_axis];
            const float scale = (float)kBins / (hi - lo);
            uint32_t* begin = &out.prim_indices[first];
            uint32_t* m = std::partition(begin, begin + count, [&](uint32_t p) {
                int b = (int)((centers[3 * p + best_axis] - lo) * scale);
                if (b >= kBins) b = kBins - 1;
                if (b < 0) b = 0;
                return b <= best_plane;
            });
            mid = (int32_t)(m - begin);
        } else if (best_axis < 0 && count > 8) {
            mid = count / 2; // all centroids coincide: split the list arbitrarily to bound leaf size
        }
        if (mid <= 0 || mid >= count) continue; // stays a leaf

        const uint32_t li = (uint32_t)out.nodes.size();
        BVHNode l{}, r{};
        Box lbx, rbx2; lbx.reset(); rbx2.reset();
        for (int32_t i = 0; i < mid; i++) lbx.grow(boxes[out.prim_indices[first + i]]);
        for (int32_t i = mid; i < count; i++) rbx2.grow(boxes[out.prim_indices[first + i]]);
        l.bmin_x = lbx.mn[0]; l.bmin_y = lbx.mn[1]; l.bmin_z = lbx.mn[2];
        l.bmax_x = lbx.mx[0]; l.bmax_y = lbx.mx[1]; l.bmax_z = lbx.mx[2];
        l.left_first = first; l.count = mid;
        r.bmin_x = rbx2.mn[0]; r.bmin_y = rbx2.mn[1]; r.bmin_z = rbx2.mn[2];
        r.bmax_x = rbx2.mx[0]; r.bmax_y = rbx2.mx[1]; r.bmax_z = rbx2.mx[2];
        r.left_first = first + mid; r.count = count - mid;
        out.nodes.push_back(l);
        out.nodes.push_back(r);
        out.nodes[ni].left_first = (int32_t)li;
        out.nodes[ni].count = -1;
        stack.push_back(li + 1);
        stack.push_back(li);
    }
}

// BVH2 -> 4-wide collapse: every MBVH node adopts grandchildren (largest surface area first)
// until it has 4 children or only leaves remain.
inline void collapse_mbvh(const BVH& bvh, MBVH& out)
{
    out.nodes.clear();
    if (bvh.nodes.empty()) return;
    struct Work { uint32_t mnode; uint32_t bnode; };
    std::vector<Work> stack;

    auto empty_node = []() {
        MBVHNode m;
        for (int i = 0; i < 4; i++) {
            m.min_x[i] = m.min_y[i] = m.min_z[i] = 1e34f;
            m.max_x[i] = m.max_y[i] = m.max_z[i] = -1e34f;
            m.children[i] = -1;
            m.counts[i] = -1;
        }
        return m;
    };
    auto set_child_box = [](MBVHNode& m, int slot, const BVHNode& b) {
        m.min_x[slot] = b.bmin_x; m.min_y[slot] = b.bmin_y; m.min_z[slot] = b.bmin_z;
        m.max_x[slot] = b.bmax_x; m.max_y[slot] = b.bmax_y; m.max_z[slot] = b.bmax_z;
    };

    out.nodes.push_back(empty_node());
    const BVHNode& root = bvh.nodes[0];
    if (root.count >= 0) { // single-leaf tree: root MBVH node with one leaf child
        set_child_box(out.nodes[0], 0, root);
        out.nodes[0].children[0] = root.left_first;
        out.nodes[0].counts[0] = root.count;
        return;
    }
    stack.push_back({0u, 0u});
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        const BVHNode& b = bvh.nodes[w.bnode];
        uint32_t kids[4];
        int nk = 2;
        kids[0] = (uint32_t)b.left_first;
        kids[1] = (uint32_t)b.left_first + 1;
        while (nk < 4) {
            int best = -1;
            float best_area = -1.0f;
            for (int i = 0; i < nk; i++) {
                const BVHNode& k = bvh.nodes[kids[i]];
                if (k.count >= 0) continue;
                Box kb;
                kb.mn[0] = k.bmin_x; kb.mn[1] = k.bmin_y; kb.mn[2] = k.bmin_z;
                kb.mx[0] = k.bmax_x; kb.mx[1] = k.bmax_y; kb.mx[2] = k.bmax_z;
                const float a = kb.half_area();
                if (a > best_area) { best_area = a; best = i; }
            }
            if (best < 0) break;
            const BVHNode& k = bvh.nodes[kids[best]];
            kids[best] = (uint32_t)k.left_first;
            kids[nk++] = (uint32_t)k.left_first + 1;
        }
        for (int i = 0; i < nk; i++) {
            const BVHNode& k = bvh.nodes[kids[i]];
            set_child_box(out.nodes[w.mnode], i, k);
            if (k.count >= 0) {
                out.nodes[w.mnode].children[i] = k.left_first;
                out.nodes[w.mnode].counts[i] = k.count;
            } else {
                const uint32_t mi = (uint32_t)out.nodes.size();
                out.nodes.push_back(empty_node());
                out.nodes[w.mnode].children[i] = (int32_t)mi;
                out.nodes[w.mnode].counts[i] = -1;
                stack.push_back({mi, kids[i]});
            }
        }
    }
}

} // namespace orc
#endif
