// bvh_host.cpp — binned-SAH BVH2 (parallel over subtrees) collapsed into 4-wide nodes in device layout.
#include "bvh_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <future>

namespace rfwhip {
namespace {

constexpr int kBinCount = 16;

struct B3 {
    float lo[3], hi[3];
    void reset()
    {
        for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    }
    void grow(const float* l, const float* h)
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l[a]); hi[a] = std::max(hi[a], h[a]); }
    }
    void grow_pt(const float* p)
    {
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); }
    }
    float half_area() const
    {
        const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
        if (!(ex >= 0.0f) || !(ey >= 0.0f) || !(ez >= 0.0f)) return 0.0f;
        return ex * ey + ey * ez + ez * ex;
    }
};

struct Node2 {
    B3 box;
    uint32_t left = 0, right = 0; // children (interior)
    uint32_t first = 0, count = 0; // leaf range in prim order (count > 0 => leaf)
};

struct Builder2 {
    const std::vector<PrimBox>& boxes;
    std::vector<float> cent; // 3 per prim
    std::vector<uint32_t> order;
    std::vector<Node2> nodes;
    std::atomic<uint32_t> next_node{1};
    int max_leaf;
    int max_par_depth;
    float trav_cost = 1.0f;

    Builder2(const std::vector<PrimBox>& b, int ml, int threads) : boxes(b), max_leaf(ml)
    {
        const size_t n = b.size();
        cent.resize(3 * n);
        order.resize(n);
        for (size_t i = 0; i < n; i++) {
            order[i] = (uint32_t)i;
            for (int a = 0; a < 3; a++) cent[3 * i + a] = 0.5f * (b[i].lo[a] + b[i].hi[a]);
        }
        nodes.resize(n ? 2 * n : 1);
        max_par_depth = 0;
        while ((1 << max_par_depth) < threads * 4) max_par_depth++;
    }

    void build(uint32_t ni, uint32_t first, uint32_t count, int depth)
    {
        Node2& node = nodes[ni];
        node.box.reset();
        B3 cb;
        cb.reset();
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t p = order[first + i];
            node.box.grow(boxes[p].lo, boxes[p].hi);
            cb.grow_pt(&cent[3 * p]);
        }
        node.first = first;
        node.count = count;
        if (count <= 1) return;

        float best_cost = INFINITY;
        int best_axis = -1, best_plane = -1;
        for (int a = 0; a < 3; a++) {
            const float lo = cb.lo[a], hi = cb.hi[a];
            if (!(hi > lo)) continue;
            const float scale = (float)kBinCount / (hi - lo);
            B3 bin_box[kBinCount];
            uint32_t bin_cnt[kBinCount];
            for (int b = 0; b < kBinCount; b++) { bin_box[b].reset(); bin_cnt[b] = 0; }
            for (uint32_t i = 0; i < count; i++) {
                const uint32_t p = order[first + i];
                int b = (int)((cent[3 * p + a] - lo) * scale);
                b = std::min(std::max(b, 0), kBinCount - 1);
                bin_cnt[b]++;
                bin_box[b].grow(boxes[p].lo, boxes[p].hi);
            }
            float right_area[kBinCount];
            uint32_t right_cnt[kBinCount];
            B3 acc;
            acc.reset();
            uint32_t c = 0;
            for (int b = kBinCount - 1; b > 0; b--) {
                acc.grow(bin_box[b].lo, bin_box[b].hi);
                c += bin_cnt[b];
                right_area[b] = acc.half_area();
                right_cnt[b] = c;
            }
            acc.reset();
            c = 0;
            for (int b = 0; b < kBinCount - 1; b++) {
                acc.grow(bin_box[b].lo, bin_box[b].hi);
                c += bin_cnt[b];
                if (c == 0 || right_cnt[b + 1] == 0) continue;
                const float cost = (float)c * acc.half_area() + (float)right_cnt[b + 1] * right_area[b + 1];
                if (cost < best_cost) { best_cost = cost; best_axis = a; best_plane = b; }
            }
        }
        uint32_t mid = 0;
        const float leaf_cost = (float)count * node.box.half_area();
        // traversal step ~ 1 triangle test: split when SAH says so, or when the leaf would be too fat
        if (best_axis >= 0 && (best_cost + trav_cost * node.box.half_area() < leaf_cost || (int)count > max_leaf)) {
            const float lo = cb.lo[best_axis], hi = cb.hi[best_axis];
            const float scale = (float)kBinCount / (hi - lo);
            uint32_t* begin = &order[first];
            uint32_t* m = std::partition(begin, begin + count, [&](uint32_t p) {
                int b = (int)((cent[3 * p + best_axis] - lo) * scale);
                b = std::min(std::max(b, 0), kBinCount - 1);
                return b <= best_plane;
            });
            mid = (uint32_t)(m - begin);
        } else if ((int)count > max_leaf) {
            mid = count / 2; // coincident centroids: arbitrary halves keep leaves bounded
        }
        if (mid == 0 || mid >= count) return;
        const uint32_t li = next_node.fetch_add(2);
        node.left = li;
        node.right = li + 1;
        node.count = 0;
        if (depth < max_par_depth && count > 8192) {
            auto fut = std::async(std::launch::async, [=]() { build(li, first, mid, depth + 1); });
            build(li + 1, first + mid, count - mid, depth + 1);
            fut.get();
        } else {
            build(li, first, mid, depth + 1);
            build(li + 1, first + mid, count - mid, depth + 1);
        }
    }
};

inline void set_child(Node4& n, int slot, const B3& b, uint32_t ref)
{
    n.lox[slot] = b.lo[0]; n.loy[slot] = b.lo[1]; n.loz[slot] = b.lo[2];
    n.hix[slot] = b.hi[0]; n.hiy[slot] = b.hi[1]; n.hiz[slot] = b.hi[2];
    n.child[slot] = ref;
}
inline Node4 empty_node4()
{
    Node4 n;
    for (int i = 0; i < 4; i++) {
        n.lox[i] = n.loy[i] = n.loz[i] = INFINITY;
        n.hix[i] = n.hiy[i] = n.hiz[i] = -INFINITY;
        n.child[i] = kInvalidRef;
        n.pad[i] = 0;
    }
    return n;
}

} // namespace

void build_bvh4_host(const std::vector<PrimBox>& boxes, int max_leaf, int threads, HostBvh4& out, float trav_cost)
{
    out.nodes.clear();
    out.prim_order.clear();
    out.nodes.push_back(empty_node4());
    const uint32_t n = (uint32_t)boxes.size();
    if (n == 0) return;
    max_leaf = std::min(std::max(max_leaf, 1), kMaxLeafTris);
    Builder2 b2(boxes, max_leaf, std::max(threads, 1));
    b2.trav_cost = trav_cost;
    b2.build(0, 0, n, 0);
    out.prim_order = b2.order;
    const std::vector<Node2>& n2 = b2.nodes;

    if (n2[0].count > 0) { // the whole tree is one leaf
        set_child(out.nodes[0], 0, n2[0].box, make_leaf(n2[0].first, n2[0].count));
        return;
    }
    struct Work { uint32_t n4, n2; };
    std::vector<Work> stack;
    stack.push_back({0u, 0u});
    while (!stack.empty()) {
        const Work w = stack.back();
        stack.pop_back();
        uint32_t kids[4] = {n2[w.n2].left, n2[w.n2].right, 0, 0};
        int nk = 2;
        while (nk < 4) { // adopt grandchildren, largest surface area first
            int best = -1;
            float best_area = -1.0f;
            for (int i = 0; i < nk; i++) {
                if (n2[kids[i]].count > 0) continue;
                const float a = n2[kids[i]].box.half_area();
                if (a > best_area) { best_area = a; best = i; }
            }
            if (best < 0) break;
            const uint32_t k = kids[best];
            kids[best] = n2[k].left;
            kids[nk++] = n2[k].right;
        }
        for (int i = 0; i < nk; i++) {
            const Node2& k = n2[kids[i]];
            if (k.count > 0) {
                set_child(out.nodes[w.n4], i, k.box, make_leaf(k.first, k.count));
            } else {
                const uint32_t id = (uint32_t)out.nodes.size();
                out.nodes.push_back(empty_node4());
                set_child(out.nodes[w.n4], i, k.box, id);
                stack.push_back({id, kids[i]});
            }
        }
    }
}

uint64_t validate_bvh4(const HostBvh4& bvh, const std::vector<PrimBox>& boxes)
{
    uint64_t errors = 0;
    std::vector<uint32_t> seen(boxes.size(), 0);
    if (bvh.nodes.empty()) return boxes.empty() ? 0 : 1;
    struct W { uint32_t node; float lo[3], hi[3]; };
    std::vector<W> st;
    st.push_back({0u, {-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}});
    while (!st.empty()) {
        const W w = st.back();
        st.pop_back();
        const Node4& n = bvh.nodes[w.node];
        for (int i = 0; i < 4; i++) {
            if (n.child[i] == kInvalidRef) continue;
            const float lo[3] = {n.lox[i], n.loy[i], n.loz[i]}, hi[3] = {n.hix[i], n.hiy[i], n.hiz[i]};
            for (int a = 0; a < 3; a++)
                if (lo[a] < w.lo[a] || hi[a] > w.hi[a]) errors++;
            if (n.child[i] & kLeafBit) {
                const uint32_t first = n.child[i] & kLeafFirstMask, count = ((n.child[i] >> 27) & 15u) + 1u;
                for (uint32_t k = 0; k < count; k++) {
                    if (first + k >= bvh.prim_order.size()) { errors++; continue; }
                    const uint32_t p = bvh.prim_order[first + k];
                    seen[p]++;
                    for (int a = 0; a < 3; a++)
                        if (boxes[p].lo[a] < lo[a] || boxes[p].hi[a] > hi[a]) errors++;
                }
            } else {
                W c{n.child[i], {lo[0], lo[1], lo[2]}, {hi[0], hi[1], hi[2]}};
                st.push_back(c);
            }
        }
    }
    for (uint32_t s : seen)
        if (s != 1) errors++;
    return errors;
}

} // namespace rfwhip
