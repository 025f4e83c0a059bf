// sah_reference.h — TEST INFRASTRUCTURE: the binned SAH of rfw-rs_amd/csrc/sah_build.hip as a plain serial recursion, written from the kernels'
// header comments: 16 bins per axis over the bounds of the range's centroids, the 45 planes priced as count x half area on both sides, the
// first cheapest in (axis, plane) order, a leaf when splitting does not pay and the range fits a leaf, halves when every centroid coincides
// and the range does not fit, stable partition.  Used by k_small_emu.cpp (node for node, same primitive order) and sah_build_emu.cpp (the
// whole device build: same tree, the primitives of a leaf as a set — the level kernels of phase 1 partition by atomics, not stably).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace sahref {
struct Box { float lo[4], hi[4]; };
struct RefNode { uint32_t first, count; float lo[3], hi[3]; int left; };
inline std::vector<RefNode> g_ref;
inline std::vector<uint32_t> g_order; // position -> primitive
inline const Box* g_boxes;
inline int g_max_leaf;
inline float g_trav;

inline float ha(const float* lo, const float* hi)
{
    const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
    if (!(ex >= 0.0f) || !(ey >= 0.0f) || !(ez >= 0.0f)) return 0.0f;
    return ex * ey + ey * ez + ez * ex;
}
inline int ref_bin(float c, float lo, float hi)
{
    if (!(hi > lo)) return 0;
    int b = (int)((c - lo) * (16.0f / (hi - lo)));
    return b < 0 ? 0 : (b > 15 ? 15 : b);
}
struct RBin { uint32_t n = 0; float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}; };
inline void grow(RBin& b, const RBin& o)
{
    b.n += o.n;
    for (int c = 0; c < 3; c++) { b.lo[c] = std::min(b.lo[c], o.lo[c]); b.hi[c] = std::max(b.hi[c], o.hi[c]); }
}

inline void ref_build(int node)
{
    const uint32_t first = g_ref[node].first, count = g_ref[node].count;
    if (count <= 1) return;
    float nlo[3], nhi[3];
    for (int a = 0; a < 3; a++) { nlo[a] = g_ref[node].lo[a]; nhi[a] = g_ref[node].hi[a]; }
    std::vector<float> cen(3 * count);
    float clo[3] = {INFINITY, INFINITY, INFINITY}, chi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (uint32_t i = 0; i < count; i++) {
        const Box& b = g_boxes[g_order[first + i]];
        for (int a = 0; a < 3; a++) { const float c = 0.5f * (b.lo[a] + b.hi[a]); cen[3 * i + a] = c; clo[a] = std::min(clo[a], c); chi[a] = std::max(chi[a], c); }
    }
    RBin bins[3][16];
    std::vector<int> bin(3 * count);
    for (uint32_t i = 0; i < count; i++) {
        const Box& b = g_boxes[g_order[first + i]];
        RBin one; one.n = 1;
        for (int c = 0; c < 3; c++) { one.lo[c] = b.lo[c]; one.hi[c] = b.hi[c]; }
        for (int a = 0; a < 3; a++) { bin[3 * i + a] = ref_bin(cen[3 * i + a], clo[a], chi[a]); grow(bins[a][bin[3 * i + a]], one); }
    }
    float best = INFINITY; int axis = -1, plane = -1; RBin bl, br;
    for (int a = 0; a < 3; a++)
        for (int p = 0; p < 15; p++) {
            RBin l, r;
            for (int k = 0; k <= p; k++) grow(l, bins[a][k]);
            for (int k = p + 1; k < 16; k++) grow(r, bins[a][k]);
            if (!l.n || !r.n) continue;
            const float cost = (float)l.n * ha(l.lo, l.hi) + (float)r.n * ha(r.lo, r.hi);
            if (axis < 0 || cost < best) { best = cost; axis = a; plane = p; bl = l; br = r; }
        }
    const float area = ha(nlo, nhi), leaf_cost = (float)count * area;
    bool split = false, halves = false;
    if (axis >= 0 && (best + g_trav * area < leaf_cost || (int)count > g_max_leaf)) split = true;
    else if ((int)count > g_max_leaf) { split = true; halves = true; }
    if (!split) return;
    uint32_t lc;
    RefNode l{}, r{};
    if (halves) {
        lc = count / 2;
        for (int a = 0; a < 3; a++) { l.lo[a] = r.lo[a] = nlo[a]; l.hi[a] = r.hi[a] = nhi[a]; }
        if (count == 2)
            for (int a = 0; a < 3; a++) {
                l.lo[a] = g_boxes[g_order[first]].lo[a]; l.hi[a] = g_boxes[g_order[first]].hi[a];
                r.lo[a] = g_boxes[g_order[first + 1]].lo[a]; r.hi[a] = g_boxes[g_order[first + 1]].hi[a];
            }
    } else {
        lc = bl.n;
        for (int a = 0; a < 3; a++) { l.lo[a] = bl.lo[a]; l.hi[a] = bl.hi[a]; r.lo[a] = br.lo[a]; r.hi[a] = br.hi[a]; }
        std::vector<uint32_t> left, right;
        for (uint32_t i = 0; i < count; i++) (bin[3 * i + axis] <= plane ? left : right).push_back(g_order[first + i]);
        std::copy(left.begin(), left.end(), g_order.begin() + first);
        std::copy(right.begin(), right.end(), g_order.begin() + first + left.size());
    }
    l.first = first; l.count = lc; l.left = -1;
    r.first = first + lc; r.count = count - lc; r.left = -1;
    const int li = (int)g_ref.size();
    g_ref.push_back(l); g_ref.push_back(r);
    g_ref[node].left = li;
    ref_build(li);
    ref_build(li + 1);
}
} // namespace sahref
