// sah_build_emu.cpp — TEST INFRASTRUCTURE: the device SAH builder of rfw-rs_amd/csrc/sah_build.hip — the FILE, host side and kernels, compiled as
// C++ against tests/emu/fake_hip (launches run their workgroups one after the other under wave_emu.h) — builds a tree on the CPU, and the 4-wide
// tree it hands to the traversal (the reference: rtbvh's binned SAH + MBVH::construct, backends/gpu-rt/src/lib.rs:1345-1383,
// :1411) is compared with the one that follows from sah_reference.h's serial binned SAH: every even-depth node of the binary tree with its
// grandchildren as children, same boxes, the primitives of every leaf as a set.
//   usage: sah_build_emu <primitives> <seed> <kind> <max_leaf> <trav_cost> [trees]     trees > 1: the forest build over that many meshes;
//          trees = 0: one tree, then its REFIT (a16 / f3: launch_refit_setup + launch_refit) to triangles that have nothing to do with the boxes
//          it was built over — every child box must be the union of what lies below it, the leaves' from the padded triangle boxes
//   prints "OK nodes4=<n> leaves=<n> groups=<workgroups run> wave_collectives=<meetings of a wavefront's lanes, all kernels>" or the first difference
#include "sah_reference.h"

#include "sah_build.hip" // (found through -I rfw-rs_amd/csrc; its <hip/hip_runtime.h> and <hipcub/hipcub.hpp> are tests/emu/fake_hip's)

#include <cstdio>
#include <cstdlib>
#include <random>

namespace rfwhip {
const EnvSwitches& env_switches()
{
    static EnvSwitches e;
    return e;
}
} // namespace rfwhip

using namespace rfwhip;
using namespace sahref;

namespace {
int g_leaves = 0, g_nodes4 = 0;
const Node4* g_dev;      // the tree's region
const uint32_t* g_order; // leaf-ordered primitive ids, relative to the tree (a leaf's `first` indexes it)
uint32_t g_dev_count;

// the children a 4-wide node takes from binary node r: its grandchildren, or a child that is a leaf
int kids_of(int r, int* kids)
{
    int nk = 0;
    for (int c = sahref::g_ref[r].left; c < sahref::g_ref[r].left + 2; c++) {
        if (sahref::g_ref[c].left < 0) kids[nk++] = c;
        else { kids[nk++] = sahref::g_ref[c].left; kids[nk++] = sahref::g_ref[c].left + 1; }
    }
    return nk;
}
bool same_leaf(uint32_t ref4, const RefNode& R, uint32_t base)
{
    if (!(ref4 & 0x80000000u)) { std::printf("DIFF a leaf of %u primitives is an inner node on the device\n", R.count); return false; }
    const uint32_t first = ref4 & 0x07FFFFFFu, count = ((ref4 >> 27) & 15u) + 1u;
    if (count != R.count) { std::printf("DIFF leaf size: device %u reference %u\n", count, R.count); return false; }
    std::vector<uint32_t> a(g_order + first, g_order + first + count), b(sahref::g_order.begin() + R.first, sahref::g_order.begin() + R.first + R.count);
    for (auto& x : b) x -= base;
    std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
    if (a != b) { std::printf("DIFF leaf contents at device position %u\n", first); return false; }
    g_leaves++;
    return true;
}
bool compare4(uint32_t d, int r, uint32_t base, int depth)
{
    if (d >= g_dev_count) { std::printf("DIFF node %u outside the tree's %u nodes\n", d, g_dev_count); return false; }
    g_nodes4++;
    const Node4& D = g_dev[d];
    int kids[4];
    const int nk = kids_of(r, kids);
    for (int k = 0; k < 4; k++) {
        if (k >= nk) {
            if (D.child[k] != kInvalidRef) { std::printf("DIFF slot %d of node %u should be empty\n", k, d); return false; }
            continue;
        }
        const RefNode& R = sahref::g_ref[kids[k]];
        if (D.lox[k] != R.lo[0] || D.loy[k] != R.lo[1] || D.loz[k] != R.lo[2] || D.hix[k] != R.hi[0] || D.hiy[k] != R.hi[1] || D.hiz[k] != R.hi[2]) {
            std::printf("DIFF box of child %d of node %u (depth %d, %u primitives): device %g %g %g / %g %g %g reference %g %g %g / %g %g %g\n", k, d, depth, R.count, D.lox[k], D.loy[k],
                        D.loz[k], D.hix[k], D.hiy[k], D.hiz[k], R.lo[0], R.lo[1], R.lo[2], R.hi[0], R.hi[1], R.hi[2]);
            return false;
        }
        if (R.left < 0) { if (!same_leaf(D.child[k], R, base)) return false; }
        else {
            if (D.child[k] & 0x80000000u) { std::printf("DIFF child %d of node %u is a leaf on the device, %u primitives in the reference\n", k, d, R.count); return false; }
            if (!compare4(D.child[k], kids[k], base, depth + 2)) return false;
        }
    }
    return true;
}
// one tree over boxes[first, first + count): the reference, then the comparison with the device's region
bool check_tree(const std::vector<DevBox>& boxes, uint32_t first, uint32_t count, const Node4* dev, uint32_t dev_count, const uint32_t* order, int max_leaf, float trav)
{
    static_assert(sizeof(Box) == sizeof(DevBox), "");
    sahref::g_boxes = reinterpret_cast<const Box*>(boxes.data());
    sahref::g_max_leaf = max_leaf; sahref::g_trav = trav;
    sahref::g_order.resize(boxes.size());
    for (uint32_t i = 0; i < boxes.size(); i++) sahref::g_order[i] = i;
    sahref::g_ref.clear();
    RefNode rr{}; rr.first = first; rr.count = count; rr.left = -1;
    for (int a = 0; a < 3; a++) { rr.lo[a] = INFINITY; rr.hi[a] = -INFINITY; }
    for (uint32_t i = first; i < first + count; i++)
        for (int a = 0; a < 3; a++) { rr.lo[a] = std::min(rr.lo[a], boxes[i].lo[a]); rr.hi[a] = std::max(rr.hi[a], boxes[i].hi[a]); }
    sahref::g_ref.push_back(rr);
    sahref::ref_build(0);
    g_dev = dev; g_dev_count = dev_count; g_order = order;
    if (sahref::g_ref[0].left < 0) { // the whole tree is one leaf: a root with one child
        if (dev_count != 1 || dev[0].child[1] != kInvalidRef) { std::printf("DIFF single-leaf tree\n"); return false; }
        g_nodes4++;
        return same_leaf(dev[0].child[0], sahref::g_ref[0], first);
    }
    const int before = g_nodes4;
    if (!compare4(0, 0, first, 0)) return false;
    if ((uint32_t)(g_nodes4 - before) != dev_count) { std::printf("DIFF node count: device %u, reached %d\n", dev_count, g_nodes4 - before); return false; }
    return true;
}
// the box the refit must have put into slot k of node i; compares everything below on the way
bool refit_box(const Node4* nodes, uint32_t i, int k, const rfw_rt_triangle* tris, const uint32_t* order, float* lo, float* hi)
{
    for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    const uint32_t c = nodes[i].child[k];
    if (c & 0x80000000u) {
        const uint32_t first = c & 0x07FFFFFFu, count = ((c >> 27) & 15u) + 1u;
        for (uint32_t j = 0; j < count; j++) {
            const float* v = reinterpret_cast<const float*>(tris + order[first + j]);
            for (int d = 0; d < 3; d++) {
                const float l = std::min(v[d], std::min(v[4 + d], v[8 + d])), h = std::max(v[d], std::max(v[4 + d], v[8 + d]));
                const float e = 1e-4f + 4e-6f * std::max(std::fabs(l), std::fabs(h));
                lo[d] = std::min(lo[d], l - e); hi[d] = std::max(hi[d], h + e);
            }
        }
    } else
        for (int kk = 0; kk < 4; kk++) {
            if (nodes[c].child[kk] == kInvalidRef) continue;
            float l2[3], h2[3];
            if (!refit_box(nodes, c, kk, tris, order, l2, h2)) return false;
            for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], l2[a]); hi[a] = std::max(hi[a], h2[a]); }
        }
    const Node4& N = nodes[i];
    if (N.lox[k] != lo[0] || N.loy[k] != lo[1] || N.loz[k] != lo[2] || N.hix[k] != hi[0] || N.hiy[k] != hi[1] || N.hiz[k] != hi[2]) {
        std::printf("DIFF refitted box of child %d of node %u\n", k, i);
        return false;
    }
    return true;
}
} // namespace

int main(int argc, char** argv)
{
    if (argc < 6) { std::fprintf(stderr, "usage: sah_build_emu n seed kind max_leaf trav_cost [trees]\n"); return 2; }
    const uint32_t n = (uint32_t)std::atoi(argv[1]), seed = (uint32_t)std::atoi(argv[2]);
    const int kind = std::atoi(argv[3]), max_leaf = std::atoi(argv[4]);
    const float trav = (float)std::atof(argv[5]);
    const uint32_t n_trees = argc > 6 ? (uint32_t)std::atoi(argv[6]) : 1u;
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<DevBox> boxes(n);
    for (uint32_t i = 0; i < n; i++) {
        float c[3], e[3];
        for (int a = 0; a < 3; a++) { c[a] = U(rng) * 8.0f - 4.0f; e[a] = 0.01f + 0.3f * U(rng) * U(rng); }
        if (kind == 1) { // a mesh-like order: a surface walked in strips (neighbours in memory are neighbours in space, as the level kernels expect)
            const uint32_t w = 64;
            c[0] = 0.1f * (float)(i % w); c[2] = 0.1f * (float)(i / w); c[1] = 0.5f * std::sin(0.7f * c[0]) * std::cos(0.5f * c[2]);
            for (int a = 0; a < 3; a++) { c[a] += 0.02f * U(rng); e[a] = 0.06f; }
        }
        if (kind == 4) for (int a = 0; a < 3; a++) c[a] = std::pow(U(rng), 6.0f) * 100.0f; // very uneven: deep upper tree
        for (int a = 0; a < 3; a++) { boxes[i].lo[a] = c[a] - e[a]; boxes[i].hi[a] = c[a] + e[a]; }
        boxes[i].lo[3] = boxes[i].hi[3] = 0.0f;
    }
    std::vector<uint32_t> order(n + 1, 0xffffffffu);
    if (n_trees <= 1) {
        std::vector<char> ws(sah_workspace_bytes(n));
        std::vector<Node4> nodes(std::max(n, 1u));
        uint32_t node_count = 0;
        const hipError_t e = sah_build(nullptr, boxes.data(), n, ws.data(), ws.size(), nodes.data(), order.data(), &node_count, max_leaf, trav);
        if (e != hipSuccess) { std::printf("DIFF sah_build returned %d\n", (int)e); return 1; }
        if (!check_tree(boxes, 0, n, nodes.data(), node_count, order.data(), max_leaf, trav)) return 1;
        if (argc > 6 && n_trees == 0) {
            std::vector<rfw_rt_triangle> tris(n);
            std::memset(tris.data(), 0, tris.size() * sizeof(rfw_rt_triangle));
            for (auto& t : tris) {
                float* v = reinterpret_cast<float*>(&t);
                for (int q = 0; q < 3; q++)
                    for (int d = 0; d < 3; d++) v[4 * q + d] = 40.0f * U(rng) - 20.0f;
            }
            std::vector<uint32_t> parent_slot(node_count, 0xdeadbeefu), n_internal(node_count, 0xdeadbeefu), arrive(node_count, 0xdeadbeefu);
            launch_refit_setup(nullptr, nodes.data(), node_count, parent_slot.data(), n_internal.data());
            launch_refit(nullptr, nodes.data(), node_count, tris.data(), order.data(), parent_slot.data(), n_internal.data(), arrive.data());
            for (int k = 0; k < 4; k++) {
                float lo[3], hi[3];
                if (nodes[0].child[k] != kInvalidRef && !refit_box(nodes.data(), 0, k, tris.data(), order.data(), lo, hi)) return 1;
            }
        }
    } else {
        // meshes of uneven sizes, one after the other
        std::vector<ForestTree> trees(n_trees);
        std::vector<uint32_t> cut(n_trees + 1, 0);
        for (uint32_t t = 1; t < n_trees; t++) cut[t] = 1 + (uint32_t)(rng() % (n - 1));
        cut[n_trees] = n;
        std::sort(cut.begin(), cut.end());
        uint32_t node_base = 0, largest = 0;
        for (uint32_t t = 0; t < n_trees; t++) {
            trees[t].first = cut[t]; trees[t].count = cut[t + 1] - cut[t]; trees[t].node_base = node_base; trees[t].pad = 0;
            node_base += std::max(trees[t].count, 1u);
            largest = std::max(largest, trees[t].count);
            if (trees[t].count == 0) { std::printf("OK (skipped: an empty mesh in this draw)\n"); return 0; }
        }
        std::vector<char> ws(sah_forest_workspace_bytes(n, n_trees));
        std::vector<Node4> nodes(node_base);
        std::vector<uint32_t> counts(n_trees, 0);
        const hipError_t e = sah_build_forest(nullptr, boxes.data(), n, trees.data(), n_trees, largest, ws.data(), ws.size(), nodes.data(), order.data(), counts.data(), max_leaf, trav);
        if (e != hipSuccess) { std::printf("DIFF sah_build_forest returned %d\n", (int)e); return 1; }
        launch_forest_relative_order(nullptr, order.data(), n, trees.data(), n_trees);
        for (uint32_t t = 0; t < n_trees; t++)
            if (!check_tree(boxes, trees[t].first, trees[t].count, nodes.data() + trees[t].node_base, counts[t], order.data() + trees[t].first, max_leaf, trav)) {
                std::printf("(tree %u of %u: [%u, +%u))\n", t, n_trees, trees[t].first, trees[t].count);
                return 1;
            }
    }
    std::printf("OK nodes4=%d leaves=%d groups=%llu wave_collectives=%llu\n", g_nodes4, g_leaves, emu::g_groups_run, emu::g_meetings);
    return 0;
}
