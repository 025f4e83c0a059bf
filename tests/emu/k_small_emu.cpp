// k_small_emu.cpp — TEST INFRASTRUCTURE: runs the device source of the builder's workgroup phase (k_small of rfw-rs_amd/csrc/sah_build.hip, cut out
// into k_small_extract.inc by tests/test_builder_emulated.py) on the CPU under tests/emu/wave_emu.h and compares the tree it builds with a plain
// serial restatement of the same binned SAH (below, written from the kernel's header comments: 16 bins per axis over the centroid bounds, the
// 45 planes priced as count x half area on both sides, the first cheapest in (axis, plane) order, leaf when splitting does not pay and the
// range fits a leaf, stable partition).  Node for node: same ranges, same boxes, same primitive order.
//   usage: k_small_emu <cap: 256 | 512> <primitives> <seed> <kind> <max_leaf> <trav_cost>      prints "OK nodes=<n> leaves=<n> depth=<n>" or the first difference
#include "wave_emu.h"
#include "sah_reference.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>

namespace rfwhip {
struct alignas(16) DevBox {
    float lo[4], hi[4];
};
namespace {
#include "k_small_extract.inc"
}
} // namespace rfwhip

using namespace rfwhip;

using namespace sahref;
namespace {
int g_nodes = 0, g_leaves = 0, g_depth = 0;
bool compare(const SNode* dev, uint32_t d, int r, const uint32_t* order_out, int depth, uint32_t n_dev_nodes)
{
    g_nodes++; g_depth = std::max(g_depth, depth);
    if (d >= n_dev_nodes) { std::printf("DIFF node id %u out of the reservation (%u)\n", d, n_dev_nodes); return false; }
    const SNode& D = dev[d]; const RefNode& R = g_ref[r];
    if (D.first != R.first || D.count != R.count) { std::printf("DIFF range at depth %d: device [%u, +%u) reference [%u, +%u)\n", depth, D.first, D.count, R.first, R.count); return false; }
    for (int a = 0; a < 3; a++)
        if (!(D.lo[a] == R.lo[a]) || !(D.hi[a] == R.hi[a])) { std::printf("DIFF box at depth %d range [%u, +%u) axis %d: device %g %g reference %g %g\n", depth, D.first, D.count, a, D.lo[a], D.hi[a], R.lo[a], R.hi[a]); return false; }
    if ((D.left == 0xffffffffu) != (R.left < 0)) { std::printf("DIFF leaf / inner at depth %d range [%u, +%u): device %s\n", depth, D.first, D.count, D.left == 0xffffffffu ? "leaf" : "inner"); return false; }
    if (R.left < 0) {
        g_leaves++;
        for (uint32_t i = 0; i < R.count; i++)
            if (order_out[R.first + i] != g_order[R.first + i]) { std::printf("DIFF order at position %u: device %u reference %u\n", R.first + i, order_out[R.first + i], g_order[R.first + i]); return false; }
        return true;
    }
    if (depth > 600) { std::printf("DIFF depth\n"); return false; }
    if (dev[D.left].parent != d || dev[D.left + 1].parent != d) { std::printf("DIFF parent link below node %u\n", d); return false; }
    return compare(dev, D.left, R.left, order_out, depth + 1, n_dev_nodes) && compare(dev, D.left + 1, R.left + 1, order_out, depth + 1, n_dev_nodes);
}
} // namespace

template <uint32_t CAP> int run(uint32_t n, uint32_t seed, int kind, int max_leaf, float trav)
{
    std::mt19937 rng(seed);
    std::uniform_real_distribution<float> U(0.0f, 1.0f);
    std::vector<DevBox> boxes(n);
    for (uint32_t i = 0; i < n; i++) {
        float c[3], e[3];
        for (int a = 0; a < 3; a++) { c[a] = U(rng) * 8.0f - 4.0f; e[a] = 0.01f + 0.3f * U(rng) * U(rng); }
        if (kind == 1) for (int a = 0; a < 3; a++) { c[a] = std::floor(c[a] * 2.0f) * 0.5f; e[a] = 0.125f; }          // a lattice: equal costs, equal centroids
        if (kind == 2 && i % 3) { c[0] = 1.0f; c[1] = 2.0f; c[2] = 3.0f; }                                                 // two thirds share one centroid
        if (kind == 3) { c[1] = 0.25f * c[0]; c[2] = 0.0f; e[2] = 0.0f; }                                                    // flat and nearly collinear
        if (kind == 4) for (int a = 0; a < 3; a++) c[a] = std::pow(U(rng), 6.0f) * 100.0f;                                // very uneven
        if (kind == 5) { // pairs at exponentially growing distances: the SAH peels them off one pair at a time (more than kListCap sub-ranges)
            const float d = std::pow(1.3f, (float)(i / 2));
            c[0] = d; c[1] = 0.37f * d; c[2] = -0.11f * d;
            for (int a = 0; a < 3; a++) { c[a] += 0.01f * d * (float)(i & 1u); e[a] = 0.004f * d; }
        }
        for (int a = 0; a < 3; a++) { boxes[i].lo[a] = c[a] - e[a]; boxes[i].hi[a] = c[a] + e[a]; }
        boxes[i].lo[3] = boxes[i].hi[3] = 0.0f;
    }
    std::vector<uint32_t> order_in(n), order_out(n, 0xffffffffu), small(1, 0u);
    for (uint32_t i = 0; i < n; i++) order_in[i] = i;
    std::shuffle(order_in.begin(), order_in.end(), rng);
    if (std::getenv("K_SMALL_EMU_VERBOSE")) std::fprintf(stderr, "kind %d\n", kind);
    const uint32_t n_nodes = 1 + 2 * n;
    std::vector<SNode> nodes(n_nodes);
    std::memset(nodes.data(), 0xff, nodes.size() * sizeof(SNode));
    SNode root;
    for (int a = 0; a < 3; a++) { root.lo[a] = INFINITY; root.hi[a] = -INFINITY; root.cb[a] = 0xffffffffu; root.cb[3 + a] = 0; }
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { root.lo[a] = std::min(root.lo[a], boxes[i].lo[a]); root.hi[a] = std::max(root.hi[a], boxes[i].hi[a]); }
    root.first = 0; root.count = n; root.left = kNone; root.parent = kNone;
    nodes[0] = root;
    Counters ctr{};
    ctr.node_count = 1; ctr.n_small = 1;
    // reference first (it permutes its own copy of the order)
    static_assert(sizeof(Box) == sizeof(DevBox), "");
    g_boxes = reinterpret_cast<const Box*>(boxes.data()); g_max_leaf = max_leaf; g_trav = trav;
    g_order = order_in;
    g_ref.clear();
    RefNode rr{}; rr.first = 0; rr.count = n; rr.left = -1;
    for (int a = 0; a < 3; a++) { rr.lo[a] = root.lo[a]; rr.hi[a] = root.hi[a]; }
    g_ref.push_back(rr);
    ref_build(0);
    emu::run_group(CAP, 0, [&] { k_small<CAP>(boxes.data(), order_in.data(), order_out.data(), small.data(), &ctr, nodes.data(), max_leaf, trav); });
    if (ctr.node_count != 1 + 2 * n) { std::printf("DIFF reservation %u\n", ctr.node_count); return 1; }
    if (!compare(nodes.data(), 0, 0, order_out.data(), 0, n_nodes)) return 1;
    if ((size_t)g_nodes != g_ref.size()) { std::printf("DIFF node count %d vs %zu\n", g_nodes, g_ref.size()); return 1; }
    std::printf("OK nodes=%d leaves=%d depth=%d\n", g_nodes, g_leaves, g_depth);
    return 0;
}

int main(int argc, char** argv)
{
    if (argc < 7) { std::fprintf(stderr, "usage: k_small_emu cap n seed kind max_leaf trav_cost\n"); return 2; }
    const uint32_t cap = (uint32_t)std::atoi(argv[1]), n = (uint32_t)std::atoi(argv[2]), seed = (uint32_t)std::atoi(argv[3]);
    const int kind = std::atoi(argv[4]), max_leaf = std::atoi(argv[5]);
    const float trav = (float)std::atof(argv[6]);
    if (n < 1 || n > cap) return 2;
    return cap == 256 ? run<256>(n, seed, kind, max_leaf, trav) : run<512>(n, seed, kind, max_leaf, trav);
}
