// wide_bvh_visits.cpp — planning tool (not product, not test): how many node visits would a k-wide BVH need per ray on the bench scene?
// Builds the binned-SAH BVH2 of oracle/bvh.h over the atrium mesh (through the C API of rfw-rs_amd/host/librfw_host.so), collapses it
// k-wide (largest child surface area first, as the 4-wide collapse does), traces the camera's pinhole rays at 480x270 closest-hit with
// children visited nearest first, and prints node visits and triangle tests per ray for k = 2, 4, 8 — the arithmetic behind the
// "8-wide node" item of DESIGN.md §10.
//   g++ -O2 -std=c++17 -o wide_bvh_visits wide_bvh_visits.cpp -ldl && ./wide_bvh_visits [triangles]
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rfw_pod.h"
#include "../../oracle/bvh.h"

using namespace orc;

struct WNode { // k-wide node: child boxes + refs (>= 0 interior index; < 0: leaf ~ (first, count) packed)
    int n = 0;
    Box box[8];
    int32_t child[8];
    int32_t first[8], count[8];
};

static Box node_box(const BVHNode& n)
{
    Box b;
    b.mn[0] = n.bmin_x; b.mn[1] = n.bmin_y; b.mn[2] = n.bmin_z;
    b.mx[0] = n.bmax_x; b.mx[1] = n.bmax_y; b.mx[2] = n.bmax_z;
    return b;
}

static void collapse(const BVH& bvh, int k, std::vector<WNode>& out, bool even_depth = false)
{
    out.clear();
    struct Job { uint32_t bvh2; uint32_t wide; };
    std::vector<Job> jobs;
    out.emplace_back();
    jobs.push_back({0u, 0u});
    while (!jobs.empty()) {
        const Job j = jobs.back();
        jobs.pop_back();
        std::vector<uint32_t> kids; // BVH2 node indices adopted by this wide node
        const BVHNode& root = bvh.nodes[j.bvh2];
        if (root.count >= 0) { kids.push_back(j.bvh2); }
        else { kids.push_back((uint32_t)root.left_first); kids.push_back((uint32_t)root.left_first + 1); }
        if (even_depth) { // what csrc/sah_build.hip and csrc/lbvh.hip do: every child that is interior is replaced by ITS two children, once
            std::vector<uint32_t> g;
            for (uint32_t c : kids) {
                if (bvh.nodes[c].count >= 0 || root.count >= 0) g.push_back(c);
                else { g.push_back((uint32_t)bvh.nodes[c].left_first); g.push_back((uint32_t)bvh.nodes[c].left_first + 1); }
            }
            kids = g;
        } else
        for (;;) { // open the interior child with the largest surface area while there is room
            int best = -1;
            float area = -1.0f;
            for (size_t c = 0; c < kids.size(); c++) {
                const BVHNode& n = bvh.nodes[kids[c]];
                if (n.count >= 0) continue;
                const float a = node_box(n).half_area();
                if (a > area) { area = a; best = (int)c; }
            }
            if (best < 0 || (int)kids.size() + 1 > k) break;
            const uint32_t open = kids[best];
            kids[best] = (uint32_t)bvh.nodes[open].left_first;
            kids.push_back((uint32_t)bvh.nodes[open].left_first + 1);
        }
        WNode w;
        w.n = (int)kids.size();
        for (int c = 0; c < w.n; c++) {
            const BVHNode& n = bvh.nodes[kids[c]];
            w.box[c] = node_box(n);
            if (n.count >= 0) { w.child[c] = -1; w.first[c] = n.left_first; w.count[c] = n.count; }
            else {
                w.child[c] = (int32_t)out.size();
                out.emplace_back();
                jobs.push_back({kids[c], (uint32_t)w.child[c]});
            }
        }
        out[j.wide] = w;
    }
}

// Collapse by dynamic programming over the SAH cost (Ylitie, Karras, Laine 2017, section 3.1, for k = 4 and leaves of up to max_leaf
// triangles addressed from the parent's slot): C(n, i) = cheapest way to stand for BVH2 subtree n with at most i slots of a parent.
static void collapse_dp(const BVH& bvh, int k, float c_prim, int max_leaf, std::vector<WNode>& out)
{
    const size_t N = bvh.nodes.size();
    std::vector<float> area(N);
    std::vector<int32_t> first(N), prims(N);
    std::vector<float> C(N * 8, 0.0f); // C[n*8 + i], i = 1..k-1
    std::vector<uint8_t> leaf1(N, 0);  // C(n,1) is the leaf alternative
    std::vector<uint8_t> split(N * 8, 0); // split[n*8 + j] = m of the best distribution of j slots (j = 2..k)
    // children have larger indices than their parent in this builder's array? not guaranteed: explicit post-order
    std::vector<uint32_t> order; order.reserve(N);
    { std::vector<uint32_t> st{0u}; while (!st.empty()) { uint32_t n = st.back(); st.pop_back(); order.push_back(n); if (bvh.nodes[n].count < 0) { st.push_back((uint32_t)bvh.nodes[n].left_first); st.push_back((uint32_t)bvh.nodes[n].left_first + 1); } } }
    for (size_t q = order.size(); q-- > 0;) {
        const uint32_t n = order[q];
        const BVHNode& b = bvh.nodes[n];
        area[n] = node_box(b).half_area();
        if (b.count >= 0) {
            first[n] = b.left_first; prims[n] = b.count;
            for (int i = 1; i < k; i++) C[n * 8 + i] = area[n] * b.count * c_prim;
            leaf1[n] = 1;
            continue;
        }
        const uint32_t l = (uint32_t)b.left_first, r = l + 1;
        first[n] = std::min(first[l], first[r]); prims[n] = prims[l] + prims[r];
        auto dist = [&](int j, uint8_t& m_out) { // best split of j slots over the two children
            float best = INFINITY;
            for (int m = 1; m < j; m++) {
                const int a = std::min(m, k - 1), c = std::min(j - m, k - 1);
                const float v = C[l * 8 + a] + C[r * 8 + c];
                if (v < best) { best = v; m_out = (uint8_t)m; }
            }
            return best;
        };
        uint8_t m = 1;
        const float internal = area[n] * 1.0f + dist(k, m);
        split[n * 8 + k] = m;
        const float leaf = prims[n] <= max_leaf ? area[n] * prims[n] * c_prim : INFINITY;
        leaf1[n] = leaf <= internal;
        C[n * 8 + 1] = std::min(leaf, internal);
        for (int i = 2; i < k; i++) {
            const float d = dist(i, m);
            split[n * 8 + i] = m;
            if (d < C[n * 8 + i - 1]) C[n * 8 + i] = d; else { C[n * 8 + i] = C[n * 8 + i - 1]; split[n * 8 + i] = 0; } // 0: use fewer slots
        }
    }
    out.clear();
    struct Job { uint32_t bvh2; uint32_t wide; };
    std::vector<Job> jobs;
    out.emplace_back();
    jobs.push_back({0u, 0u});
    while (!jobs.empty()) {
        const Job j = jobs.back(); jobs.pop_back();
        std::vector<uint32_t> roots; // BVH2 nodes standing in this wide node's slots
        struct F { uint32_t n; int i; };
        std::vector<F> st;
        const BVHNode& rn = bvh.nodes[j.bvh2];
        if (rn.count >= 0 || leaf1[j.bvh2]) roots.push_back(j.bvh2); // whole tree is one leaf
        else { const int m = split[j.bvh2 * 8 + k]; st.push_back({(uint32_t)rn.left_first, m}); st.push_back({(uint32_t)rn.left_first + 1, k - m}); }
        while (!st.empty()) {
            F f = st.back(); st.pop_back();
            f.i = std::min(f.i, k - 1);
            const BVHNode& b = bvh.nodes[f.n];
            while (f.i > 1 && split[f.n * 8 + f.i] == 0) f.i--;
            if (f.i == 1 || b.count >= 0) { roots.push_back(f.n); continue; }
            const int m = split[f.n * 8 + f.i];
            st.push_back({(uint32_t)b.left_first, m}); st.push_back({(uint32_t)b.left_first + 1, f.i - m});
        }
        WNode w; w.n = (int)roots.size();
        for (int c = 0; c < w.n; c++) {
            const uint32_t n = roots[c];
            w.box[c] = node_box(bvh.nodes[n]);
            if (leaf1[n]) { w.child[c] = -1; w.first[c] = first[n]; w.count[c] = prims[n]; }
            else { w.child[c] = (int32_t)out.size(); out.emplace_back(); jobs.push_back({n, (uint32_t)w.child[c]}); }
        }
        out[j.wide] = w;
    }
}

struct Tri { float v0[3], e1[3], e2[3]; };

int main(int argc, char** argv)
{
    const uint32_t target = argc > 1 ? (uint32_t)atoi(argv[1]) : 1048576u;
    void* lib = dlopen("../../rfw-rs_amd/host/librfw_host.so", RTLD_NOW);
    if (!lib) { fprintf(stderr, "run from tools/probes: %s\n", dlerror()); return 1; }
    auto create = (void* (*)())dlsym(lib, "rfwhost_scene_create");
    auto build = (int (*)(void*, const char*, uint32_t, uint32_t, float, uint32_t))dlsym(lib, "rfwhost_build");
    auto mesh = (int (*)(void*, uint32_t, rfw_mesh_data_3d*))dlsym(lib, "rfwhost_mesh_data");
    auto view = (int (*)(void*, uint32_t, uint32_t, rfw_camera_view_3d*))dlsym(lib, "rfwhost_camera_view");
    auto aspect = (int (*)(void*, float))dlsym(lib, "rfwhost_set_aspect");
    void* sc = create();
    build(sc, "atrium", target, 0, 0.0f, 0xC0FFEE);
    rfw_mesh_data_3d md;
    mesh(sc, 0, &md); // the atrium mesh itself (mesh 1, if present, holds the baked spheres)
    const uint32_t n = md.num_triangles;
    std::vector<Box> boxes(n);
    std::vector<float> centers(3 * (size_t)n);
    std::vector<Tri> tris(n);
    for (uint32_t i = 0; i < n; i++) {
        const rfw_rt_triangle& t = md.triangles[i];
        const float* v[3] = {&t.vertex0.x, &t.vertex1.x, &t.vertex2.x};
        boxes[i].reset();
        for (int c = 0; c < 3; c++) boxes[i].grow(v[c]);
        for (int a = 0; a < 3; a++) {
            centers[3 * (size_t)i + a] = 0.5f * (boxes[i].mn[a] + boxes[i].mx[a]);
            tris[i].v0[a] = v[0][a]; tris[i].e1[a] = v[1][a] - v[0][a]; tris[i].e2[a] = v[2][a] - v[0][a];
        }
    }
    BVH bvh;
    build_binned_sah(boxes, centers, bvh);
    const uint32_t W = 480, H = 270;
    aspect(sc, (float)W / H);
    rfw_camera_view_3d cv;
    view(sc, W, H, &cv);
    printf("{\"triangles\": %u, \"bvh2_nodes\": %zu", n, bvh.nodes.size());
    for (int k : {4, 44, 451, 452, 453, 454, 455}) { // 45x: DP collapse, c_prim / max_leaf variants // 44: 4-wide by the even-depth rule of the device builders
        std::vector<WNode> wide;
        if (k >= 450) { const float cp[] = {0, 0.3f, 0.5f, 0.3f, 0.5f, 1.0f}; const int ml[] = {0, 3, 3, 8, 8, 8}; collapse_dp(bvh, 4, cp[k - 450], ml[k - 450], wide); }
        else collapse(bvh, k == 44 ? 4 : k, wide, k == 44);
        double visits = 0, tests = 0, children = 0;
        for (const WNode& w : wide) children += w.n;
        for (uint32_t py = 0; py < H; py++)
            for (uint32_t px = 0; px < W; px++) {
                const float u = (px + 0.5f) / W, v = (py + 0.5f) / H;
                float O[3] = {cv.pos.x, cv.pos.y, cv.pos.z}, D[3];
                const float P[3] = {cv.p1.x + u * cv.right.x + v * cv.up.x, cv.p1.y + u * cv.right.y + v * cv.up.y, cv.p1.z + u * cv.right.z + v * cv.up.z};
                float len = 0;
                for (int a = 0; a < 3; a++) { D[a] = P[a] - O[a]; len += D[a] * D[a]; }
                len = std::sqrt(len);
                float inv[3];
                for (int a = 0; a < 3; a++) { D[a] /= len; inv[a] = 1.0f / D[a]; }
                float t = 1e26f;
                struct E { int32_t node; float tn; };
                std::vector<E> stack{{0, 0.0f}};
                while (!stack.empty()) {
                    const E e = stack.back();
                    stack.pop_back();
                    if (e.tn > t) continue;
                    const WNode& w = wide[(size_t)e.node];
                    visits += 1;
                    struct Hc { float tn; int c; };
                    Hc hit[8];
                    int nh = 0;
                    for (int c = 0; c < w.n; c++) {
                        float tn = 0.0f, tf = t;
                        for (int a = 0; a < 3; a++) {
                            const float t0 = (w.box[c].mn[a] - O[a]) * inv[a], t1 = (w.box[c].mx[a] - O[a]) * inv[a];
                            tn = std::max(tn, std::min(t0, t1));
                            tf = std::min(tf, std::max(t0, t1));
                        }
                        if (tf >= tn) hit[nh++] = {tn, c};
                    }
                    std::sort(hit, hit + nh, [](const Hc& a, const Hc& b) { return a.tn > b.tn; }); // far first onto the stack
                    for (int h = 0; h < nh; h++) {
                        const int c = hit[h].c;
                        if (w.child[c] >= 0) { stack.push_back({w.child[c], hit[h].tn}); continue; }
                        for (int32_t q = 0; q < w.count[c]; q++) { // Moeller-Trumbore (leaves are tested when their parent is visited)
                            const Tri& tr = tris[bvh.prim_indices[(size_t)w.first[c] + q]];
                            tests += 1;
                            const float hx = D[1] * tr.e2[2] - D[2] * tr.e2[1], hy = D[2] * tr.e2[0] - D[0] * tr.e2[2], hz = D[0] * tr.e2[1] - D[1] * tr.e2[0];
                            const float a = tr.e1[0] * hx + tr.e1[1] * hy + tr.e1[2] * hz;
                            if (a > -1e-4f && a < 1e-4f) continue;
                            const float f = 1.0f / a, sx = O[0] - tr.v0[0], sy = O[1] - tr.v0[1], sz = O[2] - tr.v0[2];
                            const float uu = f * (sx * hx + sy * hy + sz * hz);
                            if (uu < 0.0f || uu > 1.0f) continue;
                            const float qx = sy * tr.e1[2] - sz * tr.e1[1], qy = sz * tr.e1[0] - sx * tr.e1[2], qz = sx * tr.e1[1] - sy * tr.e1[0];
                            const float vv = f * (D[0] * qx + D[1] * qy + D[2] * qz);
                            if (vv < 0.0f || uu + vv > 1.0f) continue;
                            const float tt = f * (tr.e2[0] * qx + tr.e2[1] * qy + tr.e2[2] * qz);
                            if (tt > 1e-4f && tt < t) t = tt;
                        }
                    }
                }
            }
        const double rays = (double)W * H;
        printf(", \"k%d\": {\"nodes\": %zu, \"mean_children\": %.2f, \"node_visits_per_ray\": %.2f, \"triangle_tests_per_ray\": %.2f}", k, wide.size(), children / wide.size(),
               visits / rays, tests / rays);
    }
    printf("}\n");
    return 0;
}
