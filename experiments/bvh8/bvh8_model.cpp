// bvh8_model.cpp — planning tool (not product, not test): the 8-wide traversal of csrc/bvh8.h modelled on the CPU next to the 4-wide one the
// kernels ran in round 2, on the bench scene's atrium mesh: node visits, triangle tests and stack depth per ray for camera rays and for
// shadow rays towards a fixed sun direction.  Both models follow the device algorithms step by step (no culling of stacked entries when the
// hit distance shrinks; leaves of an 8-wide node tested when the node is visited; octant order instead of a distance sort).
//   hipcc -x hip --offload-arch=gfx950 -O2 -std=c++17 -I../../rfw-rs_amd/csrc -o bvh8_model bvh8_model.cpp ../../rfw-rs_amd/csrc/bvh_host.cpp -ldl -pthread
//   ./bvh8_model [triangles]
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/rfw_pod.h"
#include "bvh8.h"
#include "bvh_host.h"

using namespace rfwhip;

struct Tri { float v0[3], e1[3], e2[3]; };
static bool g_sorted = false; // model: hit children of an 8-wide node by distance instead of octant order
struct Stats { double nodes = 0, tris = 0, steps = 0, rays = 0, hits = 0; int max_stack = 0; };

static bool mt(const Tri& tr, const float* O, const float* D, float t_min, float& t)
{
    const float hx = D[1] * tr.e2[2] - D[2] * tr.e2[1], hy = D[2] * tr.e2[0] - D[0] * tr.e2[2], hz = D[0] * tr.e2[1] - D[1] * tr.e2[0];
    const float a = tr.e1[0] * hx + tr.e1[1] * hy + tr.e1[2] * hz;
    if (a > -1e-4f && a < 1e-4f) return false;
    const float f = 1.0f / a, sx = O[0] - tr.v0[0], sy = O[1] - tr.v0[1], sz = O[2] - tr.v0[2];
    const float uu = f * (sx * hx + sy * hy + sz * hz);
    if (uu < 0.0f || uu > 1.0f) return false;
    const float qx = sy * tr.e1[2] - sz * tr.e1[1], qy = sz * tr.e1[0] - sx * tr.e1[2], qz = sx * tr.e1[1] - sy * tr.e1[0];
    const float vv = f * (D[0] * qx + D[1] * qy + D[2] * qz);
    if (vv < 0.0f || uu + vv > 1.0f) return false;
    const float tt = f * (tr.e2[0] * qx + tr.e2[1] * qy + tr.e2[2] * qz);
    if (tt > t_min && tt < t) { t = tt; return true; }
    return false;
}

static bool slab(const float* lo, const float* hi, const float* O, const float* inv, float t, float& tn_out, float& tf_out)
{
    float tn = 0.0f, tf = t;
    for (int a = 0; a < 3; a++) {
        const float t0 = (lo[a] - O[a]) * inv[a], t1 = (hi[a] - O[a]) * inv[a];
        tn = std::max(tn, std::min(t0, t1));
        tf = std::min(tf, std::max(t0, t1));
    }
    tn_out = tn; tf_out = tf;
    return tf >= tn;
}

// the round-2 device loop: 4-wide nodes, hit children near to far (far_first: by decreasing exit distance), the others stacked
static bool trace4(const HostBvh4& b, const std::vector<Tri>& tris, const float* O, const float* D, float t_min, float& t, bool any, bool far_first, Stats& st)
{
    float inv[3] = {1.0f / D[0], 1.0f / D[1], 1.0f / D[2]};
    std::vector<uint32_t> stack;
    uint32_t cur = 0;
    bool hit = false;
    for (;;) {
        st.steps++;
        if (!(cur & kLeafBit)) {
            st.nodes++;
            const Node4& n = b.nodes[cur];
            struct H { float key; uint32_t c; } h[4];
            int nh = 0;
            for (int i = 0; i < 4; i++) {
                if (n.child[i] == kInvalidRef) continue;
                const float lo[3] = {n.lox[i], n.loy[i], n.loz[i]}, hi[3] = {n.hix[i], n.hiy[i], n.hiz[i]};
                float tn, tf;
                if (slab(lo, hi, O, inv, t, tn, tf)) h[nh++] = {far_first ? -tf : tn, n.child[i]};
            }
            if (nh) {
                std::stable_sort(h, h + nh, [](const H& a, const H& c) { return a.key < c.key; });
                for (int k = nh - 1; k >= 1; k--) stack.push_back(h[k].c);
                st.max_stack = std::max(st.max_stack, (int)stack.size());
                cur = h[0].c;
                continue;
            }
        } else {
            const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                st.tris++;
                if (mt(tris[b.prim_order[first + k]], O, D, t_min, t)) { hit = true; if (any) return true; }
            }
        }
        if (stack.empty()) break;
        cur = stack.back();
        stack.pop_back();
    }
    return hit;
}

// the 8-wide loop of csrc/traverse8.h: one entry per visited node (child_base, hit mask in octant order, interior mask)
static bool trace8(const HostBvh8& b, const std::vector<Tri>& tris, const float* O, const float* D, float t_min, float& t, bool any, bool far_first, Stats& st)
{
    float inv[3] = {1.0f / D[0], 1.0f / D[1], 1.0f / D[2]};
    uint32_t oct = (D[0] < 0.0f ? 1u : 0u) | (D[1] < 0.0f ? 2u : 0u) | (D[2] < 0.0f ? 4u : 0u);
    uint32_t k = (7u - oct) & 7u;           // priority of slot s = s ^ k, highest first: near children first
    if (far_first) k ^= 7u;
    struct G { uint32_t base, hits, imask; };
    std::vector<G> stack;
    G g{0u, 0x80u, 1u << (0x7u ^ k)};        // the root as the only child of a pseudo group: priority bit 7 -> slot 7 ^ k
    bool hit = false;
    for (;;) {
        if (g.hits == 0u) {
            if (stack.empty()) break;
            g = stack.back();
            stack.pop_back();
        }
        st.steps++;
        const int p = 31 - __builtin_clz(g.hits);
        g.hits &= ~(1u << p);
        const uint32_t slot = (uint32_t)p ^ k;
        const uint32_t child = g.base + (uint32_t)__builtin_popcount(g.imask & ((1u << slot) - 1u));
        if (g.hits) { stack.push_back(g); st.max_stack = std::max(st.max_stack, (int)stack.size()); }
        const Node8& n = b.nodes[child];
        st.nodes++;
        const uint32_t imask = n.exps_imask >> 24;
        uint32_t hm = 0;
        float tns[8];
        for (int s = 0; s < 8; s++) {
            float lo[3], hi[3], tn, tf;
            decode_slot8(n, s, lo, hi);
            tns[s] = 1e30f;
            if (lo[0] > hi[0]) continue; // empty
            if (slab(lo, hi, O, inv, t, tn, tf)) { hm |= 1u << s; tns[s] = far_first ? -tf : tn; }
        }
        // leaves of the node, slot order
        uint32_t lh = hm & ~imask;
        while (lh) {
            const int s = __builtin_ctz(lh);
            lh &= lh - 1u;
            const uint32_t m = (n.meta[s >> 2] >> (8 * (s & 3))) & 0xffu;
            if (!m) continue;
            const uint32_t first = n.tri_base + (m & 31u), count = m >> 5;
            for (uint32_t q = 0; q < count; q++) {
                st.tris++;
                if (mt(tris[b.prim_order[first + q]], O, D, t_min, t)) { hit = true; if (any) return true; }
            }
        }
        uint32_t ih = hm & imask, perm = 0;
        for (int s = 0; s < 8; s++) if (ih & (1u << s)) perm |= 1u << ((uint32_t)s ^ k);
        if (g_sorted && ih) { // ideal order: visit by distance (modelled by walking the sorted children as single-child groups pushed far to near)
            int idx[8], ni = 0;
            for (int s = 0; s < 8; s++) if (ih & (1u << s)) idx[ni++] = s;
            std::sort(idx, idx + ni, [&](int a, int c) { return tns[a] < tns[c]; });
            for (int q = ni - 1; q >= 1; q--) stack.push_back(G{n.child_base, 1u << ((uint32_t)idx[q] ^ k), imask});
            st.max_stack = std::max(st.max_stack, (int)stack.size());
            g = G{n.child_base, 1u << ((uint32_t)idx[0] ^ k), imask};
            continue;
        }
        g = G{n.child_base, perm, imask};
    }
    return hit;
}


int main(int argc, char** argv)
{
    const uint32_t target = argc > 1 ? (uint32_t)atoi(argv[1]) : 262144u;
    const uint32_t merge = argc > 2 ? (uint32_t)atoi(argv[2]) : 4u;
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
    mesh(sc, 0, &md);
    const uint32_t n = md.num_triangles;
    std::vector<PrimBox> boxes(n);
    std::vector<Tri> tris(n);
    for (uint32_t i = 0; i < n; i++) {
        const rfw_rt_triangle& t = md.triangles[i];
        const float* v[3] = {&t.vertex0.x, &t.vertex1.x, &t.vertex2.x};
        for (int a = 0; a < 3; a++) {
            boxes[i].lo[a] = std::min(v[0][a], std::min(v[1][a], v[2][a]));
            boxes[i].hi[a] = std::max(v[0][a], std::max(v[1][a], v[2][a]));
            const float e = 1e-4f + 4e-6f * std::max(std::fabs(boxes[i].lo[a]), std::fabs(boxes[i].hi[a]));
            boxes[i].lo[a] -= e; boxes[i].hi[a] += e;
            tris[i].v0[a] = v[0][a]; tris[i].e1[a] = v[1][a] - v[0][a]; tris[i].e2[a] = v[2][a] - v[0][a];
        }
    }
    HostBvh4 b4;
    build_bvh4_host(boxes, 8, 8, b4, 1.0f);
    HostBvh8 b8;
    const bool from2 = argc > 3 && atoi(argv[3]) != 0; // collapse from the BVH2 instead of the 4-wide tree
    const float c_prim = argc > 5 ? (float)atof(argv[5]) : 0.0f; // > 0: SAH-optimal collapse from the BVH2 with this primitive cost
    if (c_prim > 0.0f) {
        build_bvh8_host(boxes, 4, 8, b8, 1.0f, c_prim, argc > 6 ? atoi(argv[6]) : 8); // the product's own collapse (csrc/bvh_host.cpp)
        const uint64_t bad = validate_bvh8(b8, boxes);
        if (bad) { fprintf(stderr, "validate_bvh8: %llu errors\n", (unsigned long long)bad); return 3; }
    } else if (from2) {
        HostBvh4 b2;
        build_bvh2_host(boxes, 8, 8, b2, 1.0f);
        collapse_bvh8_host(b2.nodes.data(), b2.prim_order.data(), n, b8, merge);
    } else
    collapse_bvh8_host(b4.nodes.data(), b4.prim_order.data(), n, b8, merge);
    g_sorted = argc > 4 && atoi(argv[4]) != 0;
    // structure check: every primitive once
    {
        std::vector<uint8_t> seen(n, 0);
        uint64_t bad = 0;
        for (uint32_t p : b8.prim_order) { if (p >= n || seen[p]) bad++; else seen[p] = 1; }
        if (b8.prim_order.size() != n || bad) { fprintf(stderr, "collapse lost primitives: %zu of %u, %llu bad\n", b8.prim_order.size(), n, (unsigned long long)bad); return 2; }
    }
    double kids = 0, leaves = 0;
    for (const Node8& nd : b8.nodes) {
        for (int s = 0; s < 8; s++) {
            float lo[3], hi[3];
            decode_slot8(nd, s, lo, hi);
            if (lo[0] <= hi[0]) { kids++; if (!((nd.exps_imask >> 24) & (1u << s))) leaves++; }
        }
    }
    const uint32_t W = 480, H = 270;
    aspect(sc, (float)W / H);
    rfw_camera_view_3d cv;
    view(sc, W, H, &cv);
    Stats p4, p8, s4n, s4f, s8n, s8f;
    uint64_t mismatch = 0;
    float sun[3] = {0.35f, 0.9f, 0.25f};
    { const float l = std::sqrt(sun[0] * sun[0] + sun[1] * sun[1] + sun[2] * sun[2]); for (float& c : sun) c /= l; }
    for (uint32_t py = 0; py < H; py++)
        for (uint32_t px = 0; px < W; px++) {
            const float u = (px + 0.5f) / W, v = (py + 0.5f) / H;
            float O[3] = {cv.pos.x, cv.pos.y, cv.pos.z}, D[3];
            const float P[3] = {cv.p1.x + u * cv.right.x + v * cv.up.x, cv.p1.y + u * cv.right.y + v * cv.up.y, cv.p1.z + u * cv.right.z + v * cv.up.z};
            float len = 0;
            for (int a = 0; a < 3; a++) { D[a] = P[a] - O[a]; len += D[a] * D[a]; }
            len = std::sqrt(len);
            for (int a = 0; a < 3; a++) D[a] /= len;
            float t4 = 1e26f, t8 = 1e26f;
            p4.rays++; p8.rays++;
            const bool h4 = trace4(b4, tris, O, D, 1e-4f, t4, false, false, p4);
            const bool h8 = trace8(b8, tris, O, D, 1e-4f, t8, false, false, p8);
            if (h4 != h8 || t4 != t8) mismatch++;
            if (!h4) continue;
            float Q[3];
            for (int a = 0; a < 3; a++) Q[a] = O[a] + t4 * D[a] + 1e-3f * sun[a];
            for (int far = 0; far < 2; far++) {
                Stats& a4 = far ? s4f : s4n; Stats& a8 = far ? s8f : s8n;
                float ta = 1e26f, tb = 1e26f;
                a4.rays++; a8.rays++;
                const bool o4 = trace4(b4, tris, Q, sun, 1e-3f, ta, true, far != 0, a4);
                const bool o8 = trace8(b8, tris, Q, sun, 1e-3f, tb, true, far != 0, a8);
                if (o4 != o8) mismatch++;
                a4.hits += o4; a8.hits += o8;
            }
        }
    auto pr = [](const char* name, const Stats& s) {
        printf(", \"%s\": {\"nodes\": %.2f, \"tris\": %.2f, \"steps\": %.2f, \"max_stack\": %d, \"occluded\": %.3f}", name, s.nodes / s.rays, s.tris / s.rays, s.steps / s.rays, s.max_stack,
               s.hits / std::max(1.0, s.rays));
    };
    printf("{\"triangles\": %u, \"nodes4\": %zu, \"nodes8\": %zu, \"children_per_node8\": %.2f, \"leaf_children_per_node8\": %.2f, \"mismatches\": %llu", n, b4.nodes.size(), b8.nodes.size(),
           kids / b8.nodes.size(), leaves / b8.nodes.size(), (unsigned long long)mismatch);
    pr("primary4", p4); pr("primary8", p8); pr("shadow4_near", s4n); pr("shadow8_near", s8n); pr("shadow4_far", s4f); pr("shadow8_far", s8f);
    printf("}\n");
    return 0;
}
