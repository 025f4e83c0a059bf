// packet_model.cpp — CPU model of wavefront-level traversal cost for 8x8-pixel blocks of primary rays on the bench scene, to decide
// BEFORE building it whether a shared-stack "packet phase" (VERDICT r03 #1a) can beat the one-ray-per-lane loop of csrc/traverse.h.
//
// Uses the PRODUCT's host builder and node quantiser (csrc/bvh_host.cpp, device_types.h) on the product's procedural atrium, all geometry
// in one mesh (one BLAS, no instances: the two-level structure costs 1 node per ray on this scene, EXPERIMENTS.md), and emulates
//   per-lane : every lane walks its own stack in lock step (one stack entry per lane per trip, as the compiled loop does);
//              wave cost = trips in which ANY lane tests a node / a leaf  (what SQ_INSTS_VALU pays for)
//   strict   : a packet phase while every lane of the wavefront agrees on (hit set, order) of the node's children, then the per-lane loop
//   masked   : ONE shared stack of (node, lane mask) for the whole traversal, order from the first active lane; a node is visited
//              once by the union of the lanes that hit it
// Build (host only):  hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 -I../../rfw-rs_amd/host -o packet_model packet_model.cpp \
//                     ../../rfw-rs_amd/csrc/bvh_host.cpp ../../rfw-rs_amd/host/{rfw_host,gltf,gltf_export,jpeg,obj}.cpp -lz -pthread
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../rfw-rs_amd/csrc/bvh_host.h"
#include "../../rfw-rs_amd/host/rfw_host.hpp"

using namespace rfwhip;

struct Tri { float v0[3], e1[3], e2[3]; };
struct Ray { float o[3], d[3], inv[3]; };

static std::vector<Node4Q> g_nodes;
static std::vector<Tri> g_tris; // leaf order

struct SlabOut { bool hit[4]; float tn[4]; };
static inline SlabOut slab(const Node4Q& n, const Ray& r, float t)
{
    SlabOut s;
    const float org[3] = {n.ox, n.oy, n.oz}, sc[3] = {n.sx, n.sy, n.sz};
    for (int i = 0; i < 4; i++) {
        float tn = -INFINITY, tf = INFINITY;
        for (int a = 0; a < 3; a++) {
            const float lo = org[a] + (float)((n.qlo[a] >> (8 * i)) & 255u) * sc[a], hi = org[a] + (float)((n.qhi[a] >> (8 * i)) & 255u) * sc[a];
            const float t0 = (lo - r.o[a]) * r.inv[a], t1 = (hi - r.o[a]) * r.inv[a];
            const float tmin = r.inv[a] < 0 ? t1 : t0, tmax = r.inv[a] < 0 ? t0 : t1;
            if (tmin == tmin) tn = std::max(tn, tmin);
            if (tmax == tmax) tf = std::min(tf, tmax);
        }
        s.hit[i] = tf >= tn && tn <= t && tf >= 0.0f && n.child[i] != kInvalidRef;
        s.tn[i] = tn;
    }
    return s;
}
static inline void leaf_test(uint32_t ref, const Ray& r, float& t, uint32_t& tests)
{
    const uint32_t first = ref & kLeafFirstMask, count = ((ref >> 27) & 15u) + 1u;
    for (uint32_t k = 0; k < count; k++) {
        const Tri& T = g_tris[first + k];
        tests++;
        const float hx = r.d[1] * T.e2[2] - r.d[2] * T.e2[1], hy = r.d[2] * T.e2[0] - r.d[0] * T.e2[2], hz = r.d[0] * T.e2[1] - r.d[1] * T.e2[0];
        const float a = T.e1[0] * hx + T.e1[1] * hy + T.e1[2] * hz;
        if (a > -1e-4f && a < 1e-4f) continue;
        const float f = 1.0f / a;
        const float sx = r.o[0] - T.v0[0], sy = r.o[1] - T.v0[1], sz = r.o[2] - T.v0[2];
        const float u = f * (sx * hx + sy * hy + sz * hz);
        if (u < 0 || u > 1) continue;
        const float qx = sy * T.e1[2] - sz * T.e1[1], qy = sz * T.e1[0] - sx * T.e1[2], qz = sx * T.e1[1] - sy * T.e1[0];
        const float v = f * (r.d[0] * qx + r.d[1] * qy + r.d[2] * qz);
        if (v < 0 || u + v > 1) continue;
        const float tt = f * (T.e2[0] * qx + T.e2[1] * qy + T.e2[2] * qz);
        if (tt > 1e-4f && tt < t) t = tt;
    }
}

// children of `cur` a lane goes on with, nearest first
static bool g_static_order = false; // children in the order of the ray octant's copy of the node (make_packet_node) instead of by entry distance
static inline int ordered_children(const Node4Q& n, const SlabOut& s, uint32_t out[4], const Ray* ray = nullptr)
{
    int idx[4], k = 0;
    for (int i = 0; i < 4; i++) if (s.hit[i]) idx[k++] = i;
    if (g_static_order && ray) {
        const uint32_t oct = (ray->inv[0] < 0 ? 1u : 0u) | (ray->inv[1] < 0 ? 2u : 0u) | (ray->inv[2] < 0 ? 4u : 0u);
        const PacketNode pn = make_packet_node(n, oct);
        int kk = 0;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++)
                if (s.hit[i] && n.child[i] == pn.child[j] && n.child[i] != kInvalidRef) { bool dup = false; for (int q = 0; q < kk; q++) dup |= out[q] == n.child[i]; if (!dup) out[kk++] = n.child[i]; }
        return kk;
    }
    std::sort(idx, idx + k, [&](int a, int b) { return s.tn[a] < s.tn[b] || (s.tn[a] == s.tn[b] && a < b); });
    for (int i = 0; i < k; i++) out[i] = n.child[idx[i]];
    return k;
}

struct Lane {
    Ray r; float t = 1e26f;
    std::vector<uint32_t> stack; uint32_t cur = 0; bool done = false;
    uint32_t nodes = 0, tris = 0;
};
struct WaveCost { double node_steps = 0, leaf_steps = 0, tri_steps = 0, uniform = 0, lane_nodes = 0, lane_tris = 0, packet_steps = 0, handover_sp = 0; };

// the compiled loop of traverse.h, in lock step.  `lanes` carry their state (so a packet phase can run first).
static void per_lane_loop(std::vector<Lane>& L, WaveCost& c)
{
    for (;;) {
        bool any = false, any_node = false, any_leaf = false;
        uint32_t first_node = 0xffffffffu; bool uni = true; uint32_t max_count = 0;
        for (auto& l : L) {
            if (l.done) continue;
            any = true;
            if (l.cur == kInvalidRef) { // pop
                if (l.stack.empty()) { l.done = true; continue; }
                l.cur = l.stack.back(); l.stack.pop_back();
            }
        }
        if (!any) break;
        for (auto& l : L) {
            if (l.done || l.cur == kInvalidRef) continue;
            if (!(l.cur & kLeafBit)) {
                any_node = true;
                if (first_node == 0xffffffffu) first_node = l.cur; else if (first_node != l.cur) uni = false;
                const Node4Q& n = g_nodes[l.cur];
                const SlabOut s = slab(n, l.r, l.t);
                uint32_t ch[4]; const int k = ordered_children(n, s, ch, &l.r);
                l.nodes++;
                if (k == 0) l.cur = kInvalidRef;
                else { l.cur = ch[0]; for (int j = k - 1; j >= 1; j--) l.stack.push_back(ch[j]); }
            } else {
                any_leaf = true;
                uint32_t tests = 0;
                leaf_test(l.cur, l.r, l.t, tests);
                l.tris += tests; max_count = std::max(max_count, tests);
                l.cur = kInvalidRef;
            }
        }
        // NOTE: in the real loop a lane that took a node this trip `continue`s (its next trip tests `cur` again); a lane with a leaf falls to the pop.
        if (any_node) { c.node_steps++; if (uni) c.uniform++; }
        if (any_leaf) { c.leaf_steps++; c.tri_steps += max_count; }
    }
}

int main(int argc, char** argv)
{
    const uint32_t tris_target = argc > 1 ? (uint32_t)atoi(argv[1]) : 1048576u;
    const int nblocks = argc > 2 ? atoi(argv[2]) : 2000;
    const uint32_t W = 1920, H = 1080;
    rfw::Scene scene; rfw::Camera3D cam;
    rfw::build_atrium(scene, cam, tris_target, 0xC0FFEE, 2);
    cam.aspect_ratio = (float)W / (float)H;
    std::vector<PrimBox> boxes; std::vector<Tri> tris;
    for (auto& kv : scene.meshes_3d) {
        for (auto& t : kv.second.triangles) {
            Tri T; PrimBox b;
            const float* v0 = &t.vertex0.x; const float* v1 = &t.vertex1.x; const float* v2 = &t.vertex2.x;
            for (int a = 0; a < 3; a++) {
                T.v0[a] = v0[a]; T.e1[a] = v1[a] - v0[a]; T.e2[a] = v2[a] - v0[a];
                const float lo = std::min(v0[a], std::min(v1[a], v2[a])), hi = std::max(v0[a], std::max(v1[a], v2[a]));
                const float pad = 1e-4f + 4e-6f * std::max(std::fabs(lo), std::fabs(hi));
                b.lo[a] = lo - pad; b.hi[a] = hi + pad;
            }
            tris.push_back(T); boxes.push_back(b);
        }
    }
    fprintf(stderr, "%zu triangles\n", tris.size());
    HostBvh4 bvh;
    build_bvh4_host(boxes, 4, 8, bvh, 1.0f);
    g_nodes.resize(bvh.nodes.size());
    for (size_t i = 0; i < bvh.nodes.size(); i++) g_nodes[i] = quantize_node(bvh.nodes[i]);
    g_tris.resize(tris.size());
    for (size_t i = 0; i < tris.size(); i++) g_tris[i] = tris[bvh.prim_order[i]];
    fprintf(stderr, "%zu nodes\n", g_nodes.size());

    const rfw_camera_view_3d view = cam.get_view(W, H);
    auto make_ray = [&](uint32_t px, uint32_t py) {
        Ray r;
        const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
        const float p[3] = {view.p1.x + u * view.right.x + v * view.up.x, view.p1.y + u * view.right.y + v * view.up.y, view.p1.z + u * view.right.z + v * view.up.z};
        float d[3] = {p[0] - view.pos.x, p[1] - view.pos.y, p[2] - view.pos.z};
        const float il = 1.0f / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        r.o[0] = view.pos.x; r.o[1] = view.pos.y; r.o[2] = view.pos.z;
        for (int a = 0; a < 3; a++) { r.d[a] = d[a] * il; r.inv[a] = 1.0f / r.d[a]; }
        return r;
    };

    WaveCost base, strict, masked;
    srand(12345);
    for (int b = 0; b < nblocks; b++) {
        const uint32_t bx = (uint32_t)rand() % (W / 8), by = (uint32_t)rand() % (H / 8);
        std::vector<Lane> L(64);
        for (int l = 0; l < 64; l++) L[l].r = make_ray(bx * 8 + (l & 7), by * 8 + (l >> 3));
        // ---- baseline
        {
            std::vector<Lane> A = L;
            per_lane_loop(A, base);
            for (auto& l : A) { base.lane_nodes += l.nodes; base.lane_tris += l.tris; }
        }
        // ---- strict packet phase: one shared stack while all 64 lanes agree on the hit set and its order
        {
            std::vector<Lane> A = L;
            std::vector<uint32_t> st; uint32_t cur = 0; bool agree = true;
            while (agree) {
                if (cur == kInvalidRef) { if (st.empty()) break; cur = st.back(); st.pop_back(); }
                if (cur & kLeafBit) { // all lanes test the leaf together
                    uint32_t mx = 0;
                    for (auto& l : A) { uint32_t tests = 0; leaf_test(cur, l.r, l.t, tests); l.tris += tests; mx = std::max(mx, tests); }
                    strict.leaf_steps++; strict.tri_steps += mx; cur = kInvalidRef; continue;
                }
                const Node4Q& n = g_nodes[cur];
                uint32_t ch0[4]; int k0 = -1;
                for (auto& l : A) {
                    const SlabOut s = slab(n, l.r, l.t);
                    uint32_t ch[4]; const int k = ordered_children(n, s, ch);
                    if (k0 < 0) { k0 = k; memcpy(ch0, ch, sizeof ch); }
                    else if (k != k0 || memcmp(ch, ch0, k * sizeof(uint32_t)) != 0) { agree = false; break; }
                }
                if (!agree) break; // this node is re-tested by the per-lane loop
                for (auto& l : A) l.nodes++;
                strict.packet_steps++;
                if (k0 == 0) cur = kInvalidRef;
                else { cur = ch0[0]; for (int j = k0 - 1; j >= 1; j--) st.push_back(ch0[j]); }
            }
            strict.handover_sp += st.size();
            bool finished = agree; // stack ran empty in agreement
            for (auto& l : A) { l.stack = st; l.cur = cur; l.done = finished; }
            per_lane_loop(A, strict);
            for (auto& l : A) { strict.lane_nodes += l.nodes; strict.lane_tris += l.tris; }
        }
        // ---- masked packet traversal: shared stack of (ref, mask); order = the first active lane's; visit if any lane hits
        {
            std::vector<Lane> A = L;
            struct E { uint32_t ref; uint64_t mask; };
            std::vector<E> st; E cur{0, ~0ull};
            for (;;) {
                if (cur.ref == kInvalidRef) { if (st.empty()) break; cur = st.back(); st.pop_back(); }
                if (cur.ref & kLeafBit) {
                    uint32_t mx = 0;
                    for (int l = 0; l < 64; l++) if ((cur.mask >> l) & 1) { uint32_t tests = 0; leaf_test(cur.ref, A[l].r, A[l].t, tests); A[l].tris += tests; mx = std::max(mx, tests); }
                    masked.leaf_steps++; masked.tri_steps += mx; cur.ref = kInvalidRef; continue;
                }
                const Node4Q& n = g_nodes[cur.ref];
                uint64_t m[4] = {0, 0, 0, 0}; float key[4] = {INFINITY, INFINITY, INFINITY, INFINITY}; bool anyl = false;
                for (int l = 0; l < 64; l++) if ((cur.mask >> l) & 1) {
                    const SlabOut s = slab(n, A[l].r, A[l].t); // t may have shrunk since the push: lanes drop out here
                    A[l].nodes++; anyl = true;
                    for (int i = 0; i < 4; i++) if (s.hit[i]) { m[i] |= 1ull << l; key[i] = std::min(key[i], s.tn[i]); } // order by the MIN entry distance over the lanes (a wave reduction); cheaper: first lane's
                }
                (void)anyl;
                masked.node_steps++;
                {   // accounting only: the node's LEAF children tested at once, every lane looping over the leaves IT hit (wave cost = the busiest lane)
                    uint32_t worst = 0;
                    for (int l = 0; l < 64; l++) {
                        uint32_t mine = 0;
                        for (int i = 0; i < 4; i++) if (((m[i] >> l) & 1) && n.child[i] != kInvalidRef && (n.child[i] & kLeafBit)) mine += ((n.child[i] >> 27) & 15u) + 1u;
                        worst = std::max(worst, mine);
                    }
                    masked.handover_sp += worst; // (reported as "handover stack": triangle steps with sibling leaves tested concurrently)
                }
                int idx[4], k = 0;
                for (int i = 0; i < 4; i++) if (m[i]) idx[k++] = i;
                std::sort(idx, idx + k, [&](int a, int b2) { return key[a] < key[b2] || (key[a] == key[b2] && a < b2); });
                if (k == 0) cur.ref = kInvalidRef;
                else { cur = E{n.child[idx[0]], m[idx[0]]}; for (int j = k - 1; j >= 1; j--) st.push_back(E{n.child[idx[j]], m[idx[j]]}); }
            }
            for (auto& l : A) { masked.lane_nodes += l.nodes; masked.lane_tris += l.tris; }
        }
    }
    // ---- masked packets with DEFERRED leaves: a leaf is not tested when it comes off the shared stack; the lanes that hit its box queue it, and
    // when some lane holds Q leaves every lane tests its own queue (per-lane triangle loop: wave cost = the busiest lane).  t shrinks later.
    auto deferred = [&](int Q, WaveCost& out) {
        srand(12345);
        for (int b = 0; b < nblocks; b++) {
            const uint32_t bx = (uint32_t)rand() % (W / 8), by = (uint32_t)rand() % (H / 8);
            std::vector<Lane> A(64);
            for (int l = 0; l < 64; l++) A[l].r = make_ray(bx * 8 + (l & 7), by * 8 + (l >> 3));
            std::vector<std::vector<uint32_t>> queue(64);
            auto flush = [&]() {
                uint32_t worst = 0; bool any = false;
                for (int l = 0; l < 64; l++) {
                    uint32_t mine = 0;
                    for (uint32_t ref : queue[l]) { uint32_t tests = 0; leaf_test(ref, A[l].r, A[l].t, tests); mine += tests; A[l].tris += tests; any = true; }
                    queue[l].clear();
                    worst = std::max(worst, mine);
                }
                if (any) { out.leaf_steps++; out.tri_steps += worst; }
            };
            struct E { uint32_t ref; uint64_t mask; };
            std::vector<E> st; E cur{0, ~0ull};
            for (;;) {
                if (cur.ref == kInvalidRef) { if (st.empty()) break; cur = st.back(); st.pop_back(); }
                if (cur.ref & kLeafBit) {
                    bool full = false;
                    for (int l = 0; l < 64; l++) if ((cur.mask >> l) & 1) { queue[l].push_back(cur.ref); full |= (int)queue[l].size() >= Q; }
                    if (full) flush();
                    cur.ref = kInvalidRef; continue;
                }
                const Node4Q& n = g_nodes[cur.ref];
                uint64_t m[4] = {0, 0, 0, 0}; float key[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
                for (int l = 0; l < 64; l++) if ((cur.mask >> l) & 1) {
                    const SlabOut s = slab(n, A[l].r, A[l].t);
                    A[l].nodes++;
                    for (int i = 0; i < 4; i++) if (s.hit[i]) { m[i] |= 1ull << l; key[i] = std::min(key[i], s.tn[i]); }
                }
                out.node_steps++;
                int idx[4], k = 0;
                for (int i = 0; i < 4; i++) if (m[i]) idx[k++] = i;
                std::sort(idx, idx + k, [&](int a, int b2) { return key[a] < key[b2] || (key[a] == key[b2] && a < b2); });
                if (k == 0) cur.ref = kInvalidRef;
                else { cur = E{n.child[idx[0]], m[idx[0]]}; for (int j = k - 1; j >= 1; j--) st.push_back(E{n.child[idx[j]], m[idx[j]]}); }
            }
            flush();
            for (auto& l : A) { out.lane_nodes += l.nodes; out.lane_tris += l.tris; }
        }
    };
    auto rep = [&](const char* name, const WaveCost& c) {
        printf("%-8s per wave: node steps %.1f (uniform %.1f) + packet steps %.1f, leaf steps %.1f, triangle steps %.1f | per lane: nodes %.1f tris %.2f | handover stack %.1f\n", name,
               c.node_steps / nblocks, c.uniform / nblocks, c.packet_steps / nblocks, c.leaf_steps / nblocks, c.tri_steps / nblocks, c.lane_nodes / nblocks / 64, c.lane_tris / nblocks / 64,
               c.handover_sp / nblocks);
    };
    rep("per-lane", base);
    {   // the same one-ray-per-lane loop with the STATIC child order of the octant copies
        g_static_order = true;
        WaveCost st;
        srand(12345);
        for (int b = 0; b < nblocks; b++) {
            const uint32_t bx = (uint32_t)rand() % (W / 8), by = (uint32_t)rand() % (H / 8);
            std::vector<Lane> L(64);
            for (int l = 0; l < 64; l++) L[l].r = make_ray(bx * 8 + (l & 7), by * 8 + (l >> 3));
            per_lane_loop(L, st);
            for (auto& l : L) { st.lane_nodes += l.nodes; st.lane_tris += l.tris; }
        }
        g_static_order = false;
        rep("static", st);
    }
    rep("strict", strict);
    rep("masked", masked);
    for (int Q : {1, 2, 3, 4, 1000}) { WaveCost d; deferred(Q, d); char nm[32]; snprintf(nm, sizeof nm, "defer%d", Q); rep(nm, d); }
    return 0;
}
