// split_model.cpp — CPU model: what do SPATIAL SPLITS (triangle pre-splitting, Karras & Aila 2013 §4.2 flavour) buy on the bench scene?
// ("nodes per ray is the one lever nobody pulled".)  Decides BEFORE the device builder is touched.
//
// Uses the PRODUCT's host builder and node quantiser (csrc/bvh_host.cpp, device_types.h) on the product's procedural atrium, all geometry in
// one mesh.  References = (triangle, box): unsplit = one per triangle; split = a triangle whose box is much larger than the triangle needs
// is cut at planes of a hierarchical grid over the scene (so neighbouring triangles are cut at the SAME planes) and each part gets the tight
// box of the clipped polygon.  The tree is built over the references; a leaf holds references (a triangle may appear in several leaves —
// the product's tie rule, lowest (instance, triangle) id at equal t, makes that image-neutral).
// Traced: camera rays of random 8x8 blocks (per ray: 4-wide nodes visited, triangles tested; per wavefront: the UNION of visited nodes =
// packet node steps), and from every hit a shadow ray to the sun and one to a point on an area light (any hit; nodes until done).
// Build (host only):  hipcc -O2 -std=c++17 -x hip --offload-arch=gfx950 -I../../rfw-rs_amd/host -o split_model split_model.cpp \
//                     ../../rfw-rs_amd/csrc/bvh_host.cpp ../../rfw-rs_amd/host/{rfw_host,gltf,gltf_export,jpeg,obj}.cpp -lz -pthread
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "../../rfw-rs_amd/csrc/bvh_host.h"
#include "../../rfw-rs_amd/host/rfw_host.hpp"

using namespace rfwhip;

struct Tri { float v[3][3]; float area; };
struct Ray { float o[3], d[3], inv[3]; };
struct Ref { uint32_t tri; PrimBox box; };

static std::vector<Node4Q> g_nodes;
static std::vector<uint32_t> g_leaf_tri; // leaf order -> triangle
static std::vector<Tri> g_tris;

static inline void slab(const Node4Q& n, const Ray& r, float t, bool hit[4], float tn_out[4], float tf_out[4])
{
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
        hit[i] = tf >= tn && tn <= t && tf >= 0.0f && n.child[i] != kInvalidRef;
        tn_out[i] = tn; tf_out[i] = tf;
    }
}
static inline bool tri_test(const Tri& T, const Ray& r, float tmin, float& t)
{
    float e1[3], e2[3];
    for (int a = 0; a < 3; a++) { e1[a] = T.v[1][a] - T.v[0][a]; e2[a] = T.v[2][a] - T.v[0][a]; }
    const float hx = r.d[1] * e2[2] - r.d[2] * e2[1], hy = r.d[2] * e2[0] - r.d[0] * e2[2], hz = r.d[0] * e2[1] - r.d[1] * e2[0];
    const float a = e1[0] * hx + e1[1] * hy + e1[2] * hz;
    if (a > -1e-4f && a < 1e-4f) return false;
    const float f = 1.0f / a;
    const float sx = r.o[0] - T.v[0][0], sy = r.o[1] - T.v[0][1], sz = r.o[2] - T.v[0][2];
    const float u = f * (sx * hx + sy * hy + sz * hz);
    if (u < 0 || u > 1) return false;
    const float qx = sy * e1[2] - sz * e1[1], qy = sz * e1[0] - sx * e1[2], qz = sx * e1[1] - sy * e1[0];
    const float v = f * (r.d[0] * qx + r.d[1] * qy + r.d[2] * qz);
    if (v < 0 || u + v > 1) return false;
    const float tt = f * (e2[0] * qx + e2[1] * qy + e2[2] * qz);
    if (tt > tmin && tt < t) { t = tt; return true; }
    return false;
}

struct Count { double nodes = 0, tris = 0, rays = 0, max_nodes = 0; };
// one ray, static octant order approximated by entry distance (closest hit) / far first (any hit towards the sun)
static bool trace(const Ray& r, float tmin, float& t, bool any_hit, bool far_first, Count& c, std::vector<uint32_t>* visited = nullptr)
{
    uint32_t stack[128]; int sp = 0; uint32_t cur = 0; uint32_t nodes = 0;
    bool found = false;
    for (;;) {
        if (!(cur & kLeafBit)) {
            const Node4Q& n = g_nodes[cur];
            bool hit[4]; float tn[4], tf[4];
            slab(n, r, t, hit, tn, tf);
            nodes++;
            if (visited) visited->push_back(cur);
            int idx[4], k = 0;
            for (int i = 0; i < 4; i++) if (hit[i]) idx[k++] = i;
            if (far_first) std::sort(idx, idx + k, [&](int a, int b) { return tf[a] > tf[b]; });
            else std::sort(idx, idx + k, [&](int a, int b) { return tn[a] < tn[b]; });
            for (int j = k - 1; j >= 1; j--) stack[sp++] = n.child[idx[j]];
            if (k) { cur = n.child[idx[0]]; continue; }
        } else {
            const uint32_t first = cur & kLeafFirstMask, count = ((cur >> 27) & 15u) + 1u;
            for (uint32_t k = 0; k < count; k++) {
                c.tris++;
                if (tri_test(g_tris[g_leaf_tri[first + k]], r, tmin, t)) { found = true; if (any_hit) break; }
            }
            if (any_hit && found) break;
        }
        if (sp == 0) break;
        cur = stack[--sp];
    }
    c.nodes += nodes; c.rays++; c.max_nodes = std::max(c.max_nodes, (double)nodes);
    return found;
}

// ---- clipping: polygon (<= 9 vertices) against an axis plane
struct Poly { int n; float p[10][3]; };
static void clip(const Poly& in, int axis, float pos, bool keep_below, Poly& out)
{
    out.n = 0;
    for (int i = 0; i < in.n; i++) {
        const float* a = in.p[i]; const float* b = in.p[(i + 1) % in.n];
        const bool ia = keep_below ? a[axis] <= pos : a[axis] >= pos, ib = keep_below ? b[axis] <= pos : b[axis] >= pos;
        if (ia) { memcpy(out.p[out.n++], a, 12); }
        if (ia != ib) {
            const float t = (pos - a[axis]) / (b[axis] - a[axis]);
            float* q = out.p[out.n++];
            for (int k = 0; k < 3; k++) q[k] = a[k] + t * (b[k] - a[k]);
            q[axis] = pos;
        }
    }
}
static PrimBox poly_box(const Poly& p)
{
    PrimBox b; for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
    for (int i = 0; i < p.n; i++) for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], p.p[i][a]); b.hi[a] = std::max(b.hi[a], p.p[i][a]); }
    return b;
}
static inline float box_area(const PrimBox& b)
{
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

int main(int argc, char** argv)
{
    const uint32_t tris_target = argc > 1 ? (uint32_t)atoi(argv[1]) : 1048576u;
    const int nblocks = argc > 2 ? atoi(argv[2]) : 1500;
    const float beta = argc > 3 ? (float)atof(argv[3]) : 0.3f;   // split budget: references added / triangles
    const int max_leaf = argc > 4 ? atoi(argv[4]) : 4;
    const uint32_t W = 1920, H = 1080;
    rfw::Scene scene; rfw::Camera3D cam;
    rfw::build_atrium(scene, cam, tris_target, 0xC0FFEE, 2);
    cam.aspect_ratio = (float)W / (float)H;
    scene.update_lights();
    PrimBox root; for (int a = 0; a < 3; a++) { root.lo[a] = INFINITY; root.hi[a] = -INFINITY; }
    for (auto& kv : scene.meshes_3d)
        for (auto& t : kv.second.triangles) {
            Tri T; const float* v[3] = {&t.vertex0.x, &t.vertex1.x, &t.vertex2.x};
            for (int k = 0; k < 3; k++) for (int a = 0; a < 3; a++) { T.v[k][a] = v[k][a]; root.lo[a] = std::min(root.lo[a], v[k][a]); root.hi[a] = std::max(root.hi[a], v[k][a]); }
            float e1[3], e2[3]; for (int a = 0; a < 3; a++) { e1[a] = T.v[1][a] - T.v[0][a]; e2[a] = T.v[2][a] - T.v[0][a]; }
            const float cx = e1[1] * e2[2] - e1[2] * e2[1], cy = e1[2] * e2[0] - e1[0] * e2[2], cz = e1[0] * e2[1] - e1[1] * e2[0];
            T.area = 0.5f * std::sqrt(cx * cx + cy * cy + cz * cz);
            g_tris.push_back(T);
        }
    const size_t n = g_tris.size();
    fprintf(stderr, "%zu triangles, %zu area / %zu directional lights, root box area %.1f\n", n, scene.area_lights.size(), scene.directional_lights.size(), box_area(root));

    auto tri_box = [&](uint32_t i) { Poly p; p.n = 3; memcpy(p.p, g_tris[i].v, 36); return poly_box(p); };
    auto pad = [&](PrimBox b) { for (int a = 0; a < 3; a++) { const float e = 1e-4f + 4e-6f * std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])); b.lo[a] -= e; b.hi[a] += e; } return b; };

    // ---- references: priority per triangle (Karras & Aila 2013, eq. 5-ish): (2^-level * (A_box - A_ideal))^(1/3), A_ideal = |n_x| + |n_y| + |n_z|
    // of the cross product (the box area a triangle of this orientation cannot do without), level = depth of the coarsest grid plane that cuts the box
    auto top_plane = [&](const PrimBox& b, int& axis, float& pos) -> int { // coarsest median plane of the root's hierarchical grid crossing b
        for (int level = 1; level < 24; level++) {
            const int cells = 1 << level;
            for (int a = 0; a < 3; a++) {
                const float ext = root.hi[a] - root.lo[a];
                const int c0 = (int)std::floor((b.lo[a] - root.lo[a]) / ext * cells), c1 = (int)std::floor((b.hi[a] - root.lo[a]) / ext * cells);
                if (c1 > c0) { // crosses at least one plane of this level: take the one nearest the box centre
                    const float mid = 0.5f * (b.lo[a] + b.hi[a]);
                    int k = (int)std::round((mid - root.lo[a]) / ext * cells); k = std::min(std::max(k, c0 + 1), c1);
                    axis = a; pos = root.lo[a] + ext * (float)k / (float)cells;
                    if (pos > b.lo[a] && pos < b.hi[a]) return level;
                }
            }
        }
        return -1;
    };
    std::vector<double> prio(n, 0.0); double prio_sum = 0;
    for (uint32_t i = 0; i < n; i++) {
        const PrimBox b = tri_box(i);
        float e1[3], e2[3]; for (int a = 0; a < 3; a++) { e1[a] = g_tris[i].v[1][a] - g_tris[i].v[0][a]; e2[a] = g_tris[i].v[2][a] - g_tris[i].v[0][a]; }
        const float ideal = std::fabs(e1[1] * e2[2] - e1[2] * e2[1]) + std::fabs(e1[2] * e2[0] - e1[0] * e2[2]) + std::fabs(e1[0] * e2[1] - e1[1] * e2[0]);
        int axis; float pos; const int level = top_plane(b, axis, pos);
        if (level < 0) continue;
        const double excess = std::max(0.0, (double)box_area(b) - (double)ideal);
        prio[i] = std::cbrt(std::ldexp(1.0, -level) * excess);
        prio_sum += prio[i];
    }
    auto build_and_measure = [&](const char* name, const std::vector<Ref>& refs, bool quiet) {
        std::vector<PrimBox> boxes(refs.size());
        for (size_t i = 0; i < refs.size(); i++) boxes[i] = pad(refs[i].box);
        HostBvh4 bvh;
        build_bvh4_host(boxes, max_leaf, 8, bvh, 1.0f);
        g_nodes.resize(bvh.nodes.size());
        for (size_t i = 0; i < bvh.nodes.size(); i++) g_nodes[i] = quantize_node(bvh.nodes[i]);
        g_leaf_tri.resize(refs.size());
        for (size_t i = 0; i < refs.size(); i++) g_leaf_tri[i] = refs[bvh.prim_order[i]].tri;
        const rfw_camera_view_3d view = cam.get_view(W, H);
        auto make_ray = [&](uint32_t px, uint32_t py) {
            Ray r; const float u = ((float)px + 0.5f) / (float)W, v = ((float)py + 0.5f) / (float)H;
            const float p[3] = {view.p1.x + u * view.right.x + v * view.up.x, view.p1.y + u * view.right.y + v * view.up.y, view.p1.z + u * view.right.z + v * view.up.z};
            float d[3] = {p[0] - view.pos.x, p[1] - view.pos.y, p[2] - view.pos.z};
            const float il = 1.0f / std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            r.o[0] = view.pos.x; r.o[1] = view.pos.y; r.o[2] = view.pos.z;
            for (int a = 0; a < 3; a++) { r.d[a] = d[a] * il; r.inv[a] = 1.0f / r.d[a]; }
            return r;
        };
        Count prim, sun, area; double union_nodes = 0, sun_unocc = 0, area_unocc = 0;
        srand(12345);
        for (int b = 0; b < nblocks; b++) {
            const uint32_t bx = (uint32_t)rand() % (W / 8), by = (uint32_t)rand() % (H / 8);
            std::set<uint32_t> uni;
            for (int l = 0; l < 64; l++) {
                const Ray r = make_ray(bx * 8 + (l & 7), by * 8 + (l >> 3));
                float t = 1e26f; std::vector<uint32_t> vis;
                const bool hit = trace(r, 1e-4f, t, false, false, prim, &vis);
                uni.insert(vis.begin(), vis.end());
                if (!hit) continue;
                float P[3]; for (int a = 0; a < 3; a++) P[a] = r.o[a] + t * r.d[a] - 1e-3f * r.d[a];
                if (!scene.directional_lights.empty()) {
                    const auto& L = scene.directional_lights[0];
                    Ray s; for (int a = 0; a < 3; a++) s.o[a] = P[a];
                    s.d[0] = -L.direction.x; s.d[1] = -L.direction.y; s.d[2] = -L.direction.z;
                    for (int a = 0; a < 3; a++) s.inv[a] = 1.0f / s.d[a];
                    float ts = 3e38f; if (!trace(s, 1e-3f, ts, true, true, sun)) sun_unocc++;
                }
                if (!scene.area_lights.empty()) {
                    const auto& L = scene.area_lights[(size_t)(b * 64 + l) % scene.area_lights.size()];
                    Ray s; float d[3] = {L.position.x - P[0], L.position.y - P[1], L.position.z - P[2]};
                    const float len = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                    for (int a = 0; a < 3; a++) { s.o[a] = P[a]; s.d[a] = d[a] / len; s.inv[a] = 1.0f / s.d[a]; }
                    float ts = len - 2e-4f; if (!trace(s, 1e-3f, ts, true, false, area)) area_unocc++;
                }
            }
            union_nodes += (double)uni.size();
        }
        printf("%-10s refs %8zu (+%4.1f %%)  nodes %7zu | primary: %.2f nodes/ray %.2f tris/ray, union %.1f nodes/wavefront | sun: %.2f nodes %.2f tris (%.0f %% unoccluded) | area: %.2f nodes %.2f tris (%.0f %% unoccluded)\n",
               name, refs.size(), 100.0 * ((double)refs.size() / (double)n - 1.0), g_nodes.size(), prim.nodes / prim.rays, prim.tris / prim.rays, union_nodes / nblocks,
               sun.nodes / std::max(sun.rays, 1.0), sun.tris / std::max(sun.rays, 1.0), 100.0 * sun_unocc / std::max(sun.rays, 1.0),
               area.nodes / std::max(area.rays, 1.0), area.tris / std::max(area.rays, 1.0), 100.0 * area_unocc / std::max(area.rays, 1.0));
        fflush(stdout);
    };

    std::vector<Ref> base(n);
    for (uint32_t i = 0; i < n; i++) base[i] = Ref{i, tri_box(i)};
    build_and_measure("unsplit", base, false);

    // ---- threshold strategy: a reference is cut (at the coarsest grid plane through it) while its box wastes more than tau: A_box - A_polygon-ideal > tau
    for (int ti = 5; ti < argc; ti++) {
        const float tau = (float)atof(argv[ti]) * (getenv("RFW_TAU_REL") ? box_area(root) : 1.0f);
        std::vector<Ref> refs;
        struct Item { Poly poly; PrimBox box; };
        for (uint32_t i = 0; i < n; i++) {
            float e1[3], e2[3]; for (int a = 0; a < 3; a++) { e1[a] = g_tris[i].v[1][a] - g_tris[i].v[0][a]; e2[a] = g_tris[i].v[2][a] - g_tris[i].v[0][a]; }
            const float nx = std::fabs(e1[1] * e2[2] - e1[2] * e2[1]), ny = std::fabs(e1[2] * e2[0] - e1[0] * e2[2]), nz = std::fabs(e1[0] * e2[1] - e1[1] * e2[0]);
            const float tri_ideal = nx + ny + nz; // box area of the whole triangle if it were axis-aligned-tight: 2 * projected areas
            Poly p0; p0.n = 3; memcpy(p0.p, g_tris[i].v, 36);
            std::vector<Item> todo{Item{p0, poly_box(p0)}};
            int made = 0;
            while (!todo.empty()) {
                Item it = todo.back(); todo.pop_back();
                int axis; float pos;
                auto poly_area = [&](const Poly& q) { double ax = 0, ay = 0, az = 0; for (int k = 1; k + 1 < q.n; k++) { float u[3], w[3]; for (int a = 0; a < 3; a++) { u[a] = q.p[k][a] - q.p[0][a]; w[a] = q.p[k + 1][a] - q.p[0][a]; } ax += u[1] * w[2] - u[2] * w[1]; ay += u[2] * w[0] - u[0] * w[2]; az += u[0] * w[1] - u[1] * w[0]; } return 0.5 * std::sqrt(ax * ax + ay * ay + az * az); };
                const double share = g_tris[i].area > 0 ? poly_area(it.poly) / g_tris[i].area : 1.0;
                const double waste = (double)box_area(it.box) - share * tri_ideal;
                bool can = top_plane(it.box, axis, pos) >= 0;
                if (getenv("RFW_SPLIT_MID")) { axis = 0; for (int a = 1; a < 3; a++) if (it.box.hi[a] - it.box.lo[a] > it.box.hi[axis] - it.box.lo[axis]) axis = a; pos = 0.5f * (it.box.lo[axis] + it.box.hi[axis]); can = it.box.hi[axis] > it.box.lo[axis]; }
                if (waste <= tau || made > 4096 || !can) { refs.push_back(Ref{i, it.box}); continue; }
                Poly lo, hi; clip(it.poly, axis, pos, true, lo); clip(it.poly, axis, pos, false, hi);
                if (lo.n < 3 || hi.n < 3) { refs.push_back(Ref{i, it.box}); continue; }
                made++;
                todo.push_back(Item{lo, poly_box(lo)}); todo.push_back(Item{hi, poly_box(hi)});
            }
        }
        char nm[32]; snprintf(nm, sizeof nm, "tau %.3g", tau);
        build_and_measure(nm, refs, false);
    }
    if (argc <= 5)
    for (float be : {beta * 0.33f, beta, beta * 2.0f}) {
        // ---- split: triangle i gets floor(budget * prio / sum) splits, applied recursively at the coarsest grid plane, budget shared by box area
        const double budget = (double)be * (double)n;
        std::vector<Ref> refs; refs.reserve((size_t)((1.0 + be) * n) + 16);
        struct Item { Poly poly; PrimBox box; int splits; };
        for (uint32_t i = 0; i < n; i++) {
            int s = prio_sum > 0 ? (int)std::floor(budget * prio[i] / prio_sum) : 0;
            Poly p0; p0.n = 3; memcpy(p0.p, g_tris[i].v, 36);
            std::vector<Item> todo{Item{p0, poly_box(p0), s}};
            while (!todo.empty()) {
                Item it = todo.back(); todo.pop_back();
                int axis; float pos;
                if (it.splits <= 0 || top_plane(it.box, axis, pos) < 0) { refs.push_back(Ref{i, it.box}); continue; }
                Poly lo, hi; clip(it.poly, axis, pos, true, lo); clip(it.poly, axis, pos, false, hi);
                if (lo.n < 3 || hi.n < 3) { refs.push_back(Ref{i, it.box}); continue; }
                const PrimBox bl = poly_box(lo), bh = poly_box(hi);
                const float al = box_area(bl), ah = box_area(bh);
                const int rest = it.splits - 1;
                const int sl = (int)std::round((double)rest * al / std::max(al + ah, 1e-30f));
                todo.push_back(Item{lo, bl, sl}); todo.push_back(Item{hi, bh, rest - sl});
            }
        }
        char nm[32]; snprintf(nm, sizeof nm, "split %.2f", be);
        build_and_measure(nm, refs, false);
    }
    return 0;
}
