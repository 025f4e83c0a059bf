// api_query.cpp — ray queries in the shape of the reference's CPU path (crates/rfw-scene/src/intersector.rs:21-166), debug taps and probes.
#include "api_internal.h"

using namespace rfwapi;

namespace rfwapi {

} // namespace rfwapi

extern "C" {

// Host-only self test of the acceleration-structure code that runs on the CPU (no HIP call): builds the 4-wide binned-SAH BVH over
// `n` boxes (6 floats each: lo.xyz, hi.xyz), validates it (every primitive in exactly one leaf, child boxes contain their
// subtree), quantises every node and checks that the decoded 8-bit planes still enclose the f32 boxes.  Returns the number of
// violations (0 = pass), or a negative code on bad arguments.
int64_t rfw_hip_selftest_bvh(const float* boxes6, uint32_t n, uint32_t max_leaf, uint32_t threads, uint32_t* out_nodes)
{
    if (n && !boxes6) return RFW_HIP_E_INVALID;
    std::vector<PrimBox> boxes(n);
    for (uint32_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { boxes[i].lo[a] = boxes6[6 * i + a]; boxes[i].hi[a] = boxes6[6 * i + 3 + a]; }
    HostBvh4 bvh;
    build_bvh4_host(boxes, (int)(max_leaf ? max_leaf : 4), (int)(threads ? threads : 1), bvh);
    int64_t errors = (int64_t)validate_bvh4(bvh, boxes);
    for (const Node4& nd : bvh.nodes) {
        const Node4Q q = quantize_node(nd);
        const float o[3] = {q.ox, q.oy, q.oz};
        const float* lo[3] = {nd.lox, nd.loy, nd.loz};
        const float* hi[3] = {nd.hix, nd.hiy, nd.hiz};
        for (int i = 0; i < 4; i++) {
            if (q.child[i] != nd.child[i]) errors++;
            if (nd.child[i] == kInvalidRef) continue;
            for (int a = 0; a < 3; a++) {
                const float scale = a == 0 ? q.sx : (a == 1 ? q.sy : q.sz);
                const float dlo = o[a] + (float)((q.qlo[a] >> (8 * i)) & 0xffu) * scale, dhi = o[a] + (float)((q.qhi[a] >> (8 * i)) & 0xffu) * scale;
                if (dlo > lo[a][i] || dhi < hi[a][i]) errors++;
            }
        }
    }
    // the eight per-octant copies of every node (device_types.h: make_packet_node for the packet kernels, make_octant_node for the one-ray-per-
    // lane kernels): the same children in the same permutation in both, every child's box still enclosed — its ENTRY planes at or below
    // the box on the axes the octant travels upwards, at or above it where it travels downwards, the EXIT planes on the other side — and the
    // children in front-to-back order along the octant's diagonal, empty slots last
    for (const Node4& nd : bvh.nodes) {
        const Node4Q q = quantize_node(nd);
        const float* lo[3] = {nd.lox, nd.loy, nd.loz};
        const float* hi[3] = {nd.hix, nd.hiy, nd.hiz};
        for (uint32_t oct = 0; oct < 8; oct++) {
            const PacketNode pn = make_packet_node(q, oct);
            const Node4Q on = make_octant_node(q, oct);
            const float* pnear[3] = {pn.nx, pn.ny, pn.nz};
            const float* pfar[3] = {pn.fx, pn.fy, pn.fz};
            const float o[3] = {on.ox, on.oy, on.oz}, sc[3] = {on.sx, on.sy, on.sz};
            bool used[4] = {false, false, false, false};
            float prev_key = -INFINITY;
            bool seen_empty = false;
            for (int k = 0; k < 4; k++) {
                if (pn.child[k] != on.child[k]) errors++;
                if (pn.child[k] == kInvalidRef) { seen_empty = true; continue; }
                if (seen_empty) errors++; // an empty slot in front of a child
                int src = -1;
                for (int i = 0; i < 4; i++)
                    if (!used[i] && nd.child[i] == pn.child[k]) { src = i; break; }
                if (src < 0) { errors++; continue; }
                used[src] = true;
                float key = 0.0f;
                for (int a = 0; a < 3; a++) {
                    const bool neg = ((oct >> a) & 1u) != 0u;
                    const float box_near = neg ? hi[a][src] : lo[a][src], box_far = neg ? lo[a][src] : hi[a][src];
                    const float qn = o[a] + (float)((on.qlo[a] >> (8 * k)) & 0xffu) * sc[a], qf = o[a] + (float)((on.qhi[a] >> (8 * k)) & 0xffu) * sc[a];
                    if (neg ? (pnear[a][k] < box_near || pfar[a][k] > box_far || qn < box_near || qf > box_far)
                            : (pnear[a][k] > box_near || pfar[a][k] < box_far || qn > box_near || qf < box_far)) errors++;
                    key += neg ? -pnear[a][k] : pnear[a][k];
                }
                if (key < prev_key) errors++;
                prev_key = key;
            }
            for (int i = 0; i < 4; i++)
                if (nd.child[i] != kInvalidRef && !used[i]) errors++; // a child got lost
        }
    }
    if (out_nodes) *out_nodes = (uint32_t)bvh.nodes.size();
    return errors;
}


static int intersect_impl(void* inst, const float* origins, const float* directions, float t_min, float t_max, uint64_t n, rfw_hip_hit* hits, uint32_t* depth)
{
    LOCK(inst);
    if (n && (!origins || !directions || !hits)) return fail(I, RFW_HIP_E_INVALID, "intersect: null pointer");
    if (!I->synchronized) return fail(I, RFW_HIP_E_STATE, "intersect: scene not synchronized");
    HIP_TRY(I, hipSetDevice(I->device));
    { const int trc = ensure_slot_tlas(I, I); if (trc != RFW_HIP_OK) return trc; }
    const uint64_t chunk = spill_stride(I);
    // scratch kept in the instance: no hipMalloc / hipFree (a device-wide synchronisation that would stall frames in flight) per call
    DevBuf<float>&d_o = I->d_q_o, &d_d = I->d_q_d;
    DevBuf<rfw_hip_hit>& d_h = I->d_q_h;
    DevBuf<uint32_t>& d_depth = I->d_q_depth;
    if (depth) HIP_TRY(I, d_depth.ensure(std::min(n, chunk)));
    HIP_TRY(I, d_o.ensure(3 * std::min(n, chunk)));
    HIP_TRY(I, d_d.ensure(3 * std::min(n, chunk)));
    HIP_TRY(I, d_h.ensure(std::min(n, chunk)));
    const SceneDev sc = scene_dev(I);
    int rc = RFW_HIP_OK;
    for (uint64_t off = 0; off < n && rc == RFW_HIP_OK; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        hipError_t e = hipMemcpyAsync(d_o.ptr, origins + 3 * off, 3 * m * sizeof(float), hipMemcpyHostToDevice, I->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_d.ptr, directions + 3 * off, 3 * m * sizeof(float), hipMemcpyHostToDevice, I->stream);
        if (e == hipSuccess) {
            launch_query_closest(I->stream, sc, d_o.ptr, d_d.ptr, t_min, t_max, m, d_h.ptr, depth ? d_depth.ptr : nullptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(hits + off, d_h.ptr, m * sizeof(rfw_hip_hit), hipMemcpyDeviceToHost, I->stream);
        if (e == hipSuccess && depth) e = hipMemcpyAsync(depth + off, d_depth.ptr, m * sizeof(uint32_t), hipMemcpyDeviceToHost, I->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(I->stream);
        if (e != hipSuccess) rc = fail(I, RFW_HIP_E_DEVICE, std::string("intersect: ") + hipGetErrorString(e));
    }
    if (rc == RFW_HIP_OK) CHECK_OVERFLOW(I);
    return rc;
}

int rfw_hip_intersect(void* inst, const float* origins, const float* directions, float t_min, float t_max, uint64_t n, rfw_hip_hit* hits)
{
    return intersect_impl(inst, origins, directions, t_min, t_max, n, hits, nullptr);
}

int rfw_hip_depth_test(void* inst, const float* origins, const float* directions, float t_min, float t_max, uint64_t n, rfw_hip_hit* hits, uint32_t* depth)
{
    if (inst && n && !depth) { LOCK(inst); return fail(I, RFW_HIP_E_INVALID, "depth_test: null pointer"); }
    return intersect_impl(inst, origins, directions, t_min, t_max, n, hits, depth);
}

static int occludes_impl(Instance* I, const float* origins, const float* directions, float t_min, const float* t_max, uint64_t n, uint8_t* occluded, uint32_t* depth);
int rfw_hip_occludes(void* inst, const float* origins, const float* directions, float t_min, const float* t_max, uint64_t n, uint8_t* occluded)
{
    LOCK(inst);
    return occludes_impl(I, origins, directions, t_min, t_max, n, occluded, nullptr);
}
// occludes() that also reports the 4-wide nodes each any-hit traversal visited (the any-hit counterpart of rfw_hip_depth_test; for the
// planning probes under tools/probes, not part of the trait)
int rfw_hip_debug_occludes_depth(void* inst, const float* origins, const float* directions, float t_min, const float* t_max, uint64_t n, uint8_t* occluded,
                                 uint32_t* depth)
{
    LOCK(inst);
    if (n && !depth) return fail(I, RFW_HIP_E_INVALID, "occludes_depth: null pointer");
    return occludes_impl(I, origins, directions, t_min, t_max, n, occluded, depth);
}
static int occludes_impl(Instance* I, const float* origins, const float* directions, float t_min, const float* t_max, uint64_t n, uint8_t* occluded, uint32_t* depth)
{
    if (n && (!origins || !directions || !t_max || !occluded)) return fail(I, RFW_HIP_E_INVALID, "occludes: null pointer");
    if (!I->synchronized) return fail(I, RFW_HIP_E_STATE, "occludes: scene not synchronized");
    HIP_TRY(I, hipSetDevice(I->device));
    { const int trc = ensure_slot_tlas(I, I); if (trc != RFW_HIP_OK) return trc; }
    const uint64_t chunk = spill_stride(I);
    DevBuf<float>&d_o = I->d_q_o, &d_d = I->d_q_d, &d_t = I->d_q_t;
    DevBuf<uint8_t>& d_r = I->d_q_r;
    HIP_TRY(I, d_o.ensure(3 * std::min(n, chunk)));
    HIP_TRY(I, d_d.ensure(3 * std::min(n, chunk)));
    HIP_TRY(I, d_t.ensure(std::min(n, chunk)));
    HIP_TRY(I, d_r.ensure(std::min(n, chunk)));
    if (depth) HIP_TRY(I, I->d_q_depth.ensure(std::min(n, chunk)));
    const SceneDev sc = scene_dev(I);
    int rc = RFW_HIP_OK;
    for (uint64_t off = 0; off < n && rc == RFW_HIP_OK; off += chunk) {
        const uint64_t m = std::min(chunk, n - off);
        hipError_t e = hipMemcpyAsync(d_o.ptr, origins + 3 * off, 3 * m * sizeof(float), hipMemcpyHostToDevice, I->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_d.ptr, directions + 3 * off, 3 * m * sizeof(float), hipMemcpyHostToDevice, I->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_t.ptr, t_max + off, m * sizeof(float), hipMemcpyHostToDevice, I->stream);
        if (e == hipSuccess) {
            launch_query_any(I->stream, sc, d_o.ptr, d_d.ptr, t_min, d_t.ptr, m, d_r.ptr, depth ? I->d_q_depth.ptr : nullptr);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(occluded + off, d_r.ptr, m, hipMemcpyDeviceToHost, I->stream);
        if (e == hipSuccess && depth) e = hipMemcpyAsync(depth + off, I->d_q_depth.ptr, m * sizeof(uint32_t), hipMemcpyDeviceToHost, I->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(I->stream);
        if (e != hipSuccess) rc = fail(I, RFW_HIP_E_DEVICE, std::string("occludes: ") + hipGetErrorString(e));
    }
    if (rc == RFW_HIP_OK) CHECK_OVERFLOW(I);
    return rc;
}


// TIntersector::intersect4 / occludes4 (crates/rfw-scene/src/intersector.rs:129-166): the reference's 4-wide CPU packets, as calls of the
// batch queries with n = 4.  A packet is SoA as rtbvh's RayPacket4: origin_x[4], origin_y[4], origin_z[4], direction_x[4], ... ; t[4] holds
// the far limits on entry and the hit distances on return (unchanged where nothing was hit), ids are -1 for a miss.
int rfw_hip_intersect4(void* inst, const float* origin_xyz4, const float* direction_xyz4, const float* t_min4, float* t4, int32_t* instance_ids4, int32_t* prim_ids4)
{
    if (!inst) return RFW_HIP_E_INVALID;
    if (!origin_xyz4 || !direction_xyz4 || !t_min4 || !t4 || !instance_ids4 || !prim_ids4) { LOCK(inst); return fail(I, RFW_HIP_E_INVALID, "intersect4: null pointer"); }
    for (int k = 0; k < 4; k++) { // every lane may carry its own interval: one single-ray query each (the device form of a packet is a batch)
        const float o[3] = {origin_xyz4[k], origin_xyz4[4 + k], origin_xyz4[8 + k]}, d[3] = {direction_xyz4[k], direction_xyz4[4 + k], direction_xyz4[8 + k]};
        rfw_hip_hit h;
        const int rc = rfw_hip_intersect(inst, o, d, t_min4[k], t4[k], 1, &h);
        if (rc != RFW_HIP_OK) return rc;
        instance_ids4[k] = h.inst;
        prim_ids4[k] = h.tri;
        if (h.inst >= 0) t4[k] = h.t;
    }
    return RFW_HIP_OK;
}
int rfw_hip_occludes4(void* inst, const float* origin_xyz4, const float* direction_xyz4, const float* t_min4, const float* t_max4, uint8_t* occluded4)
{
    if (!inst) return RFW_HIP_E_INVALID;
    if (!origin_xyz4 || !direction_xyz4 || !t_min4 || !t_max4 || !occluded4) { LOCK(inst); return fail(I, RFW_HIP_E_INVALID, "occludes4: null pointer"); }
    for (int k = 0; k < 4; k++) {
        const float o[3] = {origin_xyz4[k], origin_xyz4[4 + k], origin_xyz4[8 + k]}, d[3] = {direction_xyz4[k], direction_xyz4[4 + k], direction_xyz4[8 + k]};
        const int rc = rfw_hip_occludes(inst, o, d, t_min4[k], t_max4 + k, 1, occluded4 + k);
        if (rc != RFW_HIP_OK) return rc;
    }
    return RFW_HIP_OK;
}

int rfw_hip_debug_lbvh_stress(void* inst, uint32_t n, uint32_t iterations, uint32_t seed, uint64_t* errors, uint64_t* checked)
{
    LOCK(inst);
    if (!errors || !checked || n < 2) return fail(I, RFW_HIP_E_INVALID, "debug_lbvh_stress: needs two or more boxes and both result pointers");
    HIP_TRY(I, hipSetDevice(I->device));
    DevBuf<char> ws; DevBuf<DevBox> boxes; DevBuf<Node4> nodes; DevBuf<uint32_t> order, count, seen; DevBuf<unsigned long long> result;
    auto release = [&]() { ws.release(); boxes.release(); nodes.release(); order.release(); count.release(); seen.release(); result.release(); };
    hipError_t e = ws.ensure(lbvh_workspace_bytes(n));
    if (e == hipSuccess) e = boxes.ensure(n);
    if (e == hipSuccess) e = nodes.ensure(n);
    if (e == hipSuccess) e = order.ensure(n);
    if (e == hipSuccess) e = count.ensure(1);
    if (e == hipSuccess) e = seen.ensure(n);
    if (e == hipSuccess) e = result.ensure(2);
    if (e == hipSuccess) e = hipMemsetAsync(result.ptr, 0, 16, I->stream);
    if (e == hipSuccess) e = lbvh_stress(I->stream, n, iterations, seed, ws.ptr, ws.cap, boxes.ptr, nodes.ptr, order.ptr, count.ptr, seen.ptr, result.ptr);
    unsigned long long host[2] = {0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(host, result.ptr, 16, hipMemcpyDeviceToHost, I->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(I->stream);
    release();
    if (e != hipSuccess) return fail(I, RFW_HIP_E_DEVICE, std::string("debug_lbvh_stress: ") + hipGetErrorString(e));
    *errors = host[0];
    *checked = host[1];
    return RFW_HIP_OK;
}

// what: "hit0"/"hit1" (uint4), "ray_o0"/"ray_o1", "ray_d0"/"ray_d1", "thr0"/"thr1", "sh_o", "sh_d", "sh_e" (float4), "counters",
//       "xforms" (InstanceXform), "normals" (InstanceNormal)
int rfw_hip_debug_read(void* inst, const char* what, void* dst, uint64_t bytes, uint64_t* written)
{
    LOCK(inst);
    if (!what || !dst) return fail(I, RFW_HIP_E_INVALID, "debug_read: null pointer");
    HIP_TRY(I, hipSetDevice(I->device));
    const std::string w(what);
    if (w == "build_counters") { // host-side: full builds, incremental builds, full builds that sent the 48-B heads first, meshes registered with the runtime
        uint32_t pinned = 0;
        for (auto& kv : I->meshes) pinned += kv.second.pinned ? 1u : 0u;
        const uint32_t v[5] = {I->full_builds, I->incremental_builds, I->heads_first_builds, pinned, I->tlas_fused_builds}; // ([4]: instance updates through the fused TLAS path)
        const uint64_t n = std::min<uint64_t>(bytes, sizeof(v));
        std::memcpy(dst, v, n);
        if (written) *written = n;
        return RFW_HIP_OK;
    }
    if (!I->slots.empty() && I->cur_slot != 0 && w != "xforms" && w != "normals" && w != "triangles" && w != "blas_raw" && w != "blas_order") {
        Instance* c = slot_ptr(I, I->cur_slot); // per-frame buffers of the latest frame
        const int rc = rfw_hip_debug_read(c, what, dst, bytes, written);
        if (rc != RFW_HIP_OK) I->err = c->err;
        return rc;
    }
    const void* src = nullptr;
    uint64_t avail = 0;
    const uint64_t q = (uint64_t)I->capacity * 16;
    if (w == "hit0") { src = I->d_hit[0].ptr; avail = q; }
    else if (w == "hit1") { src = I->d_hit[1].ptr; avail = q; }
    else if (w == "ray_o0") { src = I->d_ray_o[0].ptr; avail = q; }
    else if (w == "ray_o1") { src = I->d_ray_o[1].ptr; avail = q; }
    else if (w == "ray_d0") { src = I->d_ray_d[0].ptr; avail = q; }
    else if (w == "ray_d1") { src = I->d_ray_d[1].ptr; avail = q; }
    else if (w == "thr0") { src = I->d_thr[0].ptr; avail = q; }
    else if (w == "thr1") { src = I->d_thr[1].ptr; avail = q; }
    else if (w == "sh_o") { src = I->d_sh_o.ptr; avail = q * kShadowBuckets; } // bucket b at element b * capacity
    else if (w == "sh_d") { src = I->d_sh_d.ptr; avail = q * kShadowBuckets; } // bucket b at element b * capacity
    else if (w == "sh_e") { src = I->d_sh_e.ptr; avail = q * kShadowBuckets; } // bucket b at element b * capacity
    else if (w == "counters") { src = I->d_counters.ptr + (size_t)I->counter_phase * kMaxSub; avail = sizeof(QueueCounters); }
    else if (w == "tlas_raw") { src = I->d_tlas_raw.ptr; avail = (uint64_t)I->d_tlas_raw.cap * sizeof(Node4); }         // the device-built TLAS before quantisation (entries past the node count are stale)
    else if (w == "tlas_nodes") { src = I->d_tlas_nodes.ptr; avail = (uint64_t)I->d_tlas_nodes.cap * sizeof(Node4Q); }
    else if (w == "tlas_oct") { src = I->d_tlas_oct.ptr; avail = (uint64_t)I->d_tlas_oct.cap * sizeof(Node4Q); }        // eight copies, stride = capacity / 8
    else if (w == "tlas_prims") { src = I->d_tlas_prims.ptr; avail = (uint64_t)I->n_valid_instances * 4; }
    else if (w == "xforms") { src = I->d_xforms.ptr; avail = I->n_instances * sizeof(InstanceXform); }
    else if (w == "normals") { src = I->d_normals.ptr; avail = I->n_instances * sizeof(InstanceNormal); }
    else if (w == "blas_raw") { src = I->d_blas_raw.ptr; avail = (uint64_t)I->d_blas_raw.cap * sizeof(Node4); }       // device builders: f32 nodes before quantisation
    else if (w == "blas_order") { src = I->d_blas_order.ptr; avail = (uint64_t)I->d_blas_order.cap * 4; }           // leaf-ordered primitive ids per mesh
    else if (w == "triangles") { src = I->d_triangles.ptr; avail = I->n_tris * sizeof(rfw_rt_triangle); } // static meshes, then the skinned copies
    else return fail(I, RFW_HIP_E_INVALID, "debug_read: unknown buffer " + w);
    const uint64_t n = std::min(bytes, avail);
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    if (n) HIP_TRY(I, hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
    if (written) *written = n;
    return RFW_HIP_OK;
}


// The device functions of k_shade one by one on caller-supplied inputs (tests only; layout in include/rfw_hip.h)
int rfw_hip_debug_eval_shading(void* inst, int op, uint64_t n, const float* in48, float* out12)
{
    LOCK(inst);
    if (op < 0 || op > 5 || (n && (!in48 || !out12)) || n > (1u << 24)) return fail(I, RFW_HIP_E_INVALID, "debug_eval_shading: bad arguments");
    HIP_TRY(I, hipSetDevice(I->device));
    if (op == 4) {
        if (!I->synchronized) return fail(I, RFW_HIP_E_STATE, "debug_eval_shading: light sampling needs a synchronized scene");
        if (I->area_lights.size() + I->point_lights.size() + I->spot_lights.size() + I->directional_lights.size() == 0)
            return fail(I, RFW_HIP_E_STATE, "debug_eval_shading: no lights set");
    }
    HIP_TRY(I, I->d_q_o.ensure(48 * n));
    HIP_TRY(I, I->d_q_d.ensure(12 * n));
    if (I->tables_ready) HIP_TRY(I, hipStreamWaitEvent(I->stream, I->tables_ready, 0));
    rfw_camera_view_3d v;
    std::memset(&v, 0, sizeof(v));
    const CameraParams cam = camera_params(I, v);
    if (n) HIP_TRY(I, hipMemcpyAsync(I->d_q_o.ptr, in48, 48 * n * sizeof(float), hipMemcpyHostToDevice, I->stream));
    launch_eval_shading(I->stream, scene_dev(I), cam, op, (uint32_t)n, I->d_q_o.ptr, I->d_q_d.ptr);
    HIP_TRY(I, hipGetLastError());
    if (n) HIP_TRY(I, hipMemcpyAsync(out12, I->d_q_d.ptr, 12 * n * sizeof(float), hipMemcpyDeviceToHost, I->stream));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    return RFW_HIP_OK;
}

int rfw_hip_bandwidth_probe(void* inst, uint64_t bytes, uint32_t iterations, double* gb_per_s)
{
    LOCK(inst);
    if (!gb_per_s || bytes < 16 || iterations == 0) return fail(I, RFW_HIP_E_INVALID, "bandwidth_probe: bad arguments");
    HIP_TRY(I, hipSetDevice(I->device));
    const uint64_t n = bytes / 16;
    float4 *a = nullptr, *b = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&a, n * 16);
    if (e == hipSuccess) e = hipMalloc((void**)&b, n * 16);
    if (e == hipSuccess) e = hipMemsetAsync(a, 0x3c, n * 16, I->stream);
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        launch_copy_f4(I->stream, a, b, n); // warm
        (void)hipEventRecord(e0, I->stream);
        for (uint32_t k = 0; k < iterations; k++) launch_copy_f4(I->stream, (k & 1u) ? b : a, (k & 1u) ? a : b, n);
        (void)hipEventRecord(e1, I->stream);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (e != hipSuccess) return fail(I, RFW_HIP_E_DEVICE, std::string("bandwidth_probe: ") + hipGetErrorString(e));
    *gb_per_s = ms > 0.0f ? 2.0 * (double)(n * 16) * iterations / (ms * 1e-3) / 1e9 : 0.0;
    return RFW_HIP_OK;
}

int rfw_hip_issue_probe(void* inst, int mix, uint32_t trips, double* g_instructions_per_s)
{
    LOCK(inst);
    if (!g_instructions_per_s || mix < 0 || mix > 2 || trips == 0 || trips > (1u << 20)) return fail(I, RFW_HIP_E_INVALID, "issue_probe: bad arguments");
    HIP_TRY(I, hipSetDevice(I->device));
    hipDeviceProp_t prop;
    HIP_TRY(I, hipGetDeviceProperties(&prop, I->device));
    const uint32_t cus = (uint32_t)std::max(prop.multiProcessorCount, 1);
    float* out = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipMalloc((void**)&out, (size_t)cus * 8u * 256u * sizeof(float));
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms = 0.0f;
    if (e == hipSuccess) {
        launch_issue_probe(I->stream, mix, cus, trips, out); // warm
        (void)hipEventRecord(e0, I->stream);
        launch_issue_probe(I->stream, mix, cus, trips, out);
        (void)hipEventRecord(e1, I->stream);
        e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (out) (void)hipFree(out);
    if (e != hipSuccess) return fail(I, RFW_HIP_E_DEVICE, std::string("issue_probe: ") + hipGetErrorString(e));
    const double wave_instructions = (double)cus * 8.0 * 4.0 * (double)trips * (double)issue_probe_vector_per_trip(mix); // blocks x wavefronts per block x trips x vector instructions per trip
    *g_instructions_per_s = ms > 0.0f ? wave_instructions / (ms * 1e-3) / 1e9 : 0.0;
    return RFW_HIP_OK;
}

} // extern "C"
