// api_scene.cpp — the scene side of the C ABI: set_* calls and synchronize().  Host-side counterpart of backends/gpu-rt/src/lib.rs:1309-1683:
// BLAS builds per changed mesh (device binned SAH / LBVH, host SAH), skinned copies, the per-frame TLAS, versioned material / light tables, textures.
#include "api_internal.h"
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

using namespace rfwapi;

namespace rfwapi {
// The per-octant node copies follow the quantised arrays: eight copies, stride = the quantised array's capacity, in two forms — Node4Q for
// the one-ray-per-lane kernels (always) and PacketNode for the packet kernels (`want_wide`: 1 KB per node slot, so only where a packet kernel
// can run: ADVICE r04).  (Re)allocates when that capacity changed — the stride with it, so every node in use (`keep` of them) is expanded
// again — or when the packet form is wanted and missing.
hipError_t follow_copies(DevBuf<PacketNode>& wide, DevBuf<Node4Q>& oct, const DevBuf<Node4Q>& nodes, size_t keep, hipStream_t s, bool want_wide)
{
    const size_t want = nodes.cap * kPacketNodeCopies;
    if (!want_wide && wide.ptr) wide.release();
    if (oct.cap == want && (wide.cap == want || !want_wide)) return hipSuccess;
    const bool oct_kept = oct.cap == want; // (only the packet form is missing: the quantised copies stay where they are)
    wide.release();
    if (!oct_kept) oct.release();
    if (want == 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (want_wide) {
        e = hipMalloc((void**)&wide.ptr, want * sizeof(PacketNode));
        if (e != hipSuccess) { wide.ptr = nullptr; return e; }
        wide.cap = want;
    }
    if (!oct_kept) {
        e = hipMalloc((void**)&oct.ptr, want * sizeof(Node4Q));
        if (e != hipSuccess) { oct.ptr = nullptr; return e; }
        oct.cap = want;
    }
    OctantCopies oc;
    oc.wide = wide.ptr; oc.quant = oct.ptr; oc.stride = (uint32_t)nodes.cap;
    launch_expand_nodes(s, nodes.ptr, oc, 0u, (uint32_t)std::min(keep, nodes.cap));
    return hipGetLastError();
}
inline OctantCopies copies_of(const DevBuf<PacketNode>& wide, const DevBuf<Node4Q>& oct)
{
    OctantCopies oc;
    oc.wide = wide.ptr; oc.quant = oct.ptr; oc.stride = (uint32_t)(oct.cap / kPacketNodeCopies);
    return oc;
}
// does this scene need the packet form of its BLAS copies?  Only when some option lets a packet kernel run on it: none does when the options
// say so, or while nobody has chosen (packet_auto) and the scene is beyond the size up to which packets pay (kPacketAutoMaxTriangles).
// set_option("packet_trace") marks the meshes for a rebuild when it asks for packets on a scene built without them.
bool blas_wide_wanted(const Instance* I, uint64_t n_prims)
{
    if (!(I->flags & (kFlagPacketPrimary | kFlagPacketShadow | kFlagPacketShadowFar))) return false;
    return !(I->packet_auto && n_prims > packet_auto_limit());
}

// pad so that the slab test is conservative w.r.t. the rounding of the Moeller-Trumbore arithmetic
inline void pad_box(PrimBox& b)
{
    for (int a = 0; a < 3; a++) {
        const float m = std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a]));
        const float e = 1e-4f + 4e-6f * m;
        b.lo[a] -= e;
        b.hi[a] += e;
    }
}

void build_mesh(Instance* I, MeshHost& m)
{
    const size_t n = m.n_refs; // the caller's triangles + the duplicates of split triangles
    std::vector<PrimBox> boxes(n);
    for (size_t i = 0; i < n; i++) {
        const rfw_rt_triangle& t = m.tris[i];
        const float* v[3] = {&t.vertex0.x, &t.vertex1.x, &t.vertex2.x};
        for (int a = 0; a < 3; a++) {
            boxes[i].lo[a] = std::min(v[0][a], std::min(v[1][a], v[2][a]));
            boxes[i].hi[a] = std::max(v[0][a], std::max(v[1][a], v[2][a]));
        }
        pad_box(boxes[i]);
    }
    for (const SplitPiece& sp : m.pieces) { // spatial splits: every reference of a split triangle has the box of its part
        for (int a = 0; a < 3; a++) { boxes[sp.index].lo[a] = sp.lo[a]; boxes[sp.index].hi[a] = sp.hi[a]; }
        pad_box(boxes[sp.index]);
    }
    build_bvh4_host(boxes, I->sah_max_leaf, I->build_threads, m.bvh, I->sah_trav_cost);
    m.packets.resize(n);
    for (size_t k = 0; k < n; k++) {
        const uint32_t id = m.bvh.prim_order[k];
        const rfw_rt_triangle& t = m.tris[id];
        TriPacket p;
        p.v0x = t.vertex0.x; p.v0y = t.vertex0.y; p.v0z = t.vertex0.z;
        p.tri_id = id; // mesh-local; the global offset is added when the mega-buffer is assembled
        if (id >= m.n_orig) std::memcpy(&p.tri_id, reinterpret_cast<const float*>(&t) + 15, 4); // a duplicate reports the triangle it duplicates
        // edge1 = v1 - v0, edge2 = v2 - v0 (intersection.glsl:7-8), single IEEE subtractions
        p.e1x = t.vertex1.x - t.vertex0.x; p.e1y = t.vertex1.y - t.vertex0.y; p.e1z = t.vertex1.z - t.vertex0.z;
        p.e2x = t.vertex2.x - t.vertex0.x; p.e2y = t.vertex2.y - t.vertex0.y; p.e2z = t.vertex2.z - t.vertex0.z;
        // denom = 1 / dot(gn, gn) (intersection.glsl:32)
        p.inv_gn2 = 1.0f / (t.normal.x * t.normal.x + t.normal.y * t.normal.y + t.normal.z * t.normal.z);
        p.pad = 0.0f;
        m.packets[k] = p;
    }
    m.dirty = false;
}

int ensure_stage(Instance* I, size_t bytes)
{
    const int k = I->stage_next;
    I->stage = I->stage_buf[k];
    if (bytes <= I->stage_cap[k]) return RFW_HIP_OK;
    if (I->stage_buf[k]) (void)hipHostFree(I->stage_buf[k]);
    I->stage_buf[k] = nullptr;
    I->stage = nullptr;
    I->stage_cap[k] = 0;
    const size_t want = std::max<size_t>(bytes * 2, 1 << 20);
    HIP_TRY(I, hipHostMalloc(&I->stage_buf[k], want, hipHostMallocDefault));
    I->stage_cap[k] = want;
    I->stage = I->stage_buf[k];
    return RFW_HIP_OK;
}

int ensure_lbvh_ws(Instance* I, uint32_t n)
{
    const size_t need = lbvh_workspace_bytes(n);
    HIP_TRY(I, I->d_lbvh_ws.ensure(need));
    return RFW_HIP_OK;
}

// the (mesh id, skin id) pairs that need a skinned copy: the mesh carries joint data, the skin exists and the slot is live
std::map<std::pair<uint32_t, int32_t>, DerivedMesh> wanted_derived(const Instance* I)
{
    std::map<std::pair<uint32_t, int32_t>, DerivedMesh> out;
    for (const auto& kv : I->inst_lists) {
        const auto mit = I->meshes.find(kv.first);
        if (mit == I->meshes.end() || mit->second.skin.empty()) continue;
        for (size_t s = 0; s < kv.second.matrices.size(); s++) {
            const int32_t sk = s < kv.second.skin_ids.size() ? kv.second.skin_ids[s] : -1;
            if (sk < 0 || (size_t)sk >= I->skins.size() || I->skins[sk].empty() || is_zero_matrix(kv.second.matrices[s])) continue;
            out[std::make_pair(kv.first, sk)];
        }
    }
    return out;
}

// Appends one record per skinned copy after the static meshes, reserves their regions of the mega-buffers and uploads the joint
// data of the source meshes.  The triangles, BVH and packets of these records are (re)built on the device by build_instances.
int layout_derived(Instance* I, uint32_t& tri_total, uint32_t& node_total)
{
    const uint32_t static_nodes = node_total;
    I->derived = wanted_derived(I);
    I->max_derived_tris = 0;
    std::vector<rfw_joint_data> skin_all;
    std::map<uint32_t, size_t> skin_off;
    for (auto& kv : I->derived) {
        DerivedMesh& d = kv.second;
        d.src_record = I->mesh_index[kv.first.first];
        const MeshHost& src = I->meshes[kv.first.first];
        auto so = skin_off.find(kv.first.first);
        if (so == skin_off.end()) {
            so = skin_off.emplace(kv.first.first, skin_all.size()).first;
            skin_all.insert(skin_all.end(), src.skin.begin(), src.skin.end());
        }
        d.skin_offset = so->second;
        MeshRecord r;
        std::memset(&r, 0, sizeof(r));
        r.tri_base = tri_total;
        r.tri_count = (uint32_t)src.n_orig; // (a skinnable mesh is never split: n_refs == n_orig)
        r.node_base = node_total;
        r.node_count = std::max<uint32_t>(r.tri_count, 1u);
        tri_total += r.tri_count;
        node_total += r.node_count;
        d.record = (uint32_t)I->mesh_records.size();
        I->mesh_records.push_back(r);
        I->record_tri_orig.resize(I->mesh_records.size(), 0u);
        I->record_tri_orig.back() = r.tri_count;
        I->max_derived_tris = std::max(I->max_derived_tris, r.tri_count);
    }
    if (I->derived.empty()) return RFW_HIP_OK;
    if (!I->blas_on_device) { // the host path has no raw-node buffer of its own: keep one for the skinned records only
        I->raw_node_origin = static_nodes;
        HIP_TRY(I, I->d_blas_raw.ensure(node_total - static_nodes));
    }
    HIP_TRY(I, I->d_blas_order.ensure(tri_total));
    HIP_TRY(I, I->d_tri_boxes.ensure(I->max_derived_tris));
    HIP_TRY(I, I->d_bounds_scratch.ensure(8));
    return upload(I, I->d_skin_data, skin_all.data(), skin_all.size());
}

// ALGORITHMIC bytes of the device passes over one mesh of n triangles (rfw_hip_scene_stats.blas_kernel_bytes), per primitive: boxes (176 in,
// 32 out); per builder level above the hand-over size bin (32 + 4 in) and partition (32 + 4 + 4 in, the same out) = 116; the workgroup phase
// (32 + 4 in, 4 out); packets (176 + 4 in, 48 out); and per wide node (~ n / 4) 128 B written by the emitter, 128 read and 64 written by the quantiser
uint64_t build_pass_bytes(uint64_t n)
{
    uint32_t levels = 0;
    for (uint64_t v = n / 512u; v > 0; v >>= 1) levels++;
    return n * (208u + 116u * levels + 40u + 228u) + (n / 4u) * 320u;
}

// Spatial splits at build time.  upload_split_pieces: the box overrides of the records about to be built go up as one table (used only while
// those records are built); patch_boxes: behind k_triangle_boxes, the references of split triangles get the boxes of their parts (`boxes` =
// the record's first box); resolve_duplicates: behind k_make_packets, the packets of duplicate records take their original's id.
int upload_split_pieces(Instance* I, const std::vector<uint32_t>& qs)
{
    I->record_piece_off.assign(I->mesh_records.size(), 0u);
    I->record_piece_n.assign(I->mesh_records.size(), 0u);
    std::vector<SplitPiece> all;
    for (const auto& kv : I->mesh_index) {
        if (std::find(qs.begin(), qs.end(), kv.second) == qs.end()) continue;
        const auto mit = I->meshes.find(kv.first);
        if (mit == I->meshes.end() || mit->second.pieces.empty()) continue;
        I->record_piece_off[kv.second] = (uint32_t)all.size();
        I->record_piece_n[kv.second] = (uint32_t)mit->second.pieces.size();
        all.insert(all.end(), mit->second.pieces.begin(), mit->second.pieces.end());
    }
    if (all.empty()) return RFW_HIP_OK;
    HIP_TRY(I, I->d_split_pieces.ensure(all.size()));
    HIP_TRY(I, I->pins.upload(I->d_split_pieces.ptr, all.data(), all.size() * sizeof(SplitPiece), I->stream));
    return RFW_HIP_OK;
}
inline void patch_boxes(Instance* I, hipStream_t s, uint32_t q, DevBox* boxes)
{
    if (q < I->record_piece_n.size() && I->record_piece_n[q]) launch_patch_boxes(s, I->d_split_pieces.ptr + I->record_piece_off[q], I->record_piece_n[q], boxes);
}
inline void resolve_duplicates(Instance* I, hipStream_t s, uint32_t q)
{
    const MeshRecord& r = I->mesh_records[q];
    if (q < I->record_tri_orig.size() && I->record_tri_orig[q] < r.tri_count)
        launch_resolve_duplicates(s, I->d_packets.ptr + r.tri_base, r.tri_count, r.tri_base, I->record_tri_orig[q], I->d_triangles.ptr + r.tri_base);
}
// boxes of a record's primitives (from the heads while a build runs ahead of the records), references of split triangles patched
inline void record_boxes(Instance* I, hipStream_t s, uint32_t q, DevBox* out)
{
    const MeshRecord& r = I->mesh_records[q];
    if (I->build_from_heads) launch_triangle_boxes(s, I->d_heads.ptr + r.tri_base, r.tri_count, out);
    else launch_triangle_boxes(s, I->d_triangles.ptr + r.tri_base, r.tri_count, out);
    patch_boxes(I, s, q, out);
}
inline void record_packets(Instance* I, hipStream_t s, uint32_t q)
{
    const MeshRecord& r = I->mesh_records[q];
    launch_make_packets(s, I->d_triangles.ptr + r.tri_base, I->d_blas_order.ptr + r.tri_base, r.tri_count, r.tri_base, I->d_packets.ptr + r.tri_base);
    resolve_duplicates(I, s, q);
}

// One static mesh on the device, into the region its record names: boxes -> BVH (binned SAH, or LBVH) -> leaf-ordered packets ->
// quantised nodes.  The triangles are already in d_triangles.  `quantise_count` nodes of the region are quantised (the region is sized for
// the worst case, one node per primitive; nodes behind the tree's own are never referenced).
int build_mesh_device(Instance* I, uint32_t q, uint32_t quantise_count)
{
    const MeshRecord& r = I->mesh_records[q];
    if (r.tri_count == 0) return RFW_HIP_OK;
    record_boxes(I, I->stream, q, I->d_tri_boxes.ptr);
    if (I->blas_sah_on_device) {
        HIP_TRY(I, I->d_sah_ws.ensure(sah_workspace_bytes(r.tri_count)));
        const hipError_t se = sah_build(I->stream, I->d_tri_boxes.ptr, r.tri_count, I->d_sah_ws.ptr, I->d_sah_ws.cap, I->d_blas_raw.ptr + r.node_base,
                                        I->d_blas_order.ptr + r.tri_base, I->d_mesh_node_counts.ptr + q, I->sah_max_leaf, I->sah_trav_cost);
        if (se == hipErrorInvalidValue) { // a tree deeper than the SAH builder's level budget above its LDS phase: LBVH always terminates
            HIP_TRY(I, lbvh_build(I->stream, I->d_tri_boxes.ptr, r.tri_count, I->d_lbvh_ws.ptr, I->d_lbvh_ws.cap, I->d_blas_raw.ptr + r.node_base,
                                  I->d_blas_order.ptr + r.tri_base, I->d_mesh_node_counts.ptr + q));
        } else {
            HIP_TRY(I, se);
        }
    } else {
        HIP_TRY(I, lbvh_build(I->stream, I->d_tri_boxes.ptr, r.tri_count, I->d_lbvh_ws.ptr, I->d_lbvh_ws.cap, I->d_blas_raw.ptr + r.node_base,
                              I->d_blas_order.ptr + r.tri_base, I->d_mesh_node_counts.ptr + q));
    }
    // (a full build makes the packets of all meshes at its end: they need the records, which are still on their way while the trees are built)
    if (!I->build_from_heads) record_packets(I, I->stream, q);
    launch_quantize_nodes(I->stream, I->d_blas_raw.ptr + r.node_base, I->d_blas_nodes.ptr + r.node_base, copies_of(I->d_blas_wide, I->d_blas_oct), r.node_base, quantise_count, I->d_mesh_node_counts.ptr + q);
    return RFW_HIP_OK;
}

// Several meshes: the large ones one after the other on the instance's stream (each fills the device by itself), the small ones side by
// side — a 5120-triangle mesh is ~30 dependent launches of a few microseconds and one 16-byte read-back, i.e. all latency: kBuildLanes host
// threads, each with a stream and scratch of its own, take them from one counter.  The lanes start behind what the instance's stream holds
// (the triangle uploads) and the stream continues behind the lanes.  A builder failure falls back to the one-by-one path for that mesh.
int build_meshes(Instance* I, const std::vector<uint32_t>& qs, bool incremental)
{
    constexpr uint32_t kSmallMesh = 131072;
    std::vector<uint32_t> small, large;
    for (const uint32_t q : qs) (I->blas_sah_on_device && I->mesh_records[q].tri_count && I->mesh_records[q].tri_count <= kSmallMesh ? small : large).push_back(q);
    if (small.size() < 2) { large = qs; small.clear(); }
    int rc = RFW_HIP_OK;
    const int n_lanes = (int)std::min<size_t>(Instance::kBuildLanes, small.size());
    std::vector<std::thread> workers;
    std::vector<hipError_t> lane_err((size_t)std::max(n_lanes, 1), hipSuccess);
    std::vector<uint8_t> redo(I->mesh_records.size(), 0);
    std::atomic<size_t> next{0};
    if (n_lanes) {
        uint32_t max_n = 0;
        for (const uint32_t q : small) max_n = std::max(max_n, I->mesh_records[q].tri_count);
        hipEvent_t start = I->ev_build[1]; // recorded by the caller behind the uploads
        for (int k = 0; k < n_lanes; k++) {
            Instance::BuildLane& L = I->lanes[k];
            if (!L.s) HIP_TRY(I, hipStreamCreateWithFlags(&L.s, hipStreamNonBlocking));
            if (!L.done) HIP_TRY(I, hipEventCreateWithFlags(&L.done, hipEventDisableTiming));
            HIP_TRY(I, L.ws.ensure(sah_workspace_bytes(max_n)));
            HIP_TRY(I, L.boxes.ensure(max_n));
            HIP_TRY(I, hipStreamWaitEvent(L.s, start, 0));
        }
        for (int k = 0; k < n_lanes; k++)
            workers.emplace_back([I, k, &small, &next, &lane_err, &redo, incremental] {
                Instance::BuildLane& L = I->lanes[k];
                if (hipSetDevice(I->device) != hipSuccess) { lane_err[k] = hipErrorInvalidDevice; return; }
                for (size_t i = next.fetch_add(1); i < small.size(); i = next.fetch_add(1)) {
                    const uint32_t q = small[i];
                    const MeshRecord& r = I->mesh_records[q];
                    record_boxes(I, L.s, q, L.boxes.ptr);
                    const hipError_t e = sah_build(L.s, L.boxes.ptr, r.tri_count, L.ws.ptr, L.ws.cap, I->d_blas_raw.ptr + r.node_base, I->d_blas_order.ptr + r.tri_base,
                                                   I->d_mesh_node_counts.ptr + q, I->sah_max_leaf, I->sah_trav_cost);
                    if (e == hipErrorInvalidValue) { redo[q] = 1; continue; } // deeper than the builder's level budget: LBVH, below
                    if (e != hipSuccess) { lane_err[k] = e; return; }
                    if (!I->build_from_heads) record_packets(I, L.s, q);
                    if (incremental) launch_quantize_nodes(L.s, I->d_blas_raw.ptr + r.node_base, I->d_blas_nodes.ptr + r.node_base, copies_of(I->d_blas_wide, I->d_blas_oct), r.node_base, std::max(r.tri_count, 1u), I->d_mesh_node_counts.ptr + q);
                }
                (void)hipEventRecord(L.done, L.s);
            });
    }
    for (const uint32_t q : large) // meanwhile, on the instance's own stream
        if (rc == RFW_HIP_OK) rc = build_mesh_device(I, q, incremental ? std::max(I->mesh_records[q].tri_count, 1u) : 0u);
    for (auto& t : workers) t.join();
    for (int k = 0; k < n_lanes; k++) {
        if (lane_err[k] != hipSuccess && rc == RFW_HIP_OK) rc = fail(I, RFW_HIP_E_DEVICE, std::string("build lane: ") + hipGetErrorString(lane_err[k]));
        (void)hipStreamWaitEvent(I->stream, I->lanes[k].done, 0);
    }
    if (rc != RFW_HIP_OK) return rc;
    for (const uint32_t q : small)
        if (redo[q] && (rc = build_mesh_device(I, q, incremental ? std::max(I->mesh_records[q].tri_count, 1u) : 0u))) return rc;
    return RFW_HIP_OK;
}

// The triangle-id offsets the boundary reports: meshes in mesh-id order, then the skinned copies (= the order of a full build)
void assign_logical_ids(Instance* I)
{
    uint32_t logical = 0;
    for (auto& kv : I->meshes) {
        const auto it = I->mesh_index.find(kv.first);
        if (it == I->mesh_index.end()) continue;
        I->mesh_records[it->second].tri_logical = logical;
        logical += I->record_tri_orig[it->second]; // the caller's triangles: duplicates of split triangles have no id of their own
    }
    for (auto& kv : I->derived) {
        I->mesh_records[kv.second.record].tri_logical = logical;
        logical += I->mesh_records[kv.second.record].tri_count;
    }
    I->n_tris_logical = logical;
}

// BLAS for every mesh on the device: lay the mega-buffers out afresh, upload all triangles, build every mesh
int build_blas_device_full(Instance* I)
{
    const bool kTrace = env_switches().build_trace; // (stderr: where a full build's host time goes)
    const auto t_start = std::chrono::steady_clock::now();
    auto trace = [&](const char* what) {
        if (kTrace) fprintf(stderr, "[build] %-28s %7.3f ms\n", what, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_start).count());
    };
    I->mesh_records.clear();
    I->mesh_index.clear();
    I->record_tri_cap.clear();
    I->record_tri_orig.clear();
    uint32_t tri_total = 0, node_total = 0;
    for (auto& kv : I->meshes) {
        MeshRecord r;
        std::memset(&r, 0, sizeof(r));
        r.tri_base = tri_total;
        r.tri_count = (uint32_t)kv.second.n_refs; // stored primitives: the caller's triangles + the duplicates of split ones
        I->record_tri_orig.push_back((uint32_t)kv.second.n_orig);
        r.node_base = node_total;
        r.node_count = std::max<uint32_t>(r.tri_count, 1u); // worst case (one primitive per leaf => at most n - 1 wide nodes)
        tri_total += r.tri_count;
        node_total += r.node_count;
        I->mesh_index[kv.first] = (uint32_t)I->mesh_records.size();
        I->mesh_records.push_back(r);
        I->record_tri_cap.push_back(r.tri_count);
        kv.second.dirty = false;
    }
    const uint32_t static_nodes = node_total, static_tris = tri_total;
    const size_t n_static = I->mesh_records.size();
    I->raw_node_origin = 0;
    int rc;
    if ((rc = layout_derived(I, tri_total, node_total))) return rc;
    assign_logical_ids(I);
    HIP_TRY(I, I->d_triangles.ensure(tri_total));
    HIP_TRY(I, I->d_packets.ensure(tri_total));
    HIP_TRY(I, I->d_blas_nodes.ensure(node_total));
    HIP_TRY(I, follow_copies(I->d_blas_wide, I->d_blas_oct, I->d_blas_nodes, 0, I->stream, blas_wide_wanted(I, tri_total)));
    HIP_TRY(I, I->d_blas_raw.ensure(node_total));
    HIP_TRY(I, I->d_blas_order.ensure(tri_total));
    for (auto& ev : I->ev_build)
        if (!ev) HIP_TRY(I, hipEventCreate(&ev));
    HIP_TRY(I, hipEventRecord(I->ev_build[0], I->stream));
    {
        std::vector<uint32_t> all_static(n_static);
        for (size_t q = 0; q < n_static; q++) all_static[q] = (uint32_t)q;
        if ((rc = upload_split_pieces(I, all_static))) return rc;
    }
    // Heads first: the builders read 48 of a record's 176 B, so those go up on the instance's stream and the trees are built while the records
    // follow on a stream of their own (registered host memory: both copies are asynchronous).  Only the packets and the renderer need the
    // records; the stream waits for them before the packets are made.  (C4, 185 MB: upload 3.5 ms + kernels 4.5 ms one after the other before.)
    uint32_t max_n = 0;
    size_t k = 0;
    bool any_pinned = false; // (a scene of small meshes only: nothing to overlap; an unregistered mesh is staged by the runtime on either stream)
    for (auto& kv : I->meshes) any_pinned = any_pinned || kv.second.pinned;
    I->build_from_heads = any_pinned && static_tris > 0;
    I->records_timed = I->build_from_heads;
    if (I->build_from_heads) I->heads_first_builds++;
    if (I->build_from_heads) {
        HIP_TRY(I, I->d_heads.ensure(static_tris));
        if (!I->records_stream) HIP_TRY(I, hipStreamCreateWithFlags(&I->records_stream, hipStreamNonBlocking));
        if (!I->ev_heads) HIP_TRY(I, hipEventCreate(&I->ev_heads));
        if (!I->ev_records) HIP_TRY(I, hipEventCreate(&I->ev_records));
        // meshes too small to be registered go through the pinned ring, and FIRST: a staged copy may have to wait for a ring block, i.e. for
        // an earlier copy on its stream — behind a 127 MB record copy that would hold the host back from queuing the build
        for (int pass = 0; pass < 2; pass++) {
            k = 0;
            for (auto& kv : I->meshes) {
                const MeshRecord& r = I->mesh_records[k++];
                max_n = std::max(max_n, r.tri_count);
                if (!r.tri_count || (int)kv.second.pinned != pass) continue;
                const size_t bytes = (size_t)r.tri_count * sizeof(TriHead);
                if (kv.second.pinned) HIP_TRY(I, hipMemcpyAsync(I->d_heads.ptr + r.tri_base, kv.second.heads.data(), bytes, hipMemcpyHostToDevice, I->stream));
                else HIP_TRY(I, I->pins.upload(I->d_heads.ptr + r.tri_base, kv.second.heads.data(), bytes, I->stream));
            }
        }
        HIP_TRY(I, hipEventRecord(I->ev_heads, I->stream));
        HIP_TRY(I, hipStreamWaitEvent(I->records_stream, I->ev_heads, 0)); // (behind the heads on the link, and behind everything the instance's stream waited for)
        for (int pass = 0; pass < 2; pass++) {
            k = 0;
            for (auto& kv : I->meshes) {
                const MeshRecord& r = I->mesh_records[k++];
                if (!r.tri_count || (int)kv.second.pinned != pass) continue;
                const size_t bytes = (size_t)r.tri_count * sizeof(rfw_rt_triangle);
                if (kv.second.pinned) HIP_TRY(I, hipMemcpyAsync(I->d_triangles.ptr + r.tri_base, kv.second.tris.data(), bytes, hipMemcpyHostToDevice, I->records_stream));
                else HIP_TRY(I, I->pins.upload(I->d_triangles.ptr + r.tri_base, kv.second.tris.data(), bytes, I->records_stream));
            }
        }
        HIP_TRY(I, hipEventRecord(I->ev_records, I->records_stream));
        // from here on a copy may be reading the registered host arrays: should anything below fail, the next set_3d_mesh / unload waits for
        // ev_records before it touches them (cleared behind the final synchronisation of a build that went through)
        I->records_pending = true;
        trace("uploads queued");
    } else {
        for (auto& kv : I->meshes) {
            const MeshRecord& r = I->mesh_records[k++];
            max_n = std::max(max_n, r.tri_count);
            if (r.tri_count)
                HIP_TRY(I, hipMemcpyAsync(I->d_triangles.ptr + r.tri_base, kv.second.tris.data(), (size_t)r.tri_count * sizeof(rfw_rt_triangle),
                                          hipMemcpyHostToDevice, I->stream));
        }
    }
    HIP_TRY(I, I->d_tri_boxes.ensure(std::max(max_n, I->max_derived_tris)));
    if ((rc = ensure_lbvh_ws(I, max_n))) return rc;
    HIP_TRY(I, I->d_mesh_node_counts.ensure(std::max<size_t>(n_static, 1)));
    HIP_TRY(I, hipMemsetAsync(I->d_mesh_node_counts.ptr, 0, std::max<size_t>(n_static, 1) * 4, I->stream)); // (an empty mesh is not built: its count stays 0)
    HIP_TRY(I, hipEventRecord(I->ev_build[1], I->stream));
    uint64_t kernel_bytes = 0;
    {
        std::vector<uint32_t> all(n_static);
        for (size_t q = 0; q < n_static; q++) { all[q] = (uint32_t)q; kernel_bytes += build_pass_bytes(I->mesh_records[q].tri_count); }
        // several meshes: ONE build for all of them (sah_build_forest: the meshes lie one after the other in the buffers of a full build, every
        // mesh is a root of the same level-by-level pass) — ~45 launches for the scene instead of ~30 per mesh
        bool forest_done = false;
        // (measured: two meshes of 720 k + 330 k triangles 5.4 ms together, 4.4 ms one after the other; 65 meshes 5.5 ms against 10.4 on lanes, 38 one by one)
        constexpr size_t forest_min = 4;
        if (I->blas_sah_on_device && n_static >= forest_min && static_tris > 0 && !env_switches().no_forest) {
            std::vector<ForestTree> trees(n_static);
            for (size_t q = 0; q < n_static; q++) trees[q] = ForestTree{I->mesh_records[q].tri_base, I->mesh_records[q].tri_count, I->mesh_records[q].node_base, 0u};
            HIP_TRY(I, I->d_forest.ensure(n_static));
            HIP_TRY(I, I->pins.upload(I->d_forest.ptr, trees.data(), n_static * sizeof(ForestTree), I->stream));
            HIP_TRY(I, I->d_tri_boxes.ensure(std::max(static_tris, I->max_derived_tris)));
            HIP_TRY(I, I->d_sah_ws.ensure(sah_forest_workspace_bytes(static_tris, (uint32_t)n_static)));
            if (I->build_from_heads) launch_triangle_boxes(I->stream, I->d_heads.ptr, static_tris, I->d_tri_boxes.ptr);
            else launch_triangle_boxes(I->stream, I->d_triangles.ptr, static_tris, I->d_tri_boxes.ptr);
            for (size_t q = 0; q < n_static; q++) patch_boxes(I, I->stream, (uint32_t)q, I->d_tri_boxes.ptr + I->mesh_records[q].tri_base);
            const hipError_t fe = sah_build_forest(I->stream, I->d_tri_boxes.ptr, static_tris, I->d_forest.ptr, (uint32_t)n_static, max_n, I->d_sah_ws.ptr, I->d_sah_ws.cap,
                                                   I->d_blas_raw.ptr, I->d_blas_order.ptr, I->d_mesh_node_counts.ptr, I->sah_max_leaf, I->sah_trav_cost);
            if (fe == hipSuccess) {
                if (I->build_from_heads) HIP_TRY(I, hipStreamWaitEvent(I->stream, I->ev_records, 0));
                launch_make_packets(I->stream, I->d_triangles.ptr, I->d_blas_order.ptr, static_tris, 0u, I->d_packets.ptr); // global positions: one launch
                for (size_t q = 0; q < n_static; q++) resolve_duplicates(I, I->stream, (uint32_t)q);
                launch_forest_relative_order(I->stream, I->d_blas_order.ptr, static_tris, I->d_forest.ptr, (uint32_t)n_static);
                forest_done = true;
            } else if (fe != hipErrorInvalidValue) {
                HIP_TRY(I, fe);
            } // else: some tree is deeper than the builder's level budget: mesh by mesh, where LBVH can take over for that one
        }
        if (!forest_done) {
            rc = build_meshes(I, all, false);
            trace("trees queued");
            if (I->build_from_heads) { // the packets of every mesh, now that the records are (about to be) there
                I->build_from_heads = false;
                HIP_TRY(I, hipStreamWaitEvent(I->stream, I->ev_records, 0));
                if (rc == RFW_HIP_OK)
                    for (size_t q = 0; q < n_static; q++)
                        if (I->mesh_records[q].tri_count) record_packets(I, I->stream, (uint32_t)q);
            }
            if (rc) return rc;
        }
        I->build_from_heads = false;
    }
    if (env_switches().node_order && n_static) {
        // EXPERIMENT (round 6): the builders number the 4-wide nodes in arrival order; renumber every static tree on the host —
        // 1: depth-first (a node, then the subtree of its first child, ...), 2: treelets of up to 32 nodes (breadth-first inside a treelet,
        // treelets depth-first) — and measure what the caches make of it beyond their reach (bench.py --workload atrium32m)
        HIP_TRY(I, hipStreamSynchronize(I->stream));
        std::vector<uint32_t> cnt(n_static, 0u);
        HIP_TRY(I, hipMemcpy(cnt.data(), I->d_mesh_node_counts.ptr, n_static * 4, hipMemcpyDeviceToHost));
        for (size_t q = 0; q < n_static; q++) {
            const uint32_t n = cnt[q];
            if (n < 2) continue;
            Node4* dev = I->d_blas_raw.ptr + (I->mesh_records[q].node_base - I->raw_node_origin);
            std::vector<Node4> in(n), out(n);
            HIP_TRY(I, hipMemcpy(in.data(), dev, (size_t)n * sizeof(Node4), hipMemcpyDeviceToHost));
            std::vector<uint32_t> order; // order[new] = old
            order.reserve(n);
            auto interior = [&](uint32_t c) { return c != kInvalidRef && !(c & kLeafBit); };
            if (env_switches().node_order == 1) {
                std::vector<uint32_t> stack{0u};
                while (!stack.empty()) {
                    const uint32_t v = stack.back(); stack.pop_back();
                    order.push_back(v);
                    for (int k = 3; k >= 0; k--) if (interior(in[v].child[k])) stack.push_back(in[v].child[k]);
                }
            } else {
                std::vector<uint32_t> roots{0u}; // treelet roots, depth-first (a stack)
                while (!roots.empty()) {
                    const uint32_t r = roots.back(); roots.pop_back();
                    std::vector<uint32_t> fifo{r};
                    size_t head = 0;
                    std::vector<uint32_t> spill;
                    while (head < fifo.size()) {
                        const uint32_t v = fifo[head++];
                        order.push_back(v);
                        for (int k = 0; k < 4; k++) {
                            const uint32_t c = in[v].child[k];
                            if (!interior(c)) continue;
                            if (fifo.size() < 32) fifo.push_back(c); else spill.push_back(c);
                        }
                    }
                    for (size_t k = spill.size(); k-- > 0;) roots.push_back(spill[k]);
                }
            }
            if (order.size() != n) return fail(I, RFW_HIP_E_STATE, "RFW_NODE_ORDER: the tree does not reach every node of its region");
            std::vector<uint32_t> where(n);
            for (uint32_t k = 0; k < n; k++) where[order[k]] = k;
            for (uint32_t k = 0; k < n; k++) {
                out[k] = in[order[k]];
                for (int c = 0; c < 4; c++) if (interior(out[k].child[c])) out[k].child[c] = where[out[k].child[c]];
            }
            HIP_TRY(I, hipMemcpy(dev, out.data(), (size_t)n * sizeof(Node4), hipMemcpyHostToDevice));
        }
    }
    if ((rc = upload(I, I->d_mesh_records, I->mesh_records.data(), I->mesh_records.size()))) return rc;
    // all static regions in one launch; the slots behind a tree's last node are skipped (the builders left the node counts on the device)
    launch_quantize_regions(I->stream, I->d_blas_raw.ptr, I->d_blas_nodes.ptr, copies_of(I->d_blas_wide, I->d_blas_oct), static_nodes, I->d_mesh_records.ptr,
                            I->d_mesh_node_counts.ptr, (uint32_t)n_static);
    HIP_TRY(I, hipGetLastError());
    I->n_tris = tri_total;
    I->n_split_refs = 0;
    for (size_t q = 0; q < n_static; q++) I->n_split_refs += I->mesh_records[q].tri_count - I->record_tri_orig[q];
    HIP_TRY(I, hipEventRecord(I->ev_build[2], I->stream));
    trace("all queued");
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    I->records_pending = false;
    trace("device done");
    if (kTrace && I->records_timed) {
        float a = 0, b = 0, c = 0;
        (void)hipEventElapsedTime(&c, I->ev_build[0], I->ev_heads);
        fprintf(stderr, "[build] by events: heads arrived %.3f ms\n", c);
        (void)hipEventElapsedTime(&a, I->ev_build[0], I->ev_records);
        (void)hipEventElapsedTime(&b, I->ev_build[0], I->ev_build[2]);
        fprintf(stderr, "[build] by events: records arrived %.3f ms, build done %.3f ms after the start\n", a, b);
    }
    I->build_events_pending = true;
    I->blas_upload_bytes = (uint64_t)static_tris * sizeof(rfw_rt_triangle);
    I->blas_kernel_bytes = kernel_bytes;
    // nodes actually in use (the regions are sized for the worst case, one node per primitive); skinned copies count at their worst case
    std::vector<uint32_t> counts(n_static, 0u);
    if (n_static) HIP_TRY(I, hipMemcpy(counts.data(), I->d_mesh_node_counts.ptr, n_static * 4, hipMemcpyDeviceToHost));
    I->n_blas_nodes = node_total - static_nodes;
    for (uint32_t c : counts) I->n_blas_nodes += c;
    I->node_counts_stale = false;
    // ~350 B per triangle of build scratch: kept between scene changes up to 1 GB.  Not for the allocation's own cost: a hipFree / hipMalloc of
    // 0.4 GB between two builds made the NEXT build's record upload crawl beside its kernels (measured, C4: records arrived after 7.5 ms instead
    // of 4.4, every warm build 7.9 ms instead of 5.6 — tools/probes/h2d_overlap.cpp shows the copy itself overlaps kernels at full rate)
    if (I->d_sah_ws.cap > (size_t(1) << 30)) I->d_sah_ws.release();
    for (auto& L : I->lanes)
        if (L.ws.cap > (size_t(1) << 30)) { L.ws.release(); L.boxes.release(); }
    I->tri_end = static_tris;
    I->node_end = static_nodes;
    I->hole_tris = 0;
    I->layout_valid = I->derived.empty(); // the incremental path does not move skinned copies around
    I->full_builds++;
    return RFW_HIP_OK;
}

// Only the meshes that changed (gpu-rt/src/lib.rs:1345-1383 rebuilds / refits `mesh.dirty` ones only): a changed mesh keeps its region of the
// mega-buffers when it still fits and gets a new one behind the others when it grew; a new mesh is appended; an unloaded mesh leaves a hole.
// Returns 1 when the layout has to be redone by a full build (too many holes), 0 on success, < 0 on error.
int build_blas_device_incremental(Instance* I)
{
    I->build_from_heads = false;
    I->records_timed = false;
    // what went away
    for (auto it = I->mesh_index.begin(); it != I->mesh_index.end();) {
        if (I->meshes.find(it->first) == I->meshes.end()) {
            I->hole_tris += I->record_tri_cap[it->second];
            I->mesh_records[it->second].tri_count = 0;
            it = I->mesh_index.erase(it);
        } else ++it;
    }
    std::vector<uint32_t> todo; // record indices to (re)build
    uint32_t max_n = 0;
    for (auto& kv : I->meshes) {
        MeshHost& m = kv.second;
        if (!m.dirty) continue;
        const uint32_t n = (uint32_t)m.n_refs;
        uint32_t q;
        const auto it = I->mesh_index.find(kv.first);
        if (it != I->mesh_index.end() && n <= I->record_tri_cap[it->second]) {
            q = it->second; // rebuilt in place
        } else {
            if (it != I->mesh_index.end()) { // grew: the old region becomes a hole, the record moves behind the others
                q = it->second;
                I->hole_tris += I->record_tri_cap[q];
            } else {
                q = (uint32_t)I->mesh_records.size();
                MeshRecord r;
                std::memset(&r, 0, sizeof(r));
                I->mesh_records.push_back(r);
                I->record_tri_cap.push_back(0);
                I->record_tri_orig.push_back(0);
                I->mesh_index[kv.first] = q;
            }
            if ((uint64_t)I->tri_end + n > kLeafFirstMask || (uint64_t)I->node_end + std::max(n, 1u) > 0x7fffffffu) return 1;
            I->mesh_records[q].tri_base = I->tri_end;
            I->mesh_records[q].node_base = I->node_end;
            I->record_tri_cap[q] = n;
            I->tri_end += n;
            I->node_end += std::max(n, 1u);
        }
        I->mesh_records[q].tri_count = n;
        I->record_tri_orig[q] = (uint32_t)m.n_orig;
        I->mesh_records[q].node_count = std::max(I->record_tri_cap[q], 1u);
        todo.push_back(q);
        max_n = std::max(max_n, n);
        m.dirty = false;
    }
    if (I->hole_tris > std::max<uint64_t>(I->tri_end / 2, 1u << 16)) return 1; // mostly holes: compact by a full build
    assign_logical_ids(I);
    HIP_TRY(I, I->d_triangles.grow_keep(I->tri_end, I->d_triangles.cap, I->stream));
    HIP_TRY(I, I->d_packets.grow_keep(I->tri_end, I->d_packets.cap, I->stream));
    HIP_TRY(I, I->d_blas_order.grow_keep(I->tri_end, I->d_blas_order.cap, I->stream));
    {
        const size_t nodes_before = I->d_blas_nodes.cap; // (the regions in use end below the old capacity)
        HIP_TRY(I, I->d_blas_nodes.grow_keep(I->node_end, I->d_blas_nodes.cap, I->stream));
        HIP_TRY(I, follow_copies(I->d_blas_wide, I->d_blas_oct, I->d_blas_nodes, nodes_before, I->stream, blas_wide_wanted(I, I->tri_end)));
    }
    HIP_TRY(I, I->d_blas_raw.grow_keep(I->node_end, 0, I->stream)); // build output only: nothing to keep
    HIP_TRY(I, I->d_mesh_node_counts.grow_keep(std::max<size_t>(I->mesh_records.size(), 1), I->d_mesh_node_counts.cap, I->stream));
    HIP_TRY(I, I->d_mesh_records.grow_keep(std::max<size_t>(I->mesh_records.size(), 1), 0, I->stream));
    HIP_TRY(I, I->d_tri_boxes.ensure(std::max(max_n, 1u)));
    int rc;
    if ((rc = ensure_lbvh_ws(I, max_n))) return rc;
    for (auto& ev : I->ev_build)
        if (!ev) HIP_TRY(I, hipEventCreate(&ev));
    HIP_TRY(I, hipEventRecord(I->ev_build[0], I->stream));
    if ((rc = upload_split_pieces(I, todo))) return rc;
    uint64_t upload_bytes = 0, kernel_bytes = 0;
    // ONE large registered mesh changed (a deforming mesh that is re-sent every frame): heads first, as in a full build — the tree is built
    // from the 48-B heads while the records follow on the second stream, the packets are made when they are there.  Nothing is synchronised
    // here: the next set_3d_mesh waits for ev_records before it overwrites the host copy the upload reads (records_pending).
    bool heads_first = false;
    if (todo.size() == 1 && I->mesh_records[todo[0]].tri_count * sizeof(rfw_rt_triangle) >= (size_t(1) << 20)) {
        const uint32_t q = todo[0];
        MeshHost* mh = nullptr;
        for (auto& kv : I->mesh_index)
            if (kv.second == q) mh = &I->meshes[kv.first];
        if (mh && mh->pinned) {
            const MeshRecord& r = I->mesh_records[q];
            HIP_TRY(I, I->d_heads.ensure(I->tri_end));
            if (!I->records_stream) HIP_TRY(I, hipStreamCreateWithFlags(&I->records_stream, hipStreamNonBlocking));
            if (!I->ev_heads) HIP_TRY(I, hipEventCreate(&I->ev_heads));
            if (!I->ev_records) HIP_TRY(I, hipEventCreate(&I->ev_records));
            HIP_TRY(I, hipMemcpyAsync(I->d_heads.ptr + r.tri_base, mh->heads.data(), (size_t)r.tri_count * sizeof(TriHead), hipMemcpyHostToDevice, I->stream));
            HIP_TRY(I, hipEventRecord(I->ev_heads, I->stream));
            HIP_TRY(I, hipStreamWaitEvent(I->records_stream, I->ev_heads, 0));
            HIP_TRY(I, hipMemcpyAsync(I->d_triangles.ptr + r.tri_base, mh->tris.data(), (size_t)r.tri_count * sizeof(rfw_rt_triangle), hipMemcpyHostToDevice, I->records_stream));
            HIP_TRY(I, hipEventRecord(I->ev_records, I->records_stream));
            I->records_pending = true;
            I->records_timed = true;
            I->heads_first_builds++;
            upload_bytes += (uint64_t)r.tri_count * sizeof(rfw_rt_triangle);
            heads_first = true;
        }
    }
    for (const uint32_t q : todo) { // the changed meshes' triangles first (their regions are disjoint) ...
        if (heads_first) break;
        const MeshRecord& r = I->mesh_records[q];
        const MeshHost* mh = nullptr;
        for (auto& kv : I->mesh_index)
            if (kv.second == q) mh = &I->meshes[kv.first];
        if (r.tri_count && mh) {
            upload_bytes += (uint64_t)r.tri_count * sizeof(rfw_rt_triangle);
            // through the pinned ring when small (the host copy may be replaced by the next set_3d_mesh before a pageable copy has run)
            if ((size_t)r.tri_count * sizeof(rfw_rt_triangle) <= (8u << 20))
                HIP_TRY(I, I->pins.upload(I->d_triangles.ptr + r.tri_base, mh->tris.data(), (size_t)r.tri_count * sizeof(rfw_rt_triangle), I->stream));
            else {
                HIP_TRY(I, hipMemcpyAsync(I->d_triangles.ptr + r.tri_base, mh->tris.data(), (size_t)r.tri_count * sizeof(rfw_rt_triangle), hipMemcpyHostToDevice, I->stream));
                HIP_TRY(I, hipStreamSynchronize(I->stream));
            }
        }
    }
    HIP_TRY(I, hipEventRecord(I->ev_build[1], I->stream));
    for (const uint32_t q : todo) kernel_bytes += build_pass_bytes(I->mesh_records[q].tri_count);
    I->build_from_heads = heads_first;
    rc = build_meshes(I, todo, true); // ... then their trees
    I->build_from_heads = false;
    if (rc) return rc;
    if (heads_first) {
        const MeshRecord& r = I->mesh_records[todo[0]];
        HIP_TRY(I, hipStreamWaitEvent(I->stream, I->ev_records, 0));
        (void)r;
        record_packets(I, I->stream, todo[0]);
    }
    HIP_TRY(I, hipEventRecord(I->ev_build[2], I->stream));
    I->build_events_pending = true;
    I->blas_upload_bytes = upload_bytes;
    I->blas_kernel_bytes = kernel_bytes;
    HIP_TRY(I, hipGetLastError());
    HIP_TRY(I, I->pins.upload(I->d_mesh_records.ptr, I->mesh_records.data(), I->mesh_records.size() * sizeof(MeshRecord), I->stream));
    uint64_t live = 0;
    for (auto& kv : I->mesh_index) live += I->mesh_records[kv.second].tri_count;
    I->n_tris = live;
    I->n_split_refs = 0;
    for (auto& kv : I->mesh_index) I->n_split_refs += I->mesh_records[kv.second].tri_count - I->record_tri_orig[kv.second];
    I->node_counts_stale = true;
    I->incremental_builds++;
    return RFW_HIP_OK;
}

int build_blas_device(Instance* I)
{
    bool any_dirty = false, removed = false, all_dirty = !I->meshes.empty();
    for (auto& kv : I->meshes) { any_dirty = any_dirty || kv.second.dirty; all_dirty = all_dirty && kv.second.dirty; }
    for (auto& kv : I->mesh_index) removed = removed || I->meshes.find(kv.first) == I->meshes.end();
    // (every mesh changed: nothing to keep — the full build lays the buffers out afresh and builds all meshes in one pass)
    if (I->layout_valid && I->derived.empty() && wanted_derived(I).empty() && (any_dirty || removed) && !(all_dirty && I->meshes.size() >= 2)) {
        const int rc = build_blas_device_incremental(I);
        // build scratch is not kept between scene changes when it is large (as after a full build: ~350 B per triangle); the scratch of small
        // edits stays — hipFree waits for the device, and an edit of one 5120-triangle mesh would pay its own build time on the host for it
        if (I->d_sah_ws.cap > (64u << 20)) I->d_sah_ws.release();
        for (auto& L : I->lanes)
            if (L.ws.cap > (64u << 20)) { L.ws.release(); L.boxes.release(); }
        if (rc < 0) { // an error part-way through: records, capacities and dirty flags may be half-updated — the next synchronize() starts over
            I->layout_valid = false;
            for (auto& kv : I->meshes) kv.second.dirty = true;
            I->meshes_dirty = true;
        }
        if (rc <= 0) return rc;
        for (auto& kv : I->meshes) kv.second.dirty = true; // (only matters for the host builder; the full device build takes every mesh)
    }
    return build_blas_device_full(I);
}

// BLAS on the host cores (binned SAH, multi-threaded), flattened into the mega-buffers (gpu-rt/src/lib.rs:1387-1548)
int build_blas_host(Instance* I)
{
    for (auto& kv : I->meshes)
        if (kv.second.dirty) build_mesh(I, kv.second);
    I->mesh_records.clear();
    I->mesh_index.clear();
    I->record_tri_orig.clear();
    I->n_split_refs = 0;
    I->layout_valid = false;
    std::vector<Node4> nodes;
    std::vector<TriPacket> packets;
    std::vector<rfw_rt_triangle> tris;
    for (auto& kv : I->meshes) {
        MeshHost& m = kv.second;
        MeshRecord r;
        std::memset(&r, 0, sizeof(r));
        r.node_base = (uint32_t)nodes.size();
        r.node_count = (uint32_t)m.bvh.nodes.size();
        r.tri_base = (uint32_t)tris.size();
        r.tri_count = (uint32_t)m.n_refs;
        I->mesh_index[kv.first] = (uint32_t)I->mesh_records.size();
        I->mesh_records.push_back(r);
        I->record_tri_orig.push_back((uint32_t)m.n_orig);
        I->n_split_refs += m.n_refs - m.n_orig;
        nodes.insert(nodes.end(), m.bvh.nodes.begin(), m.bvh.nodes.end());
        const size_t p0 = packets.size();
        packets.insert(packets.end(), m.packets.begin(), m.packets.end());
        for (size_t k = p0; k < packets.size(); k++) packets[k].tri_id += r.tri_base; // global triangle id
        tris.insert(tris.end(), m.tris.begin(), m.tris.begin() + (ptrdiff_t)m.n_refs);
    }
    uint32_t tri_total = (uint32_t)tris.size(), node_total = (uint32_t)nodes.size();
    int rc;
    if ((rc = layout_derived(I, tri_total, node_total))) return rc;
    assign_logical_ids(I);
    I->n_tris = tri_total;
    I->n_blas_nodes = node_total;
    HIP_TRY(I, I->d_blas_nodes.ensure(node_total)); // room for the skinned copies behind the static meshes
    HIP_TRY(I, follow_copies(I->d_blas_wide, I->d_blas_oct, I->d_blas_nodes, 0, I->stream, blas_wide_wanted(I, tri_total)));
    HIP_TRY(I, I->d_packets.ensure(tri_total));
    HIP_TRY(I, I->d_triangles.ensure(tri_total));
    std::vector<Node4Q> qnodes(nodes.size());
    for (size_t k = 0; k < nodes.size(); k++) qnodes[k] = quantize_node(nodes[k]);
    if ((rc = upload(I, I->d_blas_nodes, qnodes.data(), qnodes.size()))) return rc;
    launch_expand_nodes(I->stream, I->d_blas_nodes.ptr, copies_of(I->d_blas_wide, I->d_blas_oct), 0u, (uint32_t)qnodes.size());
    if ((rc = upload(I, I->d_packets, packets.data(), packets.size()))) return rc;
    if ((rc = upload(I, I->d_triangles, tris.data(), tris.size()))) return rc;
    if ((rc = upload(I, I->d_mesh_records, I->mesh_records.data(), I->mesh_records.size()))) return rc;
    HIP_TRY(I, hipStreamSynchronize(I->stream)); // the host vectors above go out of scope
    return RFW_HIP_OK;
}

// instances + TLAS (gpu-rt/src/lib.rs:1576-1615): global instance id = mesh_base[mesh] + slot
int build_instances(Instance* I, Instance* T)
{
    // I: the scene (instance lists, mesh records, skins); T: whose instance-level device buffers, staging blocks and stream are used —
    // I itself, or one of its frame slots (each slot keeps its own TLAS so that a scene whose instances move every frame still pipelines)
    const auto wait0 = std::chrono::steady_clock::now();
    // sizes first, then ONE pinned staging block: [matrices | mesh_of | valid_gids | mesh_local]
    size_t n_all = 0;
    for (auto& kv : I->inst_lists) n_all += kv.second.matrices.size();
    const size_t n_mesh = I->mesh_records.size();
    const size_t off_mats = 0, off_meshof = off_mats + n_all * sizeof(rfw_mat4), off_valid = off_meshof + n_all * 4,
                 off_local = (off_valid + n_all * 4 + 63) / 64 * 64, off_joints = off_local + std::max<size_t>(n_mesh, 1) * sizeof(DevBox);
    size_t n_joints = 0;
    std::vector<size_t> joint_off(I->skins.size(), 0);
    if (!I->derived.empty())
        for (size_t k = 0; k < I->skins.size(); k++) { joint_off[k] = n_joints; n_joints += I->skins[k].size(); }
    const size_t total = off_joints + n_joints * sizeof(rfw_mat4);
    if (T->stage_pending[T->stage_next]) { // the upload that last used this block (two synchronizes ago) must have left it
        HIP_TRY(I, hipEventSynchronize(T->stage_event[T->stage_next]));
        T->stage_pending[T->stage_next] = false;
    }
    // back-pressure, not work: a host that runs ahead of the GPU waits here for the copy of two synchronizes ago
    T->ms_stage_wait = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wait0).count();
    int rc;
    if ((rc = ensure_stage(T, total))) return rc;
    char* st = static_cast<char*>(T->stage);
    rfw_mat4* mats = reinterpret_cast<rfw_mat4*>(st + off_mats);
    uint32_t* mesh_of = reinterpret_cast<uint32_t*>(st + off_meshof);
    uint32_t* valid = reinterpret_cast<uint32_t*>(st + off_valid);
    DevBox* local = reinterpret_cast<DevBox*>(st + off_local);
    rfw_mat4* joints = reinterpret_cast<rfw_mat4*>(st + off_joints);
    std::memset(local, 0, std::max<size_t>(n_mesh, 1) * sizeof(DevBox));
    if (n_joints)
        for (size_t k = 0; k < I->skins.size(); k++)
            if (!I->skins[k].empty()) std::memcpy(joints + joint_off[k], I->skins[k].data(), I->skins[k].size() * sizeof(rfw_mat4));
    uint32_t gid = 0, n_valid = 0;
    for (auto& kv : I->inst_lists) {
        const auto mit = I->mesh_index.find(kv.first);
        const bool mesh_ok = mit != I->mesh_index.end() && I->mesh_records[mit->second].tri_count > 0;
        if (mesh_ok) {
            DevBox& lb = local[mit->second];
            for (int a = 0; a < 3; a++) { lb.lo[a] = kv.second.local_aabb.min[a]; lb.hi[a] = kv.second.local_aabb.max[a]; }
        }
        const size_t cnt = kv.second.matrices.size();
        if (cnt) std::memcpy(mats + gid, kv.second.matrices.data(), cnt * sizeof(rfw_mat4));
        for (size_t s = 0; s < cnt; s++, gid++) {
            mesh_of[gid] = mesh_ok ? mit->second : 0xffffffffu;
            const int32_t sk = s < kv.second.skin_ids.size() ? kv.second.skin_ids[s] : -1;
            if (mesh_ok && sk >= 0 && !I->derived.empty()) { // skinned slot: its own record (geometry, BVH, bounds)
                const auto dit = I->derived.find(std::make_pair(kv.first, sk));
                if (dit != I->derived.end()) mesh_of[gid] = dit->second.record;
            }
            if (mesh_ok && !is_zero_matrix(kv.second.matrices[s])) valid[n_valid++] = gid; // zero matrix = removed slot (instances_3d.rs:79-86)
        }
    }
    T->n_instances = n_all;
    T->n_valid_instances = n_valid;
    HIP_TRY(I, T->d_matrices.ensure(n_all));
    HIP_TRY(I, T->d_mesh_of_instance.ensure(n_all));
    HIP_TRY(I, T->d_valid_gids.ensure(n_all));
    HIP_TRY(I, T->d_mesh_local.ensure(std::max<size_t>(n_mesh, 1)));
    HIP_TRY(I, T->d_xforms.ensure(n_all));
    HIP_TRY(I, T->d_normals.ensure(n_all));
    HIP_TRY(I, T->d_tlas_prims.ensure(n_all));
    HIP_TRY(I, T->d_tlas_nodes.ensure(std::max<size_t>(n_valid, 1)));
    HIP_TRY(I, follow_copies(T->d_tlas_wide, T->d_tlas_oct, T->d_tlas_nodes, 0, T->stream, true));
    HIP_TRY(I, T->d_tlas_raw.ensure(std::max<size_t>(n_valid, 1)));
    HIP_TRY(I, T->d_node_count.ensure(1));
    hipStream_t s = T->stream;
    // Up to kTlasFusedMax live instances and no skinned copies (BASELINE config 3): the staging block goes up as ONE copy, the tree is built
    // by ONE workgroup (lbvh.hip, k_tlas_fused), nodes are quantised / copied per octant and the instance descriptors made by one more launch —
    // 3 API calls where the chain below takes 27, and the host thread that issues them was what bound that configuration
    const bool fused = I->tlas_fused == 1 || (I->tlas_fused == 2 && !I->slots.empty()); // (api_internal.h: throughput with frames in flight, latency without)
    if (I->tlas_on_device && fused && I->derived.empty() && n_valid >= 2 && n_valid <= kTlasFusedMax) {
        HIP_TRY(I, T->d_stage_dev.ensure(total));
        HIP_TRY(I, T->d_inst_boxes.ensure(n_valid));
        if ((rc = ensure_lbvh_ws(T, n_valid))) return rc;
        HIP_TRY(I, hipMemcpyAsync(T->d_stage_dev.ptr, st, off_joints, hipMemcpyHostToDevice, s));
        const rfw_mat4* d_mats = reinterpret_cast<const rfw_mat4*>(T->d_stage_dev.ptr + off_mats);
        const uint32_t* d_mesh_of = reinterpret_cast<const uint32_t*>(T->d_stage_dev.ptr + off_meshof);
        const uint32_t* d_valid = reinterpret_cast<const uint32_t*>(T->d_stage_dev.ptr + off_valid);
        const DevBox* d_local = reinterpret_cast<const DevBox*>(T->d_stage_dev.ptr + off_local);
        HIP_TRY(I, tlas_build_fused(s, d_mats, d_mesh_of, d_local, d_valid, n_valid, T->d_lbvh_ws.ptr, T->d_lbvh_ws.cap, T->d_inst_boxes.ptr, T->d_tlas_raw.ptr,
                                    T->d_tlas_prims.ptr, T->d_node_count.ptr));
        launch_tlas_finish(s, T->d_tlas_raw.ptr, T->d_tlas_nodes.ptr, copies_of(T->d_tlas_wide, T->d_tlas_oct), n_valid, T->d_node_count.ptr, d_mats, d_mesh_of,
                           I->d_mesh_records.ptr, (uint32_t)n_all, T->d_xforms.ptr, T->d_normals.ptr);
        HIP_TRY(I, hipGetLastError());
        T->n_tlas_nodes = 0; // read back lazily (get_scene_stats)
        HIP_TRY(I, hipEventRecord(T->stage_event[T->stage_next], s));
        T->stage_pending[T->stage_next] = true;
        T->stage_next = (T->stage_next + 1) % Instance::kStages;
        I->tlas_fused_builds++;
        return RFW_HIP_OK;
    }
    if (n_all) {
        HIP_TRY(I, hipMemcpyAsync(T->d_matrices.ptr, mats, n_all * sizeof(rfw_mat4), hipMemcpyHostToDevice, s));
        HIP_TRY(I, hipMemcpyAsync(T->d_mesh_of_instance.ptr, mesh_of, n_all * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(I, hipMemcpyAsync(T->d_valid_gids.ptr, valid, n_all * 4, hipMemcpyHostToDevice, s));
    }
    HIP_TRY(I, hipMemcpyAsync(T->d_mesh_local.ptr, local, std::max<size_t>(n_mesh, 1) * sizeof(DevBox), hipMemcpyHostToDevice, s));
    if (!I->derived.empty()) {
        // skinned copies (structs.rs:820-877) and their BLAS, every synchronize, all on-stream: skin -> refit of the tree built over the
        // first pose (gpu-rt: refit_bvh, lib.rs:1350-1352) -> packets -> bounds; with builder = DEVICE_LBVH: skin -> boxes -> LBVH rebuild
        HIP_TRY(I, I->d_joints.ensure(n_joints));
        HIP_TRY(I, hipMemcpyAsync(I->d_joints.ptr, joints, n_joints * sizeof(rfw_mat4), hipMemcpyHostToDevice, s));
        if ((rc = ensure_lbvh_ws(T, std::max<uint32_t>(I->max_derived_tris, n_valid)))) return rc;
        const bool refit = I->builder != RFW_HIP_BUILDER_DEVICE_LBVH; // DEVICE_LBVH keeps the rebuild-every-frame path
        if (refit) {
            const size_t raw_nodes = I->d_blas_raw.cap;
            HIP_TRY(I, I->d_refit_parent.ensure(raw_nodes));
            HIP_TRY(I, I->d_refit_nint.ensure(raw_nodes));
            HIP_TRY(I, I->d_refit_arrive.ensure(raw_nodes));
        }
        for (auto& kv : I->derived) {
            DerivedMesh& d = kv.second;
            const MeshRecord& r = I->mesh_records[d.record];
            const MeshRecord& src = I->mesh_records[d.src_record];
            rfw_rt_triangle* tris = I->d_triangles.ptr + r.tri_base;
            const size_t raw_off = r.node_base - I->raw_node_origin;
            Node4* raw = I->d_blas_raw.ptr + raw_off;
            uint32_t* order = I->d_blas_order.ptr + r.tri_base;
            launch_skin_triangles(s, I->d_triangles.ptr + src.tri_base, I->d_skin_data.ptr + d.skin_offset, I->d_joints.ptr + joint_off[kv.first.second],
                                  (uint32_t)I->skins[kv.first.second].size(), r.tri_count, tris);
            uint32_t quantise_count = r.node_count;
            if (!refit) {
                launch_triangle_boxes(s, tris, r.tri_count, I->d_tri_boxes.ptr);
                HIP_TRY(I, lbvh_build(s, I->d_tri_boxes.ptr, r.tri_count, T->d_lbvh_ws.ptr, T->d_lbvh_ws.cap, raw, order, nullptr));
            } else if (!d.topology_built) {
                // first pose of this (mesh, skin) pair: the tree, by binned SAH (blocking, once), and what a refit needs to climb it
                launch_triangle_boxes(s, tris, r.tri_count, I->d_tri_boxes.ptr);
                HIP_TRY(I, I->d_sah_ws.ensure(sah_workspace_bytes(r.tri_count)));
                HIP_TRY(I, T->d_node_count.ensure(1));
                const hipError_t se = sah_build(s, I->d_tri_boxes.ptr, r.tri_count, I->d_sah_ws.ptr, I->d_sah_ws.cap, raw, order, T->d_node_count.ptr, I->sah_max_leaf,
                                                I->sah_trav_cost);
                if (se == hipErrorInvalidValue) { // deeper than the SAH builder's level budget: LBVH always terminates (as for static meshes)
                    HIP_TRY(I, lbvh_build(s, I->d_tri_boxes.ptr, r.tri_count, T->d_lbvh_ws.ptr, T->d_lbvh_ws.cap, raw, order, T->d_node_count.ptr));
                } else {
                    HIP_TRY(I, se);
                }
                HIP_TRY(I, hipMemcpyAsync(&d.node_count, T->d_node_count.ptr, 4, hipMemcpyDeviceToHost, s));
                HIP_TRY(I, hipStreamSynchronize(s));
                if (d.node_count == 0 || d.node_count > r.node_count) return fail(I, RFW_HIP_E_STATE, "skinned BLAS: node count out of range");
                launch_refit_setup(s, raw, d.node_count, I->d_refit_parent.ptr + raw_off, I->d_refit_nint.ptr + raw_off);
                d.topology_built = true;
                quantise_count = d.node_count;
            } else {
                launch_refit(s, raw, d.node_count, tris, order, I->d_refit_parent.ptr + raw_off, I->d_refit_nint.ptr + raw_off, I->d_refit_arrive.ptr + raw_off);
                quantise_count = d.node_count;
            }
            launch_make_packets(s, tris, order, r.tri_count, r.tri_base, I->d_packets.ptr + r.tri_base);
            launch_quantize_nodes(s, raw, I->d_blas_nodes.ptr + r.node_base, copies_of(I->d_blas_wide, I->d_blas_oct), r.node_base, quantise_count);
            launch_mesh_bounds(s, tris, r.tri_count, I->d_bounds_scratch.ptr, T->d_mesh_local.ptr + d.record);
        }
        HIP_TRY(I, hipGetLastError());
        if (!I->tlas_on_device) { // the host TLAS needs the deformed bounds
            HIP_TRY(I, hipStreamSynchronize(s));
            for (const auto& kv : I->derived)
                HIP_TRY(I, hipMemcpy(local + kv.second.record, T->d_mesh_local.ptr + kv.second.record, sizeof(DevBox), hipMemcpyDeviceToHost));
        }
    }
    launch_prepare_instances(s, T->d_matrices.ptr, T->d_mesh_of_instance.ptr, I->d_mesh_records.ptr, (uint32_t)n_all, T->d_xforms.ptr, T->d_normals.ptr);
    if (I->tlas_on_device) {
        HIP_TRY(I, T->d_inst_boxes.ensure(std::max<size_t>(n_valid, 1)));
        HIP_TRY(I, T->d_tlas_order.ensure(std::max<size_t>(n_valid, 1)));
        if ((rc = ensure_lbvh_ws(T, n_valid))) return rc;
        launch_instance_boxes(s, T->d_matrices.ptr, T->d_mesh_of_instance.ptr, T->d_mesh_local.ptr, T->d_valid_gids.ptr, n_valid, T->d_inst_boxes.ptr);
        HIP_TRY(I, lbvh_build(s, T->d_inst_boxes.ptr, n_valid, T->d_lbvh_ws.ptr, T->d_lbvh_ws.cap, T->d_tlas_raw.ptr, T->d_tlas_order.ptr,
                              T->d_node_count.ptr));
        launch_quantize_nodes(s, T->d_tlas_raw.ptr, T->d_tlas_nodes.ptr, copies_of(T->d_tlas_wide, T->d_tlas_oct), 0u, std::max<uint32_t>(n_valid, 1u), T->d_node_count.ptr);
        launch_gather_u32(s, T->d_valid_gids.ptr, T->d_tlas_order.ptr, n_valid, T->d_tlas_prims.ptr);
        HIP_TRY(I, hipGetLastError());
        T->n_tlas_nodes = 0; // read back lazily (get_scene_stats)
        HIP_TRY(I, hipEventRecord(T->stage_event[T->stage_next], s));
        T->stage_pending[T->stage_next] = true;
        T->stage_next = (T->stage_next + 1) % Instance::kStages;
    } else {
        // host TLAS (builder = HOST_SAH): boxes on the host, binned SAH, upload
        std::vector<PrimBox> boxes(n_valid);
        for (uint32_t k = 0; k < n_valid; k++) {
            const rfw_mat4& m = mats[valid[k]];
            const DevBox& lb = local[mesh_of[valid[k]]];
            PrimBox b;
            for (int a = 0; a < 3; a++) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            for (int c = 0; c < 8; c++) {
                const float x = (c & 1) ? lb.hi[0] : lb.lo[0], y = (c & 2) ? lb.hi[1] : lb.lo[1], z = (c & 4) ? lb.hi[2] : lb.lo[2];
                const float w[3] = {m.m[0] * x + m.m[4] * y + m.m[8] * z + m.m[12], m.m[1] * x + m.m[5] * y + m.m[9] * z + m.m[13],
                                    m.m[2] * x + m.m[6] * y + m.m[10] * z + m.m[14]};
                for (int a = 0; a < 3; a++) { b.lo[a] = std::min(b.lo[a], w[a]); b.hi[a] = std::max(b.hi[a], w[a]); }
            }
            for (int a = 0; a < 3; a++) {
                const float ext = b.hi[a] - b.lo[a];
                const float e = 2e-4f + 1e-5f * std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])) + 1e-5f * ext;
                b.lo[a] -= e;
                b.hi[a] += e;
            }
            boxes[k] = b;
        }
        HostBvh4 tlas;
        build_bvh4_host(boxes, 1, I->build_threads, tlas);
        std::vector<uint32_t> prims(tlas.prim_order.size());
        for (size_t k = 0; k < prims.size(); k++) prims[k] = valid[tlas.prim_order[k]];
        T->n_tlas_nodes = tlas.nodes.size();
        std::vector<Node4Q> qn(tlas.nodes.size());
        for (size_t k = 0; k < qn.size(); k++) qn[k] = quantize_node(tlas.nodes[k]);
        // on the stream of the instance whose TLAS this is (a frame slot builds its own): upload() copies on the OWNER's stream, and the expansion
        // below — on the slot's — read nodes that had not arrived yet (tests/soak_gpu.py, round 4: host builder x frame slots, in a process
        // whose earlier backends had left other bytes in the recycled allocation)
        HIP_TRY(I, T->d_tlas_nodes.ensure(qn.size()));
        if (!qn.empty()) HIP_TRY(I, hipMemcpyAsync(T->d_tlas_nodes.ptr, qn.data(), qn.size() * sizeof(Node4Q), hipMemcpyHostToDevice, s));
        HIP_TRY(I, follow_copies(T->d_tlas_wide, T->d_tlas_oct, T->d_tlas_nodes, 0, s, true));
        launch_expand_nodes(s, T->d_tlas_nodes.ptr, copies_of(T->d_tlas_wide, T->d_tlas_oct), 0u, (uint32_t)qn.size());
        HIP_TRY(I, T->d_tlas_prims.ensure(prims.size()));
        if (!prims.empty()) HIP_TRY(I, hipMemcpyAsync(T->d_tlas_prims.ptr, prims.data(), prims.size() * 4, hipMemcpyHostToDevice, s));
        HIP_TRY(I, hipGetLastError());
        HIP_TRY(I, hipStreamSynchronize(s));
    }
    return RFW_HIP_OK;
}

// one table of the new version: the old version copied on the device, then the changed elements (runs of consecutive indices) from
// the host copy through the pinned ring; everything from the host when the table was handed over whole or its length changed
template <typename T> int write_table(Instance* I, DevBuf<T>& dst, size_t& dst_n, const DevBuf<T>& old, size_t old_n, const std::vector<T>& host, const Instance::Dirty& d)
{
    const size_t n = host.size();
    HIP_TRY(I, dst.ensure(std::max<size_t>(n, 1)));
    dst_n = n;
    hipStream_t s = I->upload_stream;
    const bool partial = d.any && !d.all && old.ptr && old_n == n;
    if ((!d.any || partial) && old.ptr && old_n == n && n) HIP_TRY(I, hipMemcpyAsync(dst.ptr, old.ptr, n * sizeof(T), hipMemcpyDeviceToDevice, s));
    if (!d.any && old_n == n) return RFW_HIP_OK; // unchanged table: the copy is all
    if (!partial) {
        if (n) HIP_TRY(I, I->pins.upload(dst.ptr, host.data(), n * sizeof(T), s));
        return RFW_HIP_OK;
    }
    std::vector<uint32_t> idx = d.idx;
    std::sort(idx.begin(), idx.end());
    {   // many scattered elements (every other material of thousands, say): one copy of the whole table beats a pinned block and a copy per run
        size_t runs = 0;
        for (size_t a = 0; a < idx.size(); a++) runs += (a == 0 || idx[a] > idx[a - 1] + 1) ? 1 : 0;
        if (runs > 16 || idx.size() * 4 > n) {
            HIP_TRY(I, I->pins.upload(dst.ptr, host.data(), n * sizeof(T), s));
            return RFW_HIP_OK;
        }
    }
    for (size_t a = 0; a < idx.size();) {
        size_t b = a + 1;
        while (b < idx.size() && idx[b] <= idx[b - 1] + 1) b++;
        const size_t lo = idx[a], hi = std::min<size_t>((size_t)idx[b - 1] + 1, n);
        if (lo < hi) HIP_TRY(I, I->pins.upload(dst.ptr + lo, host.data() + lo, (hi - lo) * sizeof(T), s));
        a = b;
    }
    return RFW_HIP_OK;
}

int upload_tables(Instance* I)
{
    if (!I->upload_stream) HIP_TRY(I, hipStreamCreateWithFlags(&I->upload_stream, hipStreamNonBlocking));
    if (!I->tables_ready) HIP_TRY(I, hipEventCreateWithFlags(&I->tables_ready, hipEventDisableTiming));
    const uint64_t nv = I->tables_version + 1;
    Instance::Tables& dst = I->tables[nv % Instance::kTableVersions];
    const Instance::Tables& old = I->tables[I->tables_version % Instance::kTableVersions];
    // the buffer being recycled last held version nv - kTableVersions: a frame still reading it (possible only when more than
    // kTableVersions - 1 edits were synchronized since that frame was issued) has to finish first — a dependency of the UPLOAD on that
    // frame, on the device; the host does not wait
    if (nv >= (uint64_t)Instance::kTableVersions) {
        const uint64_t stale = nv - Instance::kTableVersions;
        // frame_done is recorded behind a slot's LATEST frame and a stream runs in order: waiting for it covers every earlier frame of the slot
        auto order_behind = [&](Instance* c) -> int {
            if (c->tables_oldest_pending > stale) return RFW_HIP_OK; // no frame of this slot that may still run reads the buffer
            if (c->frame_done && !I->slots.empty()) HIP_TRY(I, hipStreamWaitEvent(I->upload_stream, c->frame_done, 0));
            else HIP_TRY(I, hipStreamSynchronize(c->stream)); // no frame_done event without slots
            c->tables_oldest_pending = ~0ull;
            return RFW_HIP_OK;
        };
        int orc;
        if ((orc = order_behind(I))) return orc;
        for (Instance* c : I->slots)
            if ((orc = order_behind(c))) return orc;
    }
    int rc;
    if ((rc = write_table(I, dst.materials, dst.n_mat, old.materials, old.n_mat, I->materials, I->mat_dirty))) return rc;
    if ((rc = write_table(I, dst.area, dst.n_area, old.area, old.n_area, I->area_lights, I->area_dirty))) return rc;
    if ((rc = write_table(I, dst.point, dst.n_point, old.point, old.n_point, I->point_lights, I->point_dirty))) return rc;
    if ((rc = write_table(I, dst.spot, dst.n_spot, old.spot, old.n_spot, I->spot_lights, I->spot_dirty))) return rc;
    if ((rc = write_table(I, dst.dir, dst.n_dir, old.dir, old.n_dir, I->directional_lights, I->dir_dirty))) return rc;
    HIP_TRY(I, hipEventRecord(I->tables_ready, I->upload_stream));
    I->tables_version = nv;
    I->mat_dirty.clear(); I->area_dirty.clear(); I->point_dirty.clear(); I->spot_dirty.clear(); I->dir_dirty.clear();
    return RFW_HIP_OK;
}

int do_synchronize(Instance* I)
{
    HIP_TRY(I, hipSetDevice(I->device));
    bool any_change = false;
    int rc;
    // A new (mesh, skin) pair needs its region of the mega-buffers, i.e. a BLAS rebuild: decided FIRST, so that everything below —
    // which frames to wait for, whether scene_ready is recorded — sees the final dirty flags
    if (!I->meshes_dirty && I->instances_dirty) {
        const auto want = wanted_derived(I);
        bool same = want.size() == I->derived.size();
        if (same)
            for (auto a = want.cbegin(), b = I->derived.cbegin(); a != want.cend(); ++a, ++b)
                if (a->first != b->first) { same = false; break; }
        if (!same) I->meshes_dirty = true;
    }
    // does this call queue work on the owner's stream that the frame slots have to wait for (anything but a per-slot TLAS update)?
    // (per_slot_tlas() may flip inside build_blas_* when skinned copies appear or disappear; meshes_dirty covers both directions)
    // (material and light edits do not count: they go into a new version of their tables, see upload_tables)
    const bool shared_work = I->meshes_dirty || I->textures_dirty || (I->instances_dirty && !per_slot_tlas(I));
    if (!I->slots.empty() && shared_work) {
        // frames still in flight on the slots read the scene that is about to change: the uploads queue behind them
        for (Instance* c : I->slots)
            if (c->frame_done) HIP_TRY(I, hipStreamWaitEvent(I->stream, c->frame_done, 0));
    }
    if (I->meshes_dirty || I->instances_dirty) clear_overflow(I); // new trees: an earlier stack overflow no longer describes the scene
    if (I->meshes_dirty) { // BLAS per changed mesh (gpu-rt/src/lib.rs:1345-1383)
        const auto t0 = std::chrono::steady_clock::now();
        if ((rc = I->blas_on_device ? build_blas_device(I) : build_blas_host(I))) return rc;
        I->ms_blas_build = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
        I->meshes_dirty = false;
        I->instances_dirty = true;
        any_change = true;
    }
    if (I->instances_dirty) {
        const auto t0 = std::chrono::steady_clock::now();
        I->instances_version++;
        if (!per_slot_tlas(I)) { // else: every frame slot rebuilds its own TLAS from the new lists when it renders next
            if ((rc = build_instances(I, I))) return rc;
            I->tlas_version = I->instances_version;
        }
        // host-side work (the device part is asynchronous), without the time spent waiting for the GPU to release a staging block
        I->ms_tlas_build = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count() - I->ms_stage_wait;
        I->instances_dirty = false;
        any_change = true;
    }
    if (I->materials_dirty || I->lights_dirty) {
        if ((rc = upload_tables(I))) return rc;
        I->materials_dirty = false;
        I->lights_dirty = false;
        any_change = true;
    }
    if (I->textures_dirty && !I->tex_layout_dirty && I->tex_offsets.size() == I->textures.size()) {
        // only some textures changed and each keeps its place: their texels go over the old ones (the frame slots were drained above)
        std::sort(I->tex_dirty_idx.begin(), I->tex_dirty_idx.end());
        I->tex_dirty_idx.erase(std::unique(I->tex_dirty_idx.begin(), I->tex_dirty_idx.end()), I->tex_dirty_idx.end());
        for (const uint32_t k : I->tex_dirty_idx) {
            const TexHost& t = I->textures[k];
            if (!t.texels.empty()) HIP_TRY(I, hipMemcpyAsync(I->d_tex_data.ptr + I->tex_offsets[k], t.texels.data(), t.texels.size() * 4, hipMemcpyHostToDevice, I->stream));
        }
        HIP_TRY(I, hipStreamSynchronize(I->stream));
        I->tex_dirty_idx.clear();
        I->textures_dirty = false;
        any_change = true;
    }
    if (I->textures_dirty) { // texels of every texture, then the skybox, in one array + descriptor table
        std::vector<uint32_t> data;
        std::vector<TexDesc> desc(I->textures.size());
        auto put = [&](const TexHost& t) {
            TexDesc d;
            std::memset(&d, 0, sizeof(d));
            d.offset = (uint32_t)data.size();
            d.w = t.w; d.h = t.h; d.mips = t.mips; d.format = t.format;
            data.insert(data.end(), t.texels.begin(), t.texels.end());
            return d;
        };
        I->tex_offsets.resize(I->textures.size());
        for (size_t k = 0; k < I->textures.size(); k++) { desc[k] = put(I->textures[k]); I->tex_offsets[k] = desc[k].offset; }
        I->skybox_desc = put(I->skybox);
        I->tex_layout_dirty = false;
        I->tex_dirty_idx.clear();
        I->n_textures = (uint32_t)desc.size();
        if ((rc = upload(I, I->d_tex_data, data.data(), data.size()))) return rc;
        if ((rc = upload(I, I->d_tex_desc, desc.data(), desc.size()))) return rc;
        HIP_TRY(I, hipStreamSynchronize(I->stream));
        I->textures_dirty = false;
        any_change = true;
    }
    if (any_change) {
        I->sample_count = 0; // the accumulated image no longer matches the scene
        I->scene_version++;
        // recorded only when something was queued here: the owner's stream also carries slot 0's frames, and an event behind them
        // would make every slot wait for slot 0
        if (I->scene_ready && shared_work) HIP_TRY(I, hipEventRecord(I->scene_ready, I->stream));
    }
    I->synchronized = true;
    return RFW_HIP_OK;
}

// frame slots with their own TLAS: (re)build the TLAS and instance descriptors of slot T from the owner's current instance lists
int ensure_slot_tlas(Instance* S, Instance* T)
{
    if (!per_slot_tlas(S) || !S->synchronized || T->tlas_version == S->instances_version) return RFW_HIP_OK;
    if (T != S && S->scene_ready) HIP_TRY(S, hipStreamWaitEvent(T->stream, S->scene_ready, 0)); // the mesh records it reads may still be uploading
    const int rc = build_instances(S, T);
    if (rc == RFW_HIP_OK) T->tlas_version = S->instances_version;
    return rc;
}

} // namespace rfwapi

extern "C" {

int rfw_hip_set_2d_mesh(void* inst, uint32_t, const void*, uint32_t, int32_t) { LOCK(inst); return RFW_HIP_OK; }
int rfw_hip_set_2d_instances(void* inst, uint32_t, const rfw_mat4*, uint32_t) { LOCK(inst); return RFW_HIP_OK; }

// The host copy of a mesh: the records, and their 48-B heads beside them (MeshHost).  Several threads for a large mesh (one thread moves
// ~10 GB/s: 185 MB of C4 took 18 of the 45 ms a re-sent scene cost before anything reached the device).  A mesh of >= 128 KB is registered
// with the runtime the first time it is RE-sent at an unchanged size.
// ---- spatial splits (MeshHost, api_internal.h).  Everything here is host arithmetic on the caller's vertices: deterministic, the same on every rank.
namespace {
struct SplitPoly { int n; float p[10][3]; }; // a triangle clipped by axis planes: at most 3 + 6 vertices ... and one spare
// false: the clipped polygon does not fit the 10 entries (vertices ON the plane are kept by both halves, so repeated cuts can grow a part
// past 3 + 6); the caller then leaves the part uncut — a part with dropped vertices would get a box that does not cover it
inline bool clip_poly(const SplitPoly& in, int axis, float pos, bool keep_below, SplitPoly& out)
{
    out.n = 0;
    for (int i = 0; i < in.n; i++) {
        const float* a = in.p[i];
        const float* b = in.p[(i + 1) % in.n];
        const bool ia = keep_below ? a[axis] <= pos : a[axis] >= pos, ib = keep_below ? b[axis] <= pos : b[axis] >= pos;
        if (ia) {
            if (out.n >= 10) return false;
            std::memcpy(out.p[out.n++], a, 12);
        }
        if (ia != ib) {
            if (out.n >= 10) return false;
            // the same expression in both halves (same endpoints, same order): the two parts share their cut points exactly
            const float t = (pos - a[axis]) / (b[axis] - a[axis]);
            float* q = out.p[out.n++];
            for (int k = 0; k < 3; k++) q[k] = a[k] + t * (b[k] - a[k]);
            q[axis] = pos;
        }
    }
    return true;
}
inline void poly_box(const SplitPoly& p, float lo[3], float hi[3])
{
    for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; }
    for (int i = 0; i < p.n; i++)
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p.p[i][a]); hi[a] = std::max(hi[a], p.p[i][a]); }
}
inline double box_area6(const float lo[3], const float hi[3])
{
    const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
    return 2.0 * (dx * dy + dy * dz + dz * dx);
}
inline double poly_area(const SplitPoly& q)
{
    double ax = 0, ay = 0, az = 0;
    for (int k = 1; k + 1 < q.n; k++) {
        double u[3], w[3];
        for (int a = 0; a < 3; a++) { u[a] = (double)q.p[k][a] - q.p[0][a]; w[a] = (double)q.p[k + 1][a] - q.p[0][a]; }
        ax += u[1] * w[2] - u[2] * w[1]; ay += u[2] * w[0] - u[0] * w[2]; az += u[0] * w[1] - u[1] * w[0];
    }
    return 0.5 * std::sqrt(ax * ax + ay * ay + az * az);
}
// what the box of a triangle wastes: its area minus the area the flattest box of a triangle of this orientation and size must have
// (|n_x| + |n_y| + |n_z| of the edge cross product: Karras & Aila 2013, "Fast parallel construction of high-quality bounding volume hierarchies", §4.2)
inline float triangle_waste(const float* v0, const float* v1, const float* v2, float* ideal_out = nullptr)
{
    float lo[3], hi[3], e1[3], e2[3];
    for (int a = 0; a < 3; a++) {
        lo[a] = std::min(v0[a], std::min(v1[a], v2[a])); hi[a] = std::max(v0[a], std::max(v1[a], v2[a]));
        e1[a] = v1[a] - v0[a]; e2[a] = v2[a] - v0[a];
    }
    const float ideal = std::fabs(e1[1] * e2[2] - e1[2] * e2[1]) + std::fabs(e1[2] * e2[0] - e1[0] * e2[2]) + std::fabs(e1[0] * e2[1] - e1[1] * e2[0]);
    if (ideal_out) *ideal_out = ideal;
    return (float)box_area6(lo, hi) - ideal;
}
struct SplitScan {          // what one copy thread learns about its share of the triangles
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    std::vector<std::pair<float, uint32_t>> top; // min-heap of the kKeep most wasteful (waste, index)
    // A triangle is a candidate when its box wastes more than tau_rel x the area of the MESH's box, which the thread does not know yet — but
    // the box of what it has seen so far is inside it, and a triangle's waste is at most its box's area: a triangle whose box area is below
    // tau_rel x (area of the bounds so far) cannot be one.  `floor_area` is that bound, refreshed every 256 triangles.
    float tau_rel = 0.0f, floor_area = 0.0f;
    uint32_t since_refresh = 0;
};
// Triangles of a mesh that may be split: its kSplitCandidates most wasteful ones (what the splits buy comes from a few hundred).  Every copy
// thread keeps that many of ITS share, so the mesh's most wasteful ones are among the kept ones whatever the number of threads.
constexpr size_t kSplitCandidates = 1024;
// Meshes below kSplitMinTriangles are neither scanned nor split: their copy runs on one thread, where the scan is not hidden behind the
// memory traffic — 64 icospheres of 5120 triangles cost synchronize() 1.6 ms — and their boxes are small things in the TLAS anyway.
constexpr size_t kSplitMinTriangles = 8192;
constexpr size_t kSplitKeep = kSplitCandidates;
// a triangle that passed the floor: into the thread's heap of the most wasteful ones
inline void consider_triangle(SplitScan& sc, const float* v0, const float* v1, const float* v2, uint32_t i);
inline void scan_triangle(SplitScan& sc, const float* v0, const float* v1, const float* v2, uint32_t i)
{
    float area2 = 0.0f;
    {
        float ext[3];
        for (int a = 0; a < 3; a++) {
            const float l = std::min(v0[a], std::min(v1[a], v2[a])), h = std::max(v0[a], std::max(v1[a], v2[a]));
            sc.lo[a] = std::min(sc.lo[a], l); sc.hi[a] = std::max(sc.hi[a], h);
            ext[a] = h - l;
        }
        area2 = 2.0f * (ext[0] * ext[1] + ext[1] * ext[2] + ext[2] * ext[0]);
    }
    if (++sc.since_refresh >= 256u) {
        sc.since_refresh = 0;
        const double a = box_area6(sc.lo, sc.hi);
        sc.floor_area = a < 1e30 ? (float)((double)sc.tau_rel * a) : 0.0f;
    }
    if (!(area2 > sc.floor_area)) return; // (the common small triangle leaves here)
    consider_triangle(sc, v0, v1, v2, i);
}
inline void consider_triangle(SplitScan& sc, const float* v0, const float* v1, const float* v2, uint32_t i)
{
    const float w = triangle_waste(v0, v1, v2);
    if (!(w > 0.0f) || !(w < INFINITY)) return;
    const auto greater = [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); };
    if (sc.top.size() < kSplitKeep) {
        sc.top.emplace_back(w, i);
        std::push_heap(sc.top.begin(), sc.top.end(), greater);
    } else if (greater(std::make_pair(w, i), sc.top.front())) { // (ties on the waste: the lower index stays, as in the final selection)
        std::pop_heap(sc.top.begin(), sc.top.end(), greater);
        sc.top.back() = std::make_pair(w, i);
        std::push_heap(sc.top.begin(), sc.top.end(), greater);
    }
}
} // namespace

// References of the mesh's most wasteful triangles (MeshHost): the part with the largest waste is cut first (at the middle of the longest axis
// of its box) until no part wastes more than tau = split_tau x the area of the mesh's box, or the budget of duplicates is spent.
static void split_references(const float split_tau, MeshHost& m, const std::vector<SplitScan>& scans, size_t budget)
{
    m.pieces.clear();
    m.n_refs = m.n_orig;
    if (!(split_tau > 0.0f) || budget == 0 || m.n_orig < kSplitMinTriangles) return;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const SplitScan& sc : scans)
        for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], sc.lo[a]); hi[a] = std::max(hi[a], sc.hi[a]); }
    const double root_area = box_area6(lo, hi);
    if (!(root_area > 0.0) || !(root_area < 1e30)) return;
    // ... and more than 64 times an even share of the mesh's box area: in a small mesh (an icosphere of 5120 triangles) EVERY triangle wastes
    // 1e-4 of the box — splitting is for the outliers of a mesh, not for its typical triangle
    const float tau = (float)(std::max((double)split_tau, 64.0 / (double)m.n_orig) * root_area);
    // the candidates: the triangles that waste more than tau, at most the kSplitCandidates most wasteful ones (every thread kept that many of its share)
    std::vector<std::pair<float, uint32_t>> cand;
    for (const SplitScan& sc : scans)
        for (const auto& c : sc.top)
            if (c.first > tau) cand.push_back(c);
    if (cand.empty()) return;
    if (cand.size() > kSplitCandidates) { // the most wasteful ones (ties: the lower index), whatever order the threads delivered them in
        std::sort(cand.begin(), cand.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first > b.first || (a.first == b.first && a.second < b.second); });
        cand.resize(kSplitCandidates);
    }
    struct Part { float waste; uint32_t tri; uint32_t serial; SplitPoly poly; float lo[3], hi[3]; };
    std::sort(cand.begin(), cand.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.second < b.second; });
    // the parts live in `pool`; the heap orders their indices: max-heap on (waste, then the earlier part) — a total order, the same whatever
    // the thread count was
    std::vector<Part> pool;
    pool.reserve(cand.size() + 2 * budget + 2);
    const auto less = [&pool](uint32_t x, uint32_t y) {
        const Part &a = pool[x], &b = pool[y];
        return a.waste < b.waste || (a.waste == b.waste && a.serial > b.serial);
    };
    std::vector<uint32_t> heap;
    std::vector<float> ideal(cand.size()), tri_area(cand.size());
    std::vector<uint32_t> parts_of(cand.size(), 1u);
    uint32_t serial = 0;
    for (size_t c = 0; c < cand.size(); c++) {
        const float* h = m.heads[cand[c].second].v;
        Part p;
        p.tri = (uint32_t)c; p.serial = serial++;
        p.poly.n = 3;
        for (int k = 0; k < 3; k++) std::memcpy(p.poly.p[k], h + 4 * k, 12);
        poly_box(p.poly, p.lo, p.hi);
        p.waste = triangle_waste(h, h + 4, h + 8, &ideal[c]);
        tri_area[c] = (float)poly_area(p.poly);
        heap.push_back((uint32_t)pool.size());
        pool.push_back(p);
    }
    std::make_heap(heap.begin(), heap.end(), less);
    std::vector<uint32_t> done_idx;
    size_t extra = 0;
    constexpr uint32_t kMaxPartsPerTriangle = 64;
    while (!heap.empty()) {
        std::pop_heap(heap.begin(), heap.end(), less);
        const uint32_t pi = heap.back();
        heap.pop_back();
        if (!(pool[pi].waste > tau) || extra >= budget) { // nothing left above the threshold, or the budget is spent: the rest stays whole
            done_idx.push_back(pi);
            done_idx.insert(done_idx.end(), heap.begin(), heap.end());
            break;
        }
        const Part p = pool[pi];
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (p.hi[a] - p.lo[a] > p.hi[axis] - p.lo[axis]) axis = a;
        const float pos = 0.5f * (p.lo[axis] + p.hi[axis]);
        bool cut = parts_of[p.tri] < kMaxPartsPerTriangle && pos > p.lo[axis] && pos < p.hi[axis];
        Part a = p, b = p;
        if (cut) {
            const bool fits_a = clip_poly(p.poly, axis, pos, true, a.poly), fits_b = clip_poly(p.poly, axis, pos, false, b.poly);
            cut = fits_a && fits_b && a.poly.n >= 3 && b.poly.n >= 3;
        }
        if (!cut) { done_idx.push_back(pi); continue; }
        for (Part* q : {&a, &b}) {
            poly_box(q->poly, q->lo, q->hi);
            const double share = tri_area[p.tri] > 0.0f ? poly_area(q->poly) / (double)tri_area[p.tri] : 1.0;
            q->waste = (float)(box_area6(q->lo, q->hi) - share * (double)ideal[p.tri]);
            q->serial = serial++;
            heap.push_back((uint32_t)pool.size());
            pool.push_back(*q);
            std::push_heap(heap.begin(), heap.end(), less);
        }
        parts_of[p.tri]++;
        extra++;
    }
    std::vector<Part> done;
    done.reserve(done_idx.size());
    for (const uint32_t k : done_idx)
        if (parts_of[pool[k].tri] >= 2) done.push_back(pool[k]); // (a candidate that was never cut keeps its own box: no reference to write)
    // parts -> references: per split triangle (in index order) its parts in creation order; the first keeps the triangle's own entry
    std::sort(done.begin(), done.end(), [](const Part& a, const Part& b) { return a.tri < b.tri || (a.tri == b.tri && a.serial < b.serial); });
    size_t next_dup = m.n_orig;
    for (size_t k = 0; k < done.size();) {
        size_t e = k;
        while (e < done.size() && done[e].tri == done[k].tri) e++;
        if (e - k >= 2) {
            const uint32_t orig = cand[done[k].tri].second;
            for (size_t j = k; j < e; j++) {
                SplitPiece sp;
                sp.index = j == k ? orig : (uint32_t)next_dup;
                for (int a = 0; a < 3; a++) { sp.lo[a] = done[j].lo[a]; sp.hi[a] = done[j].hi[a]; }
                sp.pad = 0u;
                m.pieces.push_back(sp);
                if (j != k) {
                    // the duplicate: the original's record (the packet kernel needs its vertices and normal) with the original's index where
                    // k_resolve_duplicates looks for it (the v0 texture coordinate; a duplicate is never shaded)
                    m.tris[next_dup] = m.tris[orig];
                    const uint32_t o = orig;
                    std::memcpy(reinterpret_cast<float*>(&m.tris[next_dup]) + 15, &o, 4);
                    m.heads[next_dup] = m.heads[orig];
                    next_dup++;
                }
            }
        }
        k = e;
    }
    m.n_refs = next_dup;
    std::sort(m.pieces.begin(), m.pieces.end(), [](const SplitPiece& a, const SplitPiece& b) { return a.index < b.index; });
}

// room for the duplicates of split triangles behind the caller's n triangles (allocated with the arrays: a registered array never moves)
static size_t split_slack(const float split_tau, size_t n, bool skinned)
{
    if (!(split_tau > 0.0f) || skinned || n < kSplitMinTriangles) return 0;
    return std::min<size_t>(n / 128 + 64, 1024); // (measured on the bench scenes: 400 ... 2000 duplicates do what 30 000 do, and the host pays per part)
}

static void copy_and_split(MeshHost& m, const rfw_rt_triangle* src, size_t n, int threads, float split_tau, size_t slack);
static void copy_triangles(Instance* I, MeshHost& m, const rfw_rt_triangle* src, size_t n, int threads, bool skinned, size_t region_cap)
{
    const size_t slack = split_slack(I->split_tau, n, skinned), alloc = n + slack;
    if (m.tris.size() != alloc) {
        m.unpin();
        m.tris.clear(); m.tris.shrink_to_fit(); m.tris.resize(alloc);
        m.heads.clear(); m.heads.shrink_to_fit(); m.heads.resize(alloc);
    } else if (!m.pinned && !m.pin_failed && I->blas_on_device && n * sizeof(rfw_rt_triangle) >= (128u << 10) &&
               hipSetDevice(I->device) == hipSuccess) {
        // re-sent at the same size: a mesh that changes.  Registering costs ~0.2 ms per MB, once (a scene that is loaded and never re-sent does not pay it)
        const bool a = hipHostRegister(m.tris.data(), alloc * sizeof(rfw_rt_triangle), hipHostRegisterDefault) == hipSuccess;
        const bool b = a && hipHostRegister(m.heads.data(), alloc * sizeof(TriHead), hipHostRegisterDefault) == hipSuccess;
        if (a && !b) (void)hipHostUnregister(m.tris.data());
        m.pinned = a && b;
        if (!m.pinned) { (void)hipGetLastError(); m.pin_failed = true; } // (the limit on locked memory, say: the uploads are staged by the runtime as before)
    }
    // a re-sent mesh that still fits its region of the device buffers stays there: its duplicates are limited to the room the region has
    // (one reference more than the region holds would abandon it — a hole, the mesh appended — and duplicates are image-neutral: ADVICE r05)
    copy_and_split(m, src, n, threads, I->split_tau, (region_cap >= n && m.split_tau_used == I->split_tau) ? std::min(slack, region_cap - n) : slack);
    m.split_tau_used = I->split_tau;
}
// (host arithmetic only: also what rfw_hip_selftest_splits runs, without a device)
static void copy_and_split(MeshHost& m, const rfw_rt_triangle* src, size_t n, int threads, float split_tau, size_t slack)
{
    m.n_orig = n;
    m.n_refs = n;
    m.pieces.clear();
    const bool scan = slack != 0;
    // one pass over the source: every record is read once and goes out through non-temporal stores, its head a second time into the heads
    // array (a memcpy of the records plus a pass for the heads reads 44 % of the source twice; chunked memcpys lose the streaming stores).
    // The same pass looks for the triangles worth splitting (bounds of the share, its most wasteful triangles: SplitScan)
    // (tried in round 5: ordinary stores for meshes of one thread's worth — slower, 0.37 against 0.33 ms for 5120 triangles; the copy of a
    // small mesh is bound by reading the caller's cold memory, not by the streaming stores)
    auto part = [&m, src, scan](size_t a, size_t b, SplitScan* sc_out) {
        // (the thread's own copy: neighbouring elements of the vector share cache lines, and every triangle updates the bounds)
        SplitScan local = *sc_out;
        SplitScan* sc = &local;
        struct Publish { SplitScan* to; SplitScan* from; ~Publish() { *to = std::move(*from); } } publish{sc_out, sc};
#if defined(__SSE2__)
        if ((reinterpret_cast<uintptr_t>(m.tris.data()) & 15u) == 0 && (reinterpret_cast<uintptr_t>(m.heads.data()) & 15u) == 0) {
            constexpr int kWords = (int)(sizeof(rfw_rt_triangle) / 16);
            __m128 blo = _mm_set1_ps(INFINITY), bhi = _mm_set1_ps(-INFINITY);
            uint32_t since = 0;
            for (size_t i = a; i < b; i++) {
                const __m128i* in = reinterpret_cast<const __m128i*>(src + i);
                __m128i* out = reinterpret_cast<__m128i*>(m.tris.data() + i);
                __m128i* head = reinterpret_cast<__m128i*>(m.heads.data() + i);
                __m128i w[kWords];
                for (int k = 0; k < kWords; k++) w[k] = _mm_loadu_si128(in + k);
                for (int k = 0; k < kWords; k++) _mm_stream_si128(out + k, w[k]);
                for (int k = 0; k < 3; k++) _mm_stream_si128(head + k, w[k]);
                if (scan) {
                    // bounds and box area in vector registers (lane 3 carries texture coordinates: ignored); only a triangle whose box is
                    // large enough to matter takes the scalar path
                    const __m128 v0 = _mm_castsi128_ps(w[0]), v1 = _mm_castsi128_ps(w[1]), v2 = _mm_castsi128_ps(w[2]);
                    const __m128 mn = _mm_min_ps(v0, _mm_min_ps(v1, v2)), mx = _mm_max_ps(v0, _mm_max_ps(v1, v2));
                    blo = _mm_min_ps(blo, mn);
                    bhi = _mm_max_ps(bhi, mx);
                    const __m128 e = _mm_sub_ps(mx, mn);
                    const __m128 pr = _mm_mul_ps(e, _mm_shuffle_ps(e, e, _MM_SHUFFLE(3, 0, 2, 1))); // (ex ey, ey ez, ez ex, -)
                    const float area2 = 2.0f * (_mm_cvtss_f32(pr) + _mm_cvtss_f32(_mm_shuffle_ps(pr, pr, 1)) + _mm_cvtss_f32(_mm_shuffle_ps(pr, pr, 2)));
                    if (++since >= 256u) { // the floor follows the bounds seen so far (SplitScan::floor_area)
                        since = 0;
                        alignas(16) float l4[4], h4[4];
                        _mm_store_ps(l4, blo); _mm_store_ps(h4, bhi);
                        const double a = box_area6(l4, h4);
                        sc->floor_area = a < 1e30 ? (float)((double)sc->tau_rel * a) : 0.0f;
                    }
                    if (area2 > sc->floor_area) {
                        alignas(16) float v[12];
                        for (int k = 0; k < 3; k++) _mm_store_si128(reinterpret_cast<__m128i*>(v) + k, w[k]);
                        consider_triangle(*sc, v, v + 4, v + 8, (uint32_t)i);
                    }
                }
            }
            _mm_sfence();
            if (scan) {
                alignas(16) float l4[4], h4[4];
                _mm_store_ps(l4, blo); _mm_store_ps(h4, bhi);
                for (int a = 0; a < 3; a++) { sc->lo[a] = std::min(sc->lo[a], l4[a]); sc->hi[a] = std::max(sc->hi[a], h4[a]); }
            }
            return;
        }
#endif
        if (a < b) std::memcpy(m.tris.data() + a, src + a, (b - a) * sizeof(rfw_rt_triangle));
        for (size_t k = a; k < b; k++) {
            std::memcpy(&m.heads[k], src + k, sizeof(TriHead));
            if (scan) scan_triangle(*sc, m.heads[k].v, m.heads[k].v + 4, m.heads[k].v + 8, (uint32_t)k);
        }
    };
    const size_t bytes = n * sizeof(rfw_rt_triangle);
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(threads, 1), bytes >> 22)); // at least 4 MB per thread
    std::vector<SplitScan> scans((size_t)nt);
    for (SplitScan& sc : scans) sc.tau_rel = std::max(split_tau, 64.0f / (float)std::max<size_t>(n, 1)); // (as split_references)
    if (nt <= 1) part(0, n, &scans[0]);
    else {
        std::vector<std::thread> pool;
        const size_t per = (n + nt - 1) / nt;
        for (int k = 0; k < nt; k++) {
            const size_t a = std::min(n, per * k), b = std::min(n, a + per);
            if (a < b) pool.emplace_back(part, a, b, &scans[(size_t)k]);
        }
        for (auto& t : pool) t.join();
    }
    const bool kTrace = env_switches().build_trace;
    const auto t_split = std::chrono::steady_clock::now();
    if (scan) split_references(split_tau, m, scans, slack);
    if (kTrace && scan) {
        size_t kept = 0;
        for (const SplitScan& sc : scans) kept += sc.top.size();
        fprintf(stderr, "[build] spatial splits: %zu candidates kept by %d thread(s), %zu duplicates of %zu triangles, %.3f ms after the copy\n", kept, nt, m.n_refs - m.n_orig, n,
                std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_split).count());
    }
}

// Host-only self test of the spatial splits (no GPU needed): runs what set_3d_mesh runs on `tris` and hands back the references.
int64_t rfw_hip_selftest_splits(const rfw_rt_triangle* tris, uint32_t n, float split_tau, uint32_t threads, float* pieces7, uint32_t pieces_cap, uint32_t* duplicate_of,
                                uint32_t duplicates_cap, uint32_t* n_pieces)
{
    if (!tris || !n_pieces) return -1;
    MeshHost m;
    const size_t slack = split_slack(split_tau, n, false);
    m.tris.resize(n + slack);
    m.heads.resize(n + slack);
    copy_and_split(m, tris, n, (int)std::max(threads, 1u), split_tau, slack);
    *n_pieces = (uint32_t)m.pieces.size();
    for (size_t k = 0; k < m.pieces.size() && k < pieces_cap && pieces7; k++) {
        pieces7[7 * k] = (float)m.pieces[k].index;
        for (int a = 0; a < 3; a++) { pieces7[7 * k + 1 + a] = m.pieces[k].lo[a]; pieces7[7 * k + 4 + a] = m.pieces[k].hi[a]; }
    }
    for (size_t j = m.n_orig; j < m.n_refs && j - m.n_orig < duplicates_cap && duplicate_of; j++)
        std::memcpy(&duplicate_of[j - m.n_orig], reinterpret_cast<const float*>(&m.tris[j]) + 15, 4);
    return (int64_t)m.n_refs;
}

int rfw_hip_set_3d_mesh(void* inst, uint32_t id, const rfw_mesh_data_3d* d)
{
    LOCK(inst);
    if (!d || (d->num_triangles && !d->triangles)) return fail(I, RFW_HIP_E_INVALID, "set_3d_mesh: null data");
    if (d->num_triangles > kLeafFirstMask) return fail(I, RFW_HIP_E_INVALID, "set_3d_mesh: more than 2^27 triangles in one mesh");
    MeshHost& m = I->meshes[id];
    if (I->records_pending) { // an incremental build's upload may still be reading a registered host copy
        HIP_TRY(I, hipSetDevice(I->device));
        HIP_TRY(I, hipEventSynchronize(I->ev_records));
        I->records_pending = false;
    }
    const auto t_copy = std::chrono::steady_clock::now();
    const bool skinned = d->skin_data && d->num_skin_data == 3u * d->num_triangles && (d->flags & RFW_MESH_ALLOW_SKINNING);
    size_t region_cap = 0; // the triangles this mesh's region on the device holds, when it has one
    if (const auto it = I->mesh_index.find(id); it != I->mesh_index.end() && I->layout_valid && it->second < I->record_tri_cap.size()) region_cap = I->record_tri_cap[it->second];
    copy_triangles(I, m, d->triangles, d->num_triangles, I->build_threads, skinned, region_cap); // copy: the borrow ends with this call (a skinned mesh is refitted, not split)
    if (env_switches().build_trace) fprintf(stderr, "[build] set_3d_mesh %u: host copy of %u triangles %.3f ms (%s)\n", id, d->num_triangles, std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_copy).count(), m.pinned ? "registered" : "pageable");
    m.skin.clear();
    if (skinned) m.skin.assign(d->skin_data, d->skin_data + d->num_skin_data);
    m.dirty = true;
    I->meshes_dirty = true;
    return RFW_HIP_OK;
}

int rfw_hip_unload_3d_meshes(void* inst, const uint32_t* ids, uint32_t n)
{
    LOCK(inst);
    if (n && !ids) return fail(I, RFW_HIP_E_INVALID, "unload_3d_meshes: null ids");
    if (I->records_pending) { // (a registered host copy must not go away under an upload)
        HIP_TRY(I, hipSetDevice(I->device));
        HIP_TRY(I, hipEventSynchronize(I->ev_records));
        I->records_pending = false;
    }
    for (uint32_t i = 0; i < n; i++) {
        I->meshes.erase(ids[i]);
        I->inst_lists.erase(ids[i]);
    }
    I->meshes_dirty = true;
    I->instances_dirty = true;
    return RFW_HIP_OK;
}

int rfw_hip_set_3d_instances(void* inst, uint32_t mesh, const rfw_instances_data_3d* d)
{
    LOCK(inst);
    if (!d || (d->num_matrices && !d->matrices)) return fail(I, RFW_HIP_E_INVALID, "set_3d_instances: null data");
    InstList& l = I->inst_lists[mesh];
    l.local_aabb = d->local_aabb;
    l.matrices.assign(d->matrices, d->matrices + d->num_matrices);
    l.skin_ids.assign(d->num_matrices, -1);
    if (d->skin_ids)
        for (uint32_t i = 0; i < d->num_matrices && i < d->num_skin_ids; i++) l.skin_ids[i] = d->skin_ids[i];
    I->instances_dirty = true;
    return RFW_HIP_OK;
}

// The trait's `changed` bit slice (packed u32 words, bit i = element i; NULL = everything) folded into what the next synchronize() uploads
static void mark_dirty(Instance::Dirty& d, size_t old_n, uint32_t n, const uint32_t* changed)
{
    if (!changed || old_n != n) { // handed over whole, or the list changed its length
        d.any = true; d.all = true; d.idx.clear();
        return;
    }
    const bool was_all = d.any && d.all;
    bool some = false;
    for (uint32_t i = 0; i < n; i++)
        if (changed[i >> 5] & (1u << (i & 31u))) {
            some = true;
            if (!was_all) d.idx.push_back(i);
        }
    if (some && !was_all) { d.any = true; d.all = false; }
}

int rfw_hip_set_materials(void* inst, const rfw_device_material* m, uint32_t n, const uint32_t* changed)
{
    LOCK(inst);
    if (n && !m) return fail(I, RFW_HIP_E_INVALID, "set_materials: null data");
    mark_dirty(I->mat_dirty, I->materials.size(), n, changed);
    I->materials.assign(m, m + n);
    I->materials_dirty = I->materials_dirty || I->mat_dirty.any;
    return RFW_HIP_OK;
}

// gpu-rt keeps every material texture as one layer of a 1024 x 1024 array with 5 mip levels: a texture of another size is resampled to
// 1024^2 and its mip chain regenerated on the host (backends/gpu-rt/src/lib.rs:1230-1246: `t.resized(1024, 1024)` +
// `generate_mipmaps(Texture::MIP_LEVELS)`, both from the un-vendored crate l3d 0.3, crates/rfw-scene/Cargo.toml), so shade.comp's LOD
// arithmetic (MIPLEVELCOUNT 5, shade.comp:39,273-281) always sees that geometry.  Restated here with the one meaning this project pins
// for l3d's two helpers: point resampling (source texel of the destination texel's centre) and a 2 x 2 box filter per channel, rounded to
// nearest — the filter rfw-rs_amd/host already uses for the mips it hands over.  Option "texture_array" = 0 samples at native size.
constexpr uint32_t kTexArraySize = 1024, kTexArrayMips = 5;
static void normalise_texture(TexHost& t)
{
    if (t.w == 0 || t.h == 0 || (t.w == kTexArraySize && t.h == kTexArraySize)) return; // already an array layer: kept as handed over
    std::vector<uint32_t> out;
    out.reserve((size_t)kTexArraySize * kTexArraySize * 4 / 3 + 16);
    out.resize((size_t)kTexArraySize * kTexArraySize);
    for (uint32_t y = 0; y < kTexArraySize; y++) {
        const uint32_t sy = (uint32_t)(((uint64_t)(2 * y + 1) * t.h) / (2 * kTexArraySize)); // floor((y + 0.5) * h / 1024)
        for (uint32_t x = 0; x < kTexArraySize; x++) {
            const uint32_t sx = (uint32_t)(((uint64_t)(2 * x + 1) * t.w) / (2 * kTexArraySize));
            out[(size_t)y * kTexArraySize + x] = t.texels[(size_t)sy * t.w + sx];
        }
    }
    size_t src = 0;
    uint32_t w = kTexArraySize, h = kTexArraySize;
    for (uint32_t l = 1; l < kTexArrayMips; l++) {
        const uint32_t nw = w >> 1, nh = h >> 1;
        const size_t dst = out.size();
        out.resize(dst + (size_t)nw * nh);
        for (uint32_t y = 0; y < nh; y++)
            for (uint32_t x = 0; x < nw; x++) {
                const uint32_t a = out[src + (size_t)(2 * y) * w + 2 * x], b = out[src + (size_t)(2 * y) * w + 2 * x + 1],
                               c = out[src + (size_t)(2 * y + 1) * w + 2 * x], d = out[src + (size_t)(2 * y + 1) * w + 2 * x + 1];
                uint32_t r = 0;
                for (int ch = 0; ch < 4; ch++) {
                    const uint32_t sum = ((a >> (8 * ch)) & 255u) + ((b >> (8 * ch)) & 255u) + ((c >> (8 * ch)) & 255u) + ((d >> (8 * ch)) & 255u);
                    r |= ((sum + 2u) / 4u) << (8 * ch);
                }
                out[dst + (size_t)y * nw + x] = r;
            }
        src = dst;
        w = nw; h = nh;
    }
    t.texels.swap(out);
    t.w = kTexArraySize; t.h = kTexArraySize; t.mips = kTexArrayMips;
}

static bool copy_texture(TexHost& t, const rfw_texture_data* d)
{
    t = TexHost();
    if (!d || !d->bytes || d->width == 0 || d->height == 0) return true; // an empty texture samples as zero
    if (d->format != RFW_FORMAT_BGRA8 && d->format != RFW_FORMAT_RGBA8) return false;
    t.w = d->width; t.h = d->height; t.format = d->format;
    uint32_t w = d->width, h = d->height, levels = 0;
    size_t texels = 0;
    for (uint32_t l = 0; l < (d->mip_levels ? d->mip_levels : 1u) && w > 0 && h > 0; l++) { // structs.rs:79-121: level l is (w >> l) x (h >> l)
        texels += (size_t)w * h;
        w >>= 1; h >>= 1;
        levels++;
    }
    t.mips = levels;
    t.texels.resize(texels);
    std::memcpy(t.texels.data(), d->bytes, texels * 4); // copy: the borrow ends with this call
    return true;
}

// `changed` (the trait's BitSlice, bit k = texture k): textures whose bit is clear are not looked at — not copied, not resampled into the
// 1024 x 1024 x 5 array — and synchronize() uploads only the changed ones in place when the array's layout stays the same (same count,
// same stored size per texture; with texture_array on, every texture has the same stored size).
int rfw_hip_set_textures(void* inst, const rfw_texture_data* textures, uint32_t n, const uint32_t* changed)
{
    LOCK(inst);
    if (n && !textures) return fail(I, RFW_HIP_E_INVALID, "set_textures: null data");
    const bool partial = changed && I->textures.size() == n && I->tex_offsets.size() == n; // a laid-out array of the same length exists
    I->textures.resize(n);
    for (uint32_t k = 0; k < n; k++) {
        if (partial && !((changed[k / 32] >> (k % 32)) & 1u)) continue;
        TexHost t;
        if (!copy_texture(t, textures + k)) return fail(I, RFW_HIP_E_INVALID, "set_textures: unknown texel format");
        if (I->texture_array) normalise_texture(t);
        const TexHost& old = I->textures[k];
        if (partial && !I->tex_layout_dirty && t.texels.size() == old.texels.size() && t.w == old.w && t.h == old.h && t.mips == old.mips && t.format == old.format)
            I->tex_dirty_idx.push_back(k);
        else
            I->tex_layout_dirty = true;
        I->textures[k] = std::move(t);
    }
    if (!partial) I->tex_layout_dirty = true;
    I->textures_dirty = true;
    return RFW_HIP_OK;
}


int rfw_hip_synchronize(void* inst)
{
    LOCK(inst);
    return do_synchronize(I);
}


int rfw_hip_set_blue_noise(void* inst, const uint32_t* table, uint32_t n_words)
{
    LOCK(inst);
    HIP_TRY(I, hipSetDevice(I->device));
    if (n_words != 0 && (!table || n_words != kBlueNoiseWords))
        return fail(I, RFW_HIP_E_INVALID, "set_blue_noise: expected the 5 * 65536 words of gpu_rt::blue_noise::create_blue_noise_buffer() (or 0 words to clear)");
    std::vector<uint8_t> bytes(n_words);
    for (uint32_t k = 0; k < n_words; k++) {
        if (table[k] > 255u) return fail(I, RFW_HIP_E_INVALID, "set_blue_noise: table entries are bytes (0..255)");
        bytes[k] = (uint8_t)table[k];
    }
    // frames in flight may still sample the old tables
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    for (Instance* c : I->slots) HIP_TRY(I, hipStreamSynchronize(c->stream));
    if (n_words) {
        HIP_TRY(I, I->d_blue_noise.ensure(n_words));
        HIP_TRY(I, hipMemcpy(I->d_blue_noise.ptr, bytes.data(), n_words, hipMemcpyHostToDevice));
    }
    I->has_blue_noise = n_words != 0;
    I->sample_count = 0; // the image accumulated so far was drawn from other numbers
    I->restart = true;
    return RFW_HIP_OK;
}


int rfw_hip_set_point_lights(void* inst, const rfw_point_light* l, uint32_t n, const uint32_t* changed)
{
    LOCK(inst);
    if (n && !l) return fail(I, RFW_HIP_E_INVALID, "set_point_lights: null data");
    mark_dirty(I->point_dirty, I->point_lights.size(), n, changed);
    I->point_lights.assign(l, l + n);
    I->lights_dirty = I->lights_dirty || I->point_dirty.any;
    return RFW_HIP_OK;
}
int rfw_hip_set_spot_lights(void* inst, const rfw_spot_light* l, uint32_t n, const uint32_t* changed)
{
    LOCK(inst);
    if (n && !l) return fail(I, RFW_HIP_E_INVALID, "set_spot_lights: null data");
    mark_dirty(I->spot_dirty, I->spot_lights.size(), n, changed);
    I->spot_lights.assign(l, l + n);
    I->lights_dirty = I->lights_dirty || I->spot_dirty.any;
    return RFW_HIP_OK;
}
int rfw_hip_set_area_lights(void* inst, const rfw_area_light* l, uint32_t n, const uint32_t* changed)
{
    LOCK(inst);
    if (n && !l) return fail(I, RFW_HIP_E_INVALID, "set_area_lights: null data");
    mark_dirty(I->area_dirty, I->area_lights.size(), n, changed);
    I->area_lights.assign(l, l + n);
    I->lights_dirty = I->lights_dirty || I->area_dirty.any;
    return RFW_HIP_OK;
}
int rfw_hip_set_directional_lights(void* inst, const rfw_directional_light* l, uint32_t n, const uint32_t* changed)
{
    LOCK(inst);
    if (n && !l) return fail(I, RFW_HIP_E_INVALID, "set_directional_lights: null data");
    mark_dirty(I->dir_dirty, I->directional_lights.size(), n, changed);
    I->directional_lights.assign(l, l + n);
    I->lights_dirty = I->lights_dirty || I->dir_dirty.any;
    return RFW_HIP_OK;
}

int rfw_hip_set_skybox(void* inst, const rfw_texture_data* skybox)
{
    LOCK(inst);
    if (!copy_texture(I->skybox, skybox)) return fail(I, RFW_HIP_E_INVALID, "set_skybox: unknown texel format");
    I->textures_dirty = true;
    I->tex_layout_dirty = true; // the skybox lives behind the textures in the same array
    return RFW_HIP_OK;
}
int rfw_hip_set_skins(void* inst, const rfw_skin_data* skins, uint32_t n, const uint32_t* /*changed*/)
{
    LOCK(inst);
    if (n && !skins) return fail(I, RFW_HIP_E_INVALID, "set_skins: null data");
    I->skins.resize(n);
    for (uint32_t i = 0; i < n; i++) {
        if (skins[i].num_joint_matrices && !skins[i].joint_matrices) return fail(I, RFW_HIP_E_INVALID, "set_skins: null joint matrices");
        I->skins[i].assign(skins[i].joint_matrices, skins[i].joint_matrices + skins[i].num_joint_matrices);
    }
    I->instances_dirty = true; // the skinned copies and their BLAS are rebuilt with the instances (gpu-rt/src/lib.rs:1318-1336)
    return RFW_HIP_OK;
}


int rfw_hip_get_scene_stats(void* inst, rfw_hip_scene_stats* out)
{
    LOCK(inst);
    if (!out) return fail(I, RFW_HIP_E_INVALID, "get_scene_stats: null out");
    if (per_slot_tlas(I) && I->synchronized) {
        (void)hipSetDevice(I->device);
        const int trc = ensure_slot_tlas(I, I); // the owner's own TLAS may be stale: its slots rebuild theirs independently
        if (trc != RFW_HIP_OK) return trc;
    }
    if (I->node_counts_stale && I->d_mesh_node_counts.ptr) { // after an incremental build: the builders' node counts, read when somebody asks
        (void)hipSetDevice(I->device);
        (void)hipStreamSynchronize(I->stream);
        std::vector<uint32_t> counts(I->mesh_records.size(), 0u);
        if (!counts.empty() && hipMemcpy(counts.data(), I->d_mesh_node_counts.ptr, counts.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            uint64_t n = 0;
            for (auto& kv : I->mesh_index) n += counts[kv.second];
            I->n_blas_nodes = n;
            I->node_counts_stale = false;
        }
    }
    out->triangles = I->n_tris_logical; // the caller's triangles (I->n_tris also counts the duplicates of split triangles)
    out->split_references = I->n_split_refs;
    out->accel_bytes = I->d_blas_nodes.cap * sizeof(Node4Q) + I->d_blas_oct.cap * sizeof(Node4Q) + I->d_blas_wide.cap * sizeof(PacketNode) + I->d_packets.cap * sizeof(TriPacket) +
                       I->d_tlas_nodes.cap * sizeof(Node4Q) + I->d_tlas_oct.cap * sizeof(Node4Q) + I->d_tlas_wide.cap * sizeof(PacketNode);
    out->packet_copies = I->d_blas_wide.ptr ? 1u : 0u;
    out->pad = 0u;
    out->instances = I->n_valid_instances;
    out->blas_nodes = I->n_blas_nodes;
    if (I->tlas_on_device && I->d_node_count.ptr && I->synchronized) {
        uint32_t nc = 0;
        (void)hipSetDevice(I->device);
        (void)hipStreamSynchronize(I->stream);
        if (hipMemcpy(&nc, I->d_node_count.ptr, 4, hipMemcpyDeviceToHost) == hipSuccess) I->n_tlas_nodes = nc;
    }
    out->tlas_nodes = I->n_tlas_nodes;
    out->node_bytes = sizeof(Node4Q);
    out->tri_bytes = sizeof(TriPacket);
    out->ms_blas_build = I->ms_blas_build;
    out->ms_tlas_build = I->ms_tlas_build;
    if (I->build_events_pending) {
        (void)hipSetDevice(I->device);
        if (hipEventSynchronize(I->ev_build[2]) == hipSuccess) {
            // (a full build overlaps the two: the records are still on the bus while the trees are built, so upload + kernels > the build)
            if (I->ev_records && I->records_timed) (void)hipEventElapsedTime(&I->ms_blas_upload, I->ev_build[0], I->ev_records);
            else (void)hipEventElapsedTime(&I->ms_blas_upload, I->ev_build[0], I->ev_build[1]);
            (void)hipEventElapsedTime(&I->ms_blas_kernels, I->ev_build[1], I->ev_build[2]);
        }
        I->build_events_pending = false;
    }
    out->ms_blas_upload = I->ms_blas_upload;
    out->ms_blas_kernels = I->ms_blas_kernels;
    out->blas_upload_bytes = I->blas_upload_bytes;
    out->blas_kernel_bytes = I->blas_kernel_bytes;
    return RFW_HIP_OK;
}

} // extern "C"
