#pragma once
// api_internal.h — what the translation units of the C ABI share (include/rfw_hip.h): the instance state and its small helpers.
// The ABI itself is split by concern (round 4; it was one 3100-line file):
//   api_scene.cpp     set_* calls, synchronize: BLAS / TLAS builds, versioned material and light tables, textures
//   api_frame.cpp     create / destroy, frame slots, the per-frame launch sequence (render, batches, samples), reads and downloads, options, timing
//   api_exchange.cpp  the sharded frame: packing, RCCL all-gather inside the library, the peer-store exchange, de-tiling
//   api_query.cpp     ray queries (TIntersector shape), debug taps, probes
// Instance state, scene upload, acceleration-structure builds and the per-frame launch sequence together are the  Host-side counterpart of backends/gpu-rt/src/lib.rs
// (synchronize :1309-1683, render :1685-1780) with every buffer resident in HBM and no per-bounce
// host read-back.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <unistd.h>
// RCCL: types and prototypes only — librccl is opened at run time by rfw_hip_comm_* (no link-time dependency for single-GPU hosts), and a ROCm
// installation without the rccl development headers can still build this library: the handful of declarations used here are then made locally
#if __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclFloat = 7 } ncclDataType_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* id);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream);
const char* ncclGetErrorString(ncclResult_t r);
}
#endif

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rfw_hip.h"
#include "bvh_host.h"
#include "env_switches.h"
#include "kernels.h"
#include "lbvh.h"
#include "sah_build.h"
#include "traverse.h"

using namespace rfwhip;

namespace rfwapi {

extern thread_local std::string g_create_error; // (api_frame.cpp)

// RCCL, resolved lazily: a process that already carries a librccl (PyTorch bundles one under the same SONAME) keeps using that one
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) get_unique_id = nullptr;
    decltype(&ncclCommInitRank) comm_init_rank = nullptr;
    decltype(&ncclCommDestroy) comm_destroy = nullptr;
    decltype(&ncclAllGather) all_gather = nullptr;
    decltype(&ncclGetErrorString) error_string = nullptr;
    std::string error;
    bool load()
    {
        if (all_gather) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) { error = std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?"); return false; }
        get_unique_id = (decltype(get_unique_id))dlsym(lib, "ncclGetUniqueId");
        comm_init_rank = (decltype(comm_init_rank))dlsym(lib, "ncclCommInitRank");
        comm_destroy = (decltype(comm_destroy))dlsym(lib, "ncclCommDestroy");
        error_string = (decltype(error_string))dlsym(lib, "ncclGetErrorString");
        all_gather = (decltype(all_gather))dlsym(lib, "ncclAllGather");
        if (!get_unique_id || !comm_init_rank || !comm_destroy || !all_gather || !error_string) { error = "librccl lacks a symbol"; all_gather = nullptr; return false; }
        return true;
    }
};
extern Rccl g_rccl;        // (api_exchange.cpp)
struct Instance;
int loop_all_gather(Instance* I, hipStream_t s, uint64_t n_words); // (api_exchange.cpp) what ncclAllGather does for a communicator, for a loop-back hub
void loop_leave(Instance* owner);
extern std::mutex g_rccl_mu;

template <typename T> struct DevBuf {
    T* ptr = nullptr;
    size_t cap = 0; // elements
    hipError_t ensure(size_t n)
    {
        if (n <= cap) return hipSuccess;
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
        const size_t want = std::max<size_t>(n, 16);
        hipError_t e = hipMalloc((void**)&ptr, want * sizeof(T));
        if (e == hipSuccess) cap = want;
        return e;
    }
    // capacity for n elements, KEEPING the first `keep` elements (an append to a mega-buffer): new allocation, device-to-device copy on
    // `s`, then the old one is freed (hipFree waits for the device, so work still reading the old pointer finishes first)
    hipError_t grow_keep(size_t n, size_t keep, hipStream_t s)
    {
        if (n <= cap) return hipSuccess;
        const size_t want = std::max<size_t>(n + n / 2, 16);
        T* np = nullptr;
        hipError_t e = hipMalloc((void**)&np, want * sizeof(T));
        if (e != hipSuccess) return e;
        if (ptr && keep) e = hipMemcpyAsync(np, ptr, std::min(keep, cap) * sizeof(T), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (ptr) (void)hipFree(ptr);
        ptr = np;
        cap = want;
        return e;
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        cap = 0;
    }
};
// Small host -> device uploads whose source may change before the copy runs (materials, lights, mesh records ...): staged through pinned
// blocks that are reused once the copy that read them has completed (an event per block; no stream synchronisation anywhere)
struct PinnedRing {
    struct Block { void* p = nullptr; size_t cap = 0; hipEvent_t ev = nullptr; bool pending = false; };
    std::vector<Block> blocks;
    hipError_t upload(void* dst, const void* src, size_t bytes, hipStream_t s)
    {
        if (bytes == 0) return hipSuccess;
        Block* b = nullptr;
        for (Block& c : blocks) {
            if (c.cap < bytes) continue;
            if (c.pending && hipEventQuery(c.ev) == hipSuccess) c.pending = false;
            if (!c.pending) { b = &c; break; }
        }
        if (!b && blocks.size() >= 64) { // the ring is capped: wait for the oldest copy that used a block big enough instead of pinning more memory
            for (Block& c : blocks) {
                if (c.cap < bytes) continue;
                (void)hipEventSynchronize(c.ev);
                c.pending = false;
                b = &c;
                break;
            }
        }
        if (!b) {
            Block n;
            hipError_t e = hipHostMalloc(&n.p, std::max<size_t>(bytes, 64 << 10), hipHostMallocDefault);
            if (e != hipSuccess) return e;
            n.cap = std::max<size_t>(bytes, 64 << 10);
            if ((e = hipEventCreateWithFlags(&n.ev, hipEventDisableTiming)) != hipSuccess) { (void)hipHostFree(n.p); return e; }
            blocks.push_back(n);
            b = &blocks.back();
        }
        std::memcpy(b->p, src, bytes);
        hipError_t e = hipMemcpyAsync(dst, b->p, bytes, hipMemcpyHostToDevice, s);
        if (e == hipSuccess) e = hipEventRecord(b->ev, s);
        b->pending = e == hipSuccess;
        return e;
    }
    void release()
    {
        for (Block& c : blocks) {
            if (c.ev) (void)hipEventDestroy(c.ev);
            if (c.p) (void)hipHostFree(c.p);
        }
        blocks.clear();
    }
};

// Spatial splits (round 5): a triangle whose box is much larger than the triangle needs — a wall of two triangles across the whole scene —
// is referenced SEVERAL times, every reference with the tight box of one part of the triangle (the triangle clipped at the middle of the
// longest axis of the part's box, recursively).  The tree is built over the references.  Reference 0 of a split triangle is the triangle's
// own entry with another box; the further ones are DUPLICATE records appended behind the caller's triangles (records [n_orig, n_refs)),
// whose packets carry the original's id — hits, ties (lowest id) and shading never see a duplicate.  set_3d_mesh computes them on the host.
struct SplitPiece {
    uint32_t index;     // mesh-local primitive: the triangle itself (< n_orig) or one of its duplicates (>= n_orig)
    float lo[3], hi[3]; // unpadded box of the part (padded on the device like every primitive box)
    uint32_t pad;
};
static_assert(sizeof(SplitPiece) == 32, "SplitPiece");
struct MeshHost {
    // the caller's n_orig records, then (n_refs - n_orig) duplicates of split triangles; the arrays are allocated with room for the
    // duplicates (slack) so that a registered array never moves.  Everything that STORES the mesh counts n_refs, everything the caller sees n_orig.
    size_t n_orig = 0, n_refs = 0;
    float split_tau_used = -1.0f;     // option spatial_splits at the last copy (a re-send under the same setting keeps to its region's room)
    std::vector<SplitPiece> pieces;   // box overrides of every reference of a split triangle (sorted by index)
    std::vector<rfw_rt_triangle> tris;
    // The first 48 B of every record (vertex0 u0 | vertex1 u1 | vertex2 u2), copied beside it by set_3d_mesh: all the device builders read
    // of a triangle.  A full build sends these first (27 % of the bytes) and builds the trees while the records follow on another stream.
    std::vector<TriHead> heads;
    // both arrays registered with the HIP runtime (meshes of >= 128 KB, from their first re-send on): copies from them are asynchronous and run at the link's rate.
    // Nothing reads them behind the host's back once synchronize() has returned (it waits for both uploads).
    bool pinned = false, pin_failed = false;
    void unpin()
    {
        if (!pinned) return;
        (void)hipHostUnregister(tris.data());
        (void)hipHostUnregister(heads.data());
        pinned = false;
    }
    MeshHost() = default;
    MeshHost(const MeshHost&) = delete;
    MeshHost& operator=(const MeshHost&) = delete;
    ~MeshHost() { unpin(); }
    HostBvh4 bvh;
    std::vector<TriPacket> packets; // leaf order
    std::vector<rfw_joint_data> skin; // per vertex (3 per triangle); empty = not skinnable
    bool dirty = true;
};
// one skinned copy of a mesh per (mesh id, skin id) pair some instance references (gpu-rt/src/lib.rs:1318-1336 skins the
// mesh in place; keeping a copy per pair lets two instances of one mesh wear different skins)
struct DerivedMesh {
    uint32_t record = 0;      // index in mesh_records (after the static meshes, in (mesh id, skin id) order)
    uint32_t src_record = 0;  // the static record holding the bind-pose triangles
    size_t skin_offset = 0;   // first rfw_joint_data of the source mesh in d_skin_data
    // refit (every builder but DEVICE_LBVH): the tree is built once, by binned SAH over the first pose; afterwards only its boxes follow
    bool topology_built = false;
    uint32_t node_count = 0;  // 4-wide nodes of that tree
};
struct TexHost {
    uint32_t w = 0, h = 0, mips = 0, format = 0;
    std::vector<uint32_t> texels; // all levels back to back
};
struct InstList {
    rfw_aabb local_aabb{};
    std::vector<rfw_mat4> matrices;
    std::vector<int32_t> skin_ids; // per slot, -1 = none
};

// option "packet_trace" when nobody sets it: camera rays as packets (C4 5920 -> 7040 Mrays/s, C2 6980 -> 7780, C3 5160 -> 5850; the camera paths'
// shadow rays are not coherent enough for it: k_shadow 0.32 -> 1.09 ms, EXPERIMENTS.md)
constexpr int kDefaultPacketTrace = 1;
// ... for scenes the caches still help with.  A packet fetches its next node through the scalar cache, one dependent load per wavefront and
// step: far outside every cache (33.5 M triangles: 1.1 GB of nodes per octant copy) each of those is a trip to HBM and the one-ray-per-lane
// kernel, with 64 independent loads in flight per wavefront, wins — `bench.py --workload atrium32m`: 2310 Mrays/s with packets, 3090 without
// (round 3: 2630); tools/probes/packet_crossover.py puts the crossover between 8 M and 17 M triangles.  While nobody sets the option the
// camera rays of a scene beyond this many triangles go one per lane.
constexpr uint64_t kPacketAutoMaxTriangles = 12u << 20;
inline uint64_t packet_auto_limit() { const uint64_t e = env_switches().packet_auto_max_triangles; return e ? e : kPacketAutoMaxTriangles; } // (RFW_PACKET_AUTO_MAX_TRIANGLES moves the limit, for tests)
enum EvId { EV_FRAME0 = 0, EV_FRAME1, EV_KERNEL_BASE }; // per kernel: start, stop
constexpr int kMaxBounces = 8;
constexpr int kKernelsPerBounce = 3; // trace, shade, shadow
constexpr int kNumEvents = EV_KERNEL_BASE + 2 * (kMaxBounces * kKernelsPerBounce + 1);
constexpr int kTimingRing = 32;
constexpr size_t kSpillMargin = 65536; // extra per-thread spill slots per sub-shard for padded launch grids
constexpr int kMaxSub = 8;      // sub-shards (HIP streams) a frame is split into on one GPU // frames whose events can be pending before rfw_hip_drain_timing must be called

struct Instance {
    std::mutex mu;
    std::string err;
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    uint32_t width = 0, height = 0;
    uint32_t max_path_length = 3;
    float clamp_value = 10.0f;
    uint32_t rank = 0, world = 1, tile_size = 64;
    bool after_batch = false;
    hipEvent_t download_done = nullptr; // recorded behind the last rfw_hip_download_frame copy on this slot's stream
    // The linear accumulator frame is made on demand (rfw_hip_read_accumulator*, download what = 1) from where the samples live: the
    // instance's own slab (world == 1) or the gathered buffer of the last assemble (world > 1, caller-owned: valid until the next one)
    const void* acc_source = nullptr;
    bool acc_source_rgb = false;
    uint32_t acc_source_batch = 1;
    std::vector<const void*> download_dst; // destinations of the copies queued since the last wait on this slot
    uint32_t max_batch = 1; // frames one render_batch() call may trace together (buffers are sized for it)
    uint32_t builder = RFW_HIP_BUILDER_AUTO;
    bool texture_array = true; // material textures as layers of gpu-rt's 1024 x 1024 x 5-mip array (option "texture_array")
    uint32_t flags = 0;
    float sky[3] = {0, 0, 0};
    bool timing = true;
    int build_threads = 8;
    int sah_max_leaf = 8;
    float sah_trav_cost = 1.0f;

    // host-side scene copies (the trait's borrows end with each call)
    std::map<uint32_t, MeshHost> meshes;
    std::map<uint32_t, InstList> inst_lists;
    std::vector<rfw_device_material> materials;
    std::vector<rfw_area_light> area_lights;
    std::vector<rfw_point_light> point_lights;
    std::vector<rfw_spot_light> spot_lights;
    std::vector<rfw_directional_light> directional_lights;
    std::vector<TexHost> textures;
    std::vector<uint32_t> tex_offsets;  // word offset of texture k in d_tex_data as last laid out by synchronize()
    std::vector<uint32_t> tex_dirty_idx; // textures changed in place since then (set_textures with `changed` bits)
    bool tex_layout_dirty = true;        // count / sizes changed, or the skybox: the whole array is laid out again
    TexHost skybox;
    std::vector<std::vector<rfw_mat4>> skins; // joint matrices per skin id
    std::map<std::pair<uint32_t, int32_t>, DerivedMesh> derived;
    DevBuf<rfw_joint_data> d_skin_data;
    DevBuf<rfw_mat4> d_joints;
    DevBuf<uint32_t> d_bounds_scratch;
    uint32_t raw_node_origin = 0; // d_blas_raw[0] holds node raw_node_origin of the mega-buffer
    uint32_t max_derived_tris = 0;
    bool meshes_dirty = true, instances_dirty = true, materials_dirty = true, lights_dirty = true, textures_dirty = true;
    bool synchronized = false;

    // device scene
    DevBuf<Node4Q> d_blas_nodes, d_tlas_nodes;   // what the kernels traverse
    DevBuf<Node4> d_blas_raw, d_tlas_raw;        // device-built trees before quantisation
    // what the PACKET kernels traverse (traverse_packet.h): eight copies of d_*_nodes, one per ray octant, copy `oct` of node i at
    // [oct * stride + i] with stride = the capacity of the quantised array; nullptr when the copies would not fit kMaxPacketNodeBytes
    DevBuf<PacketNode> d_blas_wide, d_tlas_wide;
    DevBuf<Node4Q> d_blas_oct, d_tlas_oct; // the same copies as the one-ray-per-lane kernels read them (make_octant_node), same stride
    DevBuf<TriPacket> d_packets;
    DevBuf<rfw_rt_triangle> d_triangles;
    DevBuf<MeshRecord> d_mesh_records;
    DevBuf<rfw_mat4> d_matrices;
    DevBuf<uint32_t> d_mesh_of_instance, d_tlas_prims;
    DevBuf<InstanceXform> d_xforms;
    DevBuf<InstanceNormal> d_normals;
    // Materials and lights: small tables that an application edits while frames are in flight.  Every synchronize() that changes them
    // writes a NEW version (kTableVersions buffers used round-robin) on a separate upload stream — the previous version copied on the
    // device, then only the changed elements from the host (the trait's `changed` bit slices) — so frames in flight keep reading the
    // version they started with, nothing waits for them, and later frames wait only for the upload (tables_ready).
    static constexpr int kTableVersions = 4;
    struct Tables {
        DevBuf<rfw_device_material> materials;
        DevBuf<rfw_area_light> area;
        DevBuf<rfw_point_light> point;
        DevBuf<rfw_spot_light> spot;
        DevBuf<rfw_directional_light> dir;
        size_t n_mat = 0, n_area = 0, n_point = 0, n_spot = 0, n_dir = 0;
    } tables[kTableVersions];
    uint64_t tables_version = 0;             // owner: version frames rendered from now on read (buffer = version % kTableVersions)
    uint64_t tables_waited = 0, tables_used = 0; // per slot: version its stream has waited for / version its latest frame reads
    // per slot: the OLDEST version a frame of this slot reads that the upload stream has not yet been ordered behind (~0 = none).  A slot
    // renders frame after frame without the host ever waiting, so an old frame can still be executing when its latest frame already reads a
    // newer version: recycling a version buffer must look at the oldest such frame, not at the latest
    uint64_t tables_oldest_pending = ~0ull;
    hipStream_t upload_stream = nullptr;
    hipEvent_t tables_ready = nullptr;
    // what the set_* calls since the last synchronize changed: all, or a list of element indices
    struct Dirty { bool any = false, all = true; std::vector<uint32_t> idx; void clear() { any = false; all = true; idx.clear(); } };
    Dirty mat_dirty, area_dirty, point_dirty, spot_dirty, dir_dirty;
    DevBuf<uint32_t> d_spill;
    DevBuf<uint32_t> d_tex_data;
    DevBuf<TexDesc> d_tex_desc;
    TexDesc skybox_desc{};
    uint32_t n_textures = 0;
    DevBuf<uint8_t> d_blue_noise; // the blue-noise sampler's tables as bytes (rfw_hip_set_blue_noise)
    bool has_blue_noise = false;
    DevBuf<uint32_t> d_valid_gids, d_tlas_order, d_node_count;
    DevBuf<char> d_stage_dev;  // the fused TLAS path: the pinned staging block [matrices | mesh_of | valid_gids | mesh_local] as ONE device copy
    // option "tlas_fused" (owner): up to kTlasFusedMax instances, no skinned copies -> one copy + two launches per instance update instead of the
    // launch chain.  The one-workgroup build takes 0.48 ms of ONE CU where the chain takes 0.27 ms of launch latency and 27 API calls: with frame
    // slots (frames overlap: the CU is one of 256, the host thread is what is scarce) the fused path wins by 5 %, one frame at a time the chain
    // wins by 26 % (C3, measured).  0 = always the chain, 1 = always fused, 2 (default) = fused where the instance has frame slots
    int tlas_fused = 2;
    DevBuf<DevBox> d_inst_boxes, d_mesh_local, d_tri_boxes;
    DevBuf<char> d_lbvh_ws;
    DevBuf<uint32_t> d_blas_order;
    // pinned staging for the per-frame instance upload (truly asynchronous H2D; guarded by stage_event)
    // two blocks used alternately, so the host fills the next frame's block while the previous frame's copy is still queued
    static constexpr int kStages = 2;
    void* stage_buf[kStages] = {};
    size_t stage_cap[kStages] = {};
    hipEvent_t stage_event[kStages] = {};
    bool stage_pending[kStages] = {};
    int stage_next = 0;
    void* stage = nullptr; // the block of the current synchronize
    bool tlas_on_device = true, blas_on_device = false, blas_sah_on_device = false;
    DevBuf<char> d_sah_ws;
    DevBuf<uint32_t> d_mesh_node_counts;
    DevBuf<ForestTree> d_forest; // (first, count, node region) of every mesh of a full build: sah_build_forest
    DevBuf<uint32_t> d_refit_parent, d_refit_nint, d_refit_arrive; // per raw node of the skinned copies
    DevBuf<QueueCounters> d_counters;        // a ring of two blocks of kMaxSub: the frame being issued uses block `counter_phase`, its k_primary clears the other for the next one
    uint32_t counter_phase = 0;
    // traversal stack overflow: a word of pinned host memory the kernels set (device-visible mapping), so every later call can report
    // RFW_HIP_E_STATE without a read-back; cleared when synchronize() rebuilds the trees
    uint32_t* overflow_host = nullptr;
    uint32_t* overflow_dev = nullptr;
    uint32_t spill_rows = kStackSpill; // option "spill_rows" (tests): rows of the HBM spill stack a lane may use
    // persistent scratch of the ray-query calls (no hipMalloc / hipFree — a device-wide sync — per call)
    DevBuf<float> d_q_o, d_q_d, d_q_t;
    DevBuf<rfw_hip_hit> d_q_h;
    DevBuf<uint32_t> d_q_depth;
    DevBuf<uint8_t> d_q_r;
    std::vector<MeshRecord> mesh_records;
    std::map<uint32_t, uint32_t> mesh_index; // mesh id -> index in mesh_records
    // incremental synchronize (device builders, no skinned copies): a changed mesh is rebuilt in its own region of the mega-buffers (or
    // appended behind the others when it grew), the other meshes are not touched (gpu-rt/src/lib.rs:1345-1383 refits only changed meshes)
    std::vector<uint32_t> record_tri_cap;     // triangles the region of record k can hold (its node region holds max(cap, 1) nodes)
    std::vector<uint32_t> record_tri_orig;    // the caller's triangles of record k (tri_count counts the references: + duplicates of split triangles)
    DevBuf<SplitPiece> d_split_pieces;        // box overrides of the meshes being built (build-time only)
    std::vector<uint32_t> record_piece_off, record_piece_n; // where record k's overrides lie in d_split_pieces during the current build
    float split_tau = 8e-5f;                  // option "spatial_splits": a part is cut while its box wastes more than this x the mesh box's area (0 = off)
    uint64_t n_split_refs = 0;                // duplicates in the scene (scene_stats)
    uint32_t tri_end = 0, node_end = 0;       // first free triangle / node slot behind the regions in use
    uint64_t hole_tris = 0;                   // triangles' worth of regions abandoned since the last full build
    bool layout_valid = false;                // a full device build has laid the buffers out; cleared by anything the incremental path does not cover
    bool node_counts_stale = false;           // n_blas_nodes is re-read lazily (get_scene_stats) after an incremental build
    uint32_t incremental_builds = 0, full_builds = 0, heads_first_builds = 0, tlas_fused_builds = 0;
    bool packet_auto = true; // option "packet_trace" / RFW_PACKET_TRACE not set: packets only below kPacketAutoMaxTriangles
    PinnedRing pins;
    uint64_t n_instances = 0, n_valid_instances = 0, n_tris = 0, n_blas_nodes = 0, n_tlas_nodes = 0; // (n_tris: stored primitives, duplicates included)
    uint64_t n_tris_logical = 0; // the caller's triangles
    float ms_blas_build = 0, ms_tlas_build = 0, ms_stage_wait = 0;
    float ms_blas_upload = 0, ms_blas_kernels = 0; // the last full device build, by events
    uint64_t blas_upload_bytes = 0, blas_kernel_bytes = 0;
    // small meshes are built side by side: one worker thread per auxiliary stream, each with scratch of its own (build_meshes)
    struct BuildLane { hipStream_t s = nullptr; hipEvent_t done = nullptr; DevBuf<char> ws; DevBuf<DevBox> boxes; };
    static constexpr int kBuildLanes = 8;
    BuildLane lanes[kBuildLanes];
    hipEvent_t ev_build[3] = {nullptr, nullptr, nullptr};
    // full device build: the 48-B heads go up on `stream`, the 176-B records behind them on `records_stream` while the trees are built
    DevBuf<TriHead> d_heads;
    hipStream_t records_stream = nullptr;
    hipEvent_t ev_heads = nullptr, ev_records = nullptr;
    bool build_from_heads = false; // (inside build_blas_device_full only)
    bool records_pending = false;  // an upload from a registered host copy was queued and nobody has waited for ev_records yet
    bool records_timed = false;    // the last build recorded ev_records (scene stats: upload time = until the records have arrived)
    bool build_events_pending = false; // recorded, not read yet (rfw_hip_get_scene_stats reads them: no synchronisation for them in synchronize())

    // device path state
    DevBuf<float4> d_ray_o[2], d_ray_d[2], d_thr[2], d_sh_o, d_sh_d, d_sh_e, d_acc_slab, d_frame_acc, d_frame_out;
    DevBuf<uint32_t> d_present; // BGRA8 sRGB frame, made on demand by rfw_hip_download_frame(what = 2)
    DevBuf<uint4> d_hit[2];
    // extension rays traced in spatial order (option "sort_extension_rays"): (key, queue index) pairs, sorted with hipCUB on the frame's stream
    DevBuf<uint32_t> d_sort_keys[2], d_sort_vals[2];
    DevBuf<char> d_sort_ws;
    int shade_group = 0; // option "shade_group": threads per k_shade workgroup — 0 = 256 where frames overlap (several frame slots, one frame per call), 512 otherwise; or 256 / 512
    int sort_extension_rays = 2; // 0 never, 1 always, 2 only where it pays: batches of frames / samples (see do_render)
    void* external_slab = nullptr;
    // multi-GPU inside the library (rfw_hip_comm_init): this rank's RGB slab(s) -> ncclAllGather on the instance's stream -> assemble
    ncclComm_t comm = nullptr;
    DevBuf<float> d_send, d_recv;
    // WHAT travels in the all-gather (option "gather_format"): 0 = the slab's linear RGB accumulator as floats (12 B per pixel; every rank can
    // then also hand out the accumulator), 1 = the finished frame sqrt(acc / samples) as halves (6 B), 2 = the presented B, G, R, A bytes
    // (4 B: what the reference draws onto its swap chain).  With 1 and 2 the accumulators stay on the ranks that own the tiles.
    uint32_t gather_format = 0;
    // WHO de-tiles the gathered frame at once (option "present_rank"): -1 = every rank (each render leaves the row-major frame behind
    // everywhere), r >= 0 = only rank r — the one that presents; the other ranks keep the gathered tiles and de-tile when somebody reads
    int present_rank = -1;
    struct Deferred { const void* gathered = nullptr; uint32_t k = 0, samples = 1; } deferred; // a gathered frame not de-tiled yet
    bool presented_valid = false; // d_present holds the de-tiled presented frame(s) of the latest gather (format 2)
    // frame slots share the owner's communicator: collectives of ONE communicator must not run side by side, so every all-gather waits (on
    // the device) for the one issued before it, whichever slot's stream that was on, while the slots' traces overlap freely
    hipEvent_t comm_chain = nullptr;
    bool comm_chain_pending = false;
    // the exchange without a collective library (rfw_hip_p2p_*): receive buffers [slot][rank][frame][slab] and flag words
    // [slot][arrived | credit][rank] of THIS rank, and where the peers' are mapped.  Lives in the owner; a slot knows its index.
    struct LoopHub* loop = nullptr;  // rfw_hip_comm_init_loopback (owner): the test transport that stands in for the communicator
    uint32_t loop_seq = 0;           // per slot: gathers this slot has joined
    hipEvent_t loop_sent = nullptr;  // per slot: its packed tiles are in d_send
    struct P2P {
        bool connected = false;
        uint32_t* data = nullptr;   // hipMalloc: 4-byte words
        uint32_t* flags = nullptr;  // uncached device memory
        size_t slot_words = 0;      // words per frame slot: world x max_batch x capacity x 3 (room for the widest format)
        uint32_t n_slots = 0;
        std::vector<uint32_t*> peer_data, peer_flags;
        std::vector<uint8_t> opened; // bit 0: data, bit 1: flags came from hipIpcOpenMemHandle
        uint64_t timeout_ticks = 500000000ull; // 5 s of the 100 MHz wall clock
    } p2p;
    uint32_t slot_index = 0;
    uint32_t stream_leaf_gate = 16; // ... and a lane that holds a leaf waits until this many do (or nobody has a node to test)
    bool stream_auto = true; // stream_run applies where it was measured to pay (see camera_params); set_option("stream_run") switches this off
    uint32_t stream_run = 8, stream_refill = 12; // measured on C4 path traced (max path length 3, 8 frame slots): 2880 -> 3180 Mrays/s; 0 = off // streaming shadow / extension kernels: wavefront-runs of stream_run x 64 rays (0: one ray per lane)
    uint32_t p2p_seq = 0;           // frames this slot has exchanged
    bool frame_elsewhere = false;   // the latest frame was sent to the presenting rank and does not exist here
    uint32_t tiles_x = 0, tiles_y = 0, local_tiles = 0, capacity = 0;
    uint64_t local_pixels = 0;
    uint32_t sample_count = 0;
    bool have_last_view = false;
    rfw_camera_view_3d last_view{};
    std::vector<hipEvent_t> ring;  // [kTimingRing][substreams][kNumEvents]
    hipEvent_t* events = nullptr;   // event set of the current frame, sub-shard 0
    uint32_t substreams = 1;        // the frame's tiles are dealt to this many sub-shards, each traced on its own stream
    hipStream_t sub[kMaxSub] = {};
    hipEvent_t ev_fork = nullptr, ev_join[kMaxSub] = {};
    uint32_t local_tiles_v = 0, cap_v = 0; // per sub-shard
    uint64_t frame_index = 0, drained_index = 0;
    uint32_t ring_bounces[kTimingRing] = {};
    bool ring_nee[kTimingRing] = {};
    uint32_t last_bounces = 0;
    bool frame_recorded = false;
    bool last_count_flag = false;

    // frames in flight inside ONE instance (options.frames_in_flight > 1): the instance itself is slot 0, `slots` are internal
    // instances that own only per-frame state (path buffers, queues, accumulator, stream) and render the owner's scene.  A render()
    // that starts a new image (new view, changed scene, reset) goes to the next slot; one that adds a sample stays on its slot.
    Instance* scene = nullptr;            // in a slot: the owner whose scene it renders
    std::vector<Instance*> slots;         // in the owner: slots 1 .. frames_in_flight - 1
    uint32_t cur_slot = 0;                // slot of the latest render
    uint64_t scene_version = 1;           // owner: bumped by every synchronize() that changed the scene
    uint64_t rendered_version = 0;        // per slot: scene version of the image it accumulates
    uint64_t waited_version = 0;          // per slot: scene version whose scene_ready event its stream has already waited for
    uint64_t instances_version = 1;       // owner: bumped whenever the instance lists (or what they refer to) changed
    uint64_t tlas_version = 0;            // per slot: instances_version its own TLAS / instance descriptors were built from
    bool restart = false;                 // owner: reset_accumulation() -> the next render starts a new image
    hipEvent_t scene_ready = nullptr;     // owner: recorded after synchronize(); slots wait for it before they read the scene
    hipEvent_t frame_done = nullptr;      // per slot: recorded after its latest render; the owner waits for it before it edits the scene
};

inline Instance* scene_of(Instance* I) { return I->scene ? I->scene : I; }
inline const Instance* scene_of(const Instance* I) { return I->scene ? I->scene : I; }
void p2p_release(Instance* I);
inline Instance* slot_ptr(Instance* I, uint32_t k) { return k == 0 ? I : I->slots[k - 1]; }
// Whose TLAS and instance descriptors a frame reads.  With frame slots every slot keeps its OWN (rebuilt lazily from the owner's
// instance lists when stale), so a scene whose instances move every frame still pipelines; skinned copies live in the owner's
// shared mesh buffers and are rebuilt by synchronize(), so with them all slots share the owner's TLAS and synchronize() drains.
inline bool per_slot_tlas(const Instance* S) { return !S->slots.empty() && S->derived.empty(); }
inline Instance* tlas_of(Instance* I) { Instance* S = scene_of(I); return per_slot_tlas(S) ? I : S; }

#define HIP_TRY(inst, expr)                                                                     \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (inst)->err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return RFW_HIP_E_DEVICE;                                                            \
        }                                                                                       \
    } while (0)

inline int fail(Instance* I, int code, const std::string& msg)
{
    I->err = msg;
    return code;
}

// Did a traversal of this instance (or of one of its frame slots) run out of stack since the trees were last built?
inline bool overflow_seen(const Instance* I)
{
    if (I->overflow_host && *(volatile const uint32_t*)I->overflow_host) return true;
    for (const Instance* c : I->slots)
        if (c->overflow_host && *(volatile const uint32_t*)c->overflow_host) return true;
    return false;
}
inline void clear_overflow(Instance* I)
{
    if (I->overflow_host) *(volatile uint32_t*)I->overflow_host = 0u;
    for (Instance* c : I->slots)
        if (c->overflow_host) *(volatile uint32_t*)c->overflow_host = 0u;
}
#define CHECK_OVERFLOW(inst)                                                                                                          \
    do {                                                                                                                              \
        if (overflow_seen(inst))                                                                                                      \
            return fail(inst, RFW_HIP_E_STATE, "traversal stack overflow: a tree is deeper than the LDS + spill stack (results of the affected rays are incomplete)"); \
    } while (0)

inline bool is_zero_matrix(const rfw_mat4& m)
{
    for (int i = 0; i < 16; i++)
        if (m.m[i] != 0.0f) return false;
    return true;
}

template <typename T> int upload(Instance* I, DevBuf<T>& buf, const T* src, size_t n)
{
    HIP_TRY(I, buf.ensure(n));
    if (n) HIP_TRY(I, hipMemcpyAsync(buf.ptr, src, n * sizeof(T), hipMemcpyHostToDevice, I->stream));
    return RFW_HIP_OK;
}

// ---- functions one translation unit defines and another uses
// api_frame.cpp
void compute_shard(Instance* I);
int alloc_paths(Instance* I);
SceneDev scene_dev(Instance* I);
CameraParams camera_params(const Instance* I, const rfw_camera_view_3d& v, uint32_t sub = 0);
PathDev path_dev(Instance* I, uint32_t sub = 0);
int do_render(Instance* I, const rfw_camera_view_3d* views, uint32_t k = 1, bool samples = false);
uint32_t spill_stride(const Instance* I);
bool blas_wide_wanted(const Instance* I, uint64_t n_prims); // api_scene.cpp: does a packet kernel run on a scene of this size under the current options?
uint32_t index_magic(uint32_t d, uint64_t n_max); // api_frame.cpp: reciprocal for the kernels' index divisions (0: none exact far enough)
// api_scene.cpp
int do_synchronize(Instance* I);
int ensure_slot_tlas(Instance* S, Instance* T);
int ensure_lbvh_ws(Instance* I, uint32_t n);
// api_exchange.cpp
uint64_t slab_words(const Instance* I);
const float* srgb_steps();
void pack_slabs(Instance* I, hipStream_t s, void* dst, uint32_t frames);
int assemble_gathered(Instance* I, hipStream_t s, const void* gathered, uint32_t k, uint32_t samples);
int p2p_exchange(Instance* I, hipStream_t s, uint32_t frames);
int gathered_arrived(Instance* I, hipStream_t s, const void* gathered, uint32_t k);
int ensure_assembled(Instance* I);
bool gathers_tiles(const Instance* I);
bool p2p_timed_out(const Instance* I);

} // namespace rfwapi

#define LOCK(inst)                                    \
    if (!(inst)) return RFW_HIP_E_INVALID;            \
    Instance* I = static_cast<Instance*>(inst);       \
    std::lock_guard<std::mutex> guard_(I->mu)
