// rfw_hip_api.cpp — the C ABI of include/rfw_hip.h: instance state, scene upload, acceleration-structure
// builds and the per-frame launch sequence.  Host-side counterpart of backends/gpu-rt/src/lib.rs
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
#include "kernels.h"
#include "lbvh.h"
#include "sah_build.h"
#include "traverse.h"

using namespace rfwhip;

namespace {

thread_local std::string g_create_error;

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
Rccl g_rccl;
std::mutex g_rccl_mu;

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

struct MeshHost {
    std::vector<rfw_rt_triangle> tris;
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
    DevBuf<QueueCounters> d_counters;
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
    uint32_t tri_end = 0, node_end = 0;       // first free triangle / node slot behind the regions in use
    uint64_t hole_tris = 0;                   // triangles' worth of regions abandoned since the last full build
    bool layout_valid = false;                // a full device build has laid the buffers out; cleared by anything the incremental path does not cover
    bool node_counts_stale = false;           // n_blas_nodes is re-read lazily (get_scene_stats) after an incremental build
    uint32_t incremental_builds = 0, full_builds = 0;
    PinnedRing pins;
    uint64_t n_instances = 0, n_valid_instances = 0, n_tris = 0, n_blas_nodes = 0, n_tlas_nodes = 0;
    float ms_blas_build = 0, ms_tlas_build = 0, ms_stage_wait = 0;
    float ms_blas_upload = 0, ms_blas_kernels = 0; // the last full device build, by events
    uint64_t blas_upload_bytes = 0, blas_kernel_bytes = 0;
    // small meshes are built side by side: one worker thread per auxiliary stream, each with scratch of its own (build_meshes)
    struct BuildLane { hipStream_t s = nullptr; hipEvent_t done = nullptr; DevBuf<char> ws; DevBuf<DevBox> boxes; };
    static constexpr int kBuildLanes = 8;
    BuildLane lanes[kBuildLanes];
    hipEvent_t ev_build[3] = {nullptr, nullptr, nullptr};
    bool build_events_pending = false; // recorded, not read yet (rfw_hip_get_scene_stats reads them: no synchronisation for them in synchronize())

    // device path state
    DevBuf<float4> d_ray_o[2], d_ray_d[2], d_thr[2], d_sh_o, d_sh_d, d_sh_e, d_acc_slab, d_frame_acc, d_frame_out;
    DevBuf<uint32_t> d_present; // BGRA8 sRGB frame, made on demand by rfw_hip_download_frame(what = 2)
    DevBuf<uint4> d_hit[2];
    // extension rays traced in spatial order (option "sort_extension_rays"): (key, queue index) pairs, sorted with hipCUB on the frame's stream
    DevBuf<uint32_t> d_sort_keys[2], d_sort_vals[2];
    DevBuf<char> d_sort_ws;
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

int fail(Instance* I, int code, const std::string& msg)
{
    I->err = msg;
    return code;
}

// Did a traversal of this instance (or of one of its frame slots) run out of stack since the trees were last built?
bool overflow_seen(const Instance* I)
{
    if (I->overflow_host && *(volatile const uint32_t*)I->overflow_host) return true;
    for (const Instance* c : I->slots)
        if (c->overflow_host && *(volatile const uint32_t*)c->overflow_host) return true;
    return false;
}
void clear_overflow(Instance* I)
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

bool is_zero_matrix(const rfw_mat4& m)
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

void compute_shard(Instance* I)
{
    // virtual sharding: world x substreams virtual ranks; virtual rank rank*S + s belongs to this instance's sub-shard s
    const uint32_t S = I->substreams, wv = I->world * S;
    I->tiles_x = (I->width + I->tile_size - 1) / I->tile_size;
    I->tiles_y = (I->height + I->tile_size - 1) / I->tile_size;
    const uint32_t total = I->tiles_x * I->tiles_y;
    I->local_tiles_v = (total + wv - 1) / wv; // every sub-slab is padded to the same size on every rank
    I->cap_v = I->local_tiles_v * I->tile_size * I->tile_size;
    I->local_tiles = I->local_tiles_v * S;
    I->capacity = I->cap_v * S;
    uint64_t px = 0;
    for (uint32_t t = 0; t < total; t++) {
        if ((t % wv) / S != I->rank) continue;
        const uint32_t tx = t % I->tiles_x, ty = t / I->tiles_x;
        const uint32_t w = std::min(I->tile_size, I->width - tx * I->tile_size), h = std::min(I->tile_size, I->height - ty * I->tile_size);
        px += (uint64_t)w * h;
    }
    I->local_pixels = px;
}

int alloc_paths(Instance* I)
{
    compute_shard(I);
    const size_t n = (size_t)I->capacity * I->max_batch; // a batch of frames is one tall virtual frame
    for (int h = 0; h < 2; h++) {
        HIP_TRY(I, I->d_ray_o[h].ensure(n));
        HIP_TRY(I, I->d_ray_d[h].ensure(n));
        HIP_TRY(I, I->d_thr[h].ensure(n));
        HIP_TRY(I, I->d_hit[h].ensure(n));
    }
    HIP_TRY(I, I->d_sh_o.ensure(n * kShadowBuckets));
    HIP_TRY(I, I->d_sh_d.ensure(n * kShadowBuckets));
    HIP_TRY(I, I->d_sh_e.ensure(n * kShadowBuckets));
    HIP_TRY(I, I->d_acc_slab.ensure(n));
    HIP_TRY(I, hipMemsetAsync(I->d_acc_slab.ptr, 0, n * sizeof(float4), I->stream));
    const size_t px = (size_t)I->width * I->height * I->max_batch;
    HIP_TRY(I, I->d_frame_out.ensure(px));
    I->acc_source = nullptr;
    HIP_TRY(I, hipMemsetAsync(I->d_frame_out.ptr, 0, px * sizeof(float4), I->stream));
    // per-thread overflow slots: launch grids are padded (XCD tiling, shadow buckets), so leave a margin per sub-shard
    HIP_TRY(I, I->d_spill.ensure((size_t)kStackSpill * I->substreams * ((size_t)I->cap_v * I->max_batch + kSpillMargin)));
    HIP_TRY(I, I->d_counters.ensure(kMaxSub));
    HIP_TRY(I, hipMemsetAsync(I->d_counters.ptr, 0, kMaxSub * sizeof(QueueCounters), I->stream));
    I->sample_count = 0;
    return RFW_HIP_OK;
}

uint32_t spill_stride(const Instance* I) { return (uint32_t)(I->d_spill.cap / kStackSpill); }

// The per-octant node copies follow the quantised arrays: eight copies, stride = the quantised array's capacity, in both forms (PacketNode
// for the packet kernels, Node4Q for the one-ray-per-lane kernels).  (Re)allocates when that capacity changed — the stride with it, so every
// node in use (`keep` of them) is expanded again.
hipError_t follow_copies(DevBuf<PacketNode>& wide, DevBuf<Node4Q>& oct, const DevBuf<Node4Q>& nodes, size_t keep, hipStream_t s)
{
    const size_t want = nodes.cap * kPacketNodeCopies;
    if (wide.cap == want && oct.cap == want) return hipSuccess;
    wide.release();
    oct.release();
    if (want == 0) return hipSuccess;
    hipError_t e = hipMalloc((void**)&wide.ptr, want * sizeof(PacketNode));
    if (e != hipSuccess) { wide.ptr = nullptr; return e; }
    wide.cap = want;
    e = hipMalloc((void**)&oct.ptr, want * sizeof(Node4Q));
    if (e != hipSuccess) { oct.ptr = nullptr; return e; }
    oct.cap = want;
    OctantCopies oc;
    oc.wide = wide.ptr; oc.quant = oct.ptr; oc.stride = (uint32_t)nodes.cap;
    launch_expand_nodes(s, nodes.ptr, oc, 0u, (uint32_t)std::min(keep, nodes.cap));
    return hipGetLastError();
}
inline OctantCopies copies_of(const DevBuf<PacketNode>& wide, const DevBuf<Node4Q>& oct)
{
    OctantCopies oc;
    oc.wide = wide.ptr; oc.quant = oct.ptr; oc.stride = (uint32_t)(wide.cap / kPacketNodeCopies);
    return oc;
}

SceneDev scene_dev(Instance* I)
{
    SceneDev s;
    const Instance* S = scene_of(I); // a frame slot reads its owner's scene
    const Instance* TL = tlas_of(I);
    s.tlas_nodes = TL->d_tlas_nodes.ptr;
    s.tlas_prims = TL->d_tlas_prims.ptr;
    s.instances = TL->d_xforms.ptr;
    s.instance_normals = TL->d_normals.ptr;
    s.meshes = S->d_mesh_records.ptr;
    s.blas_nodes = S->d_blas_nodes.ptr;
    s.tlas_wide = TL->d_tlas_wide.ptr;
    s.blas_wide = S->d_blas_wide.ptr;
    s.tlas_wide_stride = (uint32_t)(TL->d_tlas_wide.cap / kPacketNodeCopies);
    s.blas_wide_stride = (uint32_t)(S->d_blas_wide.cap / kPacketNodeCopies);
    s.tlas_oct = TL->d_tlas_oct.ptr;
    s.blas_oct = S->d_blas_oct.ptr;
    s.tri_packets = S->d_packets.ptr;
    s.triangles = S->d_triangles.ptr;
    const Instance::Tables& tb = S->tables[S->tables_version % Instance::kTableVersions];
    s.materials = tb.materials.ptr;
    s.area_lights = tb.area.ptr;
    s.point_lights = tb.point.ptr;
    s.spot_lights = tb.spot.ptr;
    s.directional_lights = tb.dir.ptr;
    s.tex_data = S->d_tex_data.ptr;
    s.tex_desc = S->d_tex_desc.ptr;
    s.n_textures = S->n_textures;
    s.skybox = S->skybox_desc;
    s.blue_noise = S->has_blue_noise ? S->d_blue_noise.ptr : nullptr;
    s.spill = I->d_spill.ptr;
    s.spill_stride = spill_stride(I);
    s.spill_rows = std::min<uint32_t>(scene_of(I)->spill_rows, (uint32_t)kStackSpill);
    s.overflow_flag = I->overflow_dev;
    s.counters = I->d_counters.ptr;
    return s;
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
    const size_t n = m.tris.size();
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
    build_bvh4_host(boxes, I->sah_max_leaf, I->build_threads, m.bvh, I->sah_trav_cost);
    m.packets.resize(n);
    for (size_t k = 0; k < n; k++) {
        const uint32_t id = m.bvh.prim_order[k];
        const rfw_rt_triangle& t = m.tris[id];
        TriPacket p;
        p.v0x = t.vertex0.x; p.v0y = t.vertex0.y; p.v0z = t.vertex0.z;
        p.tri_id = id; // mesh-local; the global offset is added when the mega-buffer is assembled
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
        r.tri_count = (uint32_t)src.tris.size();
        r.node_base = node_total;
        r.node_count = std::max<uint32_t>(r.tri_count, 1u);
        tri_total += r.tri_count;
        node_total += r.node_count;
        d.record = (uint32_t)I->mesh_records.size();
        I->mesh_records.push_back(r);
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

// One static mesh on the device, into the region its record names: boxes -> BVH (binned SAH, or LBVH) -> leaf-ordered packets ->
// quantised nodes.  The triangles are already in d_triangles.  `quantise_count` nodes of the region are quantised (the region is sized for
// the worst case, one node per primitive; nodes behind the tree's own are never referenced).
int build_mesh_device(Instance* I, uint32_t q, uint32_t quantise_count)
{
    const MeshRecord& r = I->mesh_records[q];
    if (r.tri_count == 0) return RFW_HIP_OK;
    launch_triangle_boxes(I->stream, I->d_triangles.ptr + r.tri_base, r.tri_count, I->d_tri_boxes.ptr);
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
    launch_make_packets(I->stream, I->d_triangles.ptr + r.tri_base, I->d_blas_order.ptr + r.tri_base, r.tri_count, r.tri_base, I->d_packets.ptr + r.tri_base);
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
                    launch_triangle_boxes(L.s, I->d_triangles.ptr + r.tri_base, r.tri_count, L.boxes.ptr);
                    const hipError_t e = sah_build(L.s, L.boxes.ptr, r.tri_count, L.ws.ptr, L.ws.cap, I->d_blas_raw.ptr + r.node_base, I->d_blas_order.ptr + r.tri_base,
                                                   I->d_mesh_node_counts.ptr + q, I->sah_max_leaf, I->sah_trav_cost);
                    if (e == hipErrorInvalidValue) { redo[q] = 1; continue; } // deeper than the builder's level budget: LBVH, below
                    if (e != hipSuccess) { lane_err[k] = e; return; }
                    launch_make_packets(L.s, I->d_triangles.ptr + r.tri_base, I->d_blas_order.ptr + r.tri_base, r.tri_count, r.tri_base, I->d_packets.ptr + r.tri_base);
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
        logical += I->mesh_records[it->second].tri_count;
    }
    for (auto& kv : I->derived) {
        I->mesh_records[kv.second.record].tri_logical = logical;
        logical += I->mesh_records[kv.second.record].tri_count;
    }
}

// BLAS for every mesh on the device: lay the mega-buffers out afresh, upload all triangles, build every mesh
int build_blas_device_full(Instance* I)
{
    I->mesh_records.clear();
    I->mesh_index.clear();
    I->record_tri_cap.clear();
    uint32_t tri_total = 0, node_total = 0;
    for (auto& kv : I->meshes) {
        MeshRecord r;
        std::memset(&r, 0, sizeof(r));
        r.tri_base = tri_total;
        r.tri_count = (uint32_t)kv.second.tris.size();
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
    HIP_TRY(I, follow_copies(I->d_blas_wide, I->d_blas_oct, I->d_blas_nodes, 0, I->stream));
    HIP_TRY(I, I->d_blas_raw.ensure(node_total));
    HIP_TRY(I, I->d_blas_order.ensure(tri_total));
    for (auto& ev : I->ev_build)
        if (!ev) HIP_TRY(I, hipEventCreate(&ev));
    HIP_TRY(I, hipEventRecord(I->ev_build[0], I->stream));
    uint32_t max_n = 0;
    size_t k = 0;
    for (auto& kv : I->meshes) {
        const MeshRecord& r = I->mesh_records[k++];
        max_n = std::max(max_n, r.tri_count);
        if (r.tri_count)
            HIP_TRY(I, hipMemcpyAsync(I->d_triangles.ptr + r.tri_base, kv.second.tris.data(), (size_t)r.tri_count * sizeof(rfw_rt_triangle),
                                      hipMemcpyHostToDevice, I->stream));
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
        if (I->blas_sah_on_device && n_static >= 4 && static_tris > 0 && !getenv("RFW_NO_FOREST")) {
            std::vector<ForestTree> trees(n_static);
            for (size_t q = 0; q < n_static; q++) trees[q] = ForestTree{I->mesh_records[q].tri_base, I->mesh_records[q].tri_count, I->mesh_records[q].node_base, 0u};
            HIP_TRY(I, I->d_forest.ensure(n_static));
            HIP_TRY(I, I->pins.upload(I->d_forest.ptr, trees.data(), n_static * sizeof(ForestTree), I->stream));
            HIP_TRY(I, I->d_tri_boxes.ensure(std::max(static_tris, I->max_derived_tris)));
            HIP_TRY(I, I->d_sah_ws.ensure(sah_forest_workspace_bytes(static_tris, (uint32_t)n_static)));
            launch_triangle_boxes(I->stream, I->d_triangles.ptr, static_tris, I->d_tri_boxes.ptr);
            const hipError_t fe = sah_build_forest(I->stream, I->d_tri_boxes.ptr, static_tris, I->d_forest.ptr, (uint32_t)n_static, max_n, I->d_sah_ws.ptr, I->d_sah_ws.cap,
                                                   I->d_blas_raw.ptr, I->d_blas_order.ptr, I->d_mesh_node_counts.ptr, I->sah_max_leaf, I->sah_trav_cost);
            if (fe == hipSuccess) {
                launch_make_packets(I->stream, I->d_triangles.ptr, I->d_blas_order.ptr, static_tris, 0u, I->d_packets.ptr); // global positions: one launch
                launch_forest_relative_order(I->stream, I->d_blas_order.ptr, static_tris, I->d_forest.ptr, (uint32_t)n_static);
                forest_done = true;
            } else if (fe != hipErrorInvalidValue) {
                HIP_TRY(I, fe);
            } // else: some tree is deeper than the builder's level budget: mesh by mesh, where LBVH can take over for that one
        }
        if (!forest_done && (rc = build_meshes(I, all, false))) return rc;
    }
    if ((rc = upload(I, I->d_mesh_records, I->mesh_records.data(), I->mesh_records.size()))) return rc;
    // all static regions in one launch; the slots behind a tree's last node are skipped (the builders left the node counts on the device)
    launch_quantize_regions(I->stream, I->d_blas_raw.ptr, I->d_blas_nodes.ptr, copies_of(I->d_blas_wide, I->d_blas_oct), static_nodes, I->d_mesh_records.ptr,
                            I->d_mesh_node_counts.ptr, (uint32_t)n_static);
    HIP_TRY(I, hipGetLastError());
    I->n_tris = tri_total;
    HIP_TRY(I, hipEventRecord(I->ev_build[2], I->stream));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    I->build_events_pending = true;
    I->blas_upload_bytes = (uint64_t)static_tris * sizeof(rfw_rt_triangle);
    I->blas_kernel_bytes = kernel_bytes;
    // nodes actually in use (the regions are sized for the worst case, one node per primitive); skinned copies count at their worst case
    std::vector<uint32_t> counts(n_static, 0u);
    if (n_static) HIP_TRY(I, hipMemcpy(counts.data(), I->d_mesh_node_counts.ptr, n_static * 4, hipMemcpyDeviceToHost));
    I->n_blas_nodes = node_total - static_nodes;
    for (uint32_t c : counts) I->n_blas_nodes += c;
    I->node_counts_stale = false;
    I->d_sah_ws.release(); // ~350 B per triangle of build scratch: not kept between scene changes
    for (auto& L : I->lanes) { L.ws.release(); L.boxes.release(); }
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
        const uint32_t n = (uint32_t)m.tris.size();
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
        HIP_TRY(I, follow_copies(I->d_blas_wide, I->d_blas_oct, I->d_blas_nodes, nodes_before, I->stream));
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
    uint64_t upload_bytes = 0, kernel_bytes = 0;
    for (const uint32_t q : todo) { // the changed meshes' triangles first (their regions are disjoint) ...
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
    if ((rc = build_meshes(I, todo, true))) return rc; // ... then their trees
    HIP_TRY(I, hipEventRecord(I->ev_build[2], I->stream));
    I->build_events_pending = true;
    I->blas_upload_bytes = upload_bytes;
    I->blas_kernel_bytes = kernel_bytes;
    HIP_TRY(I, hipGetLastError());
    HIP_TRY(I, I->pins.upload(I->d_mesh_records.ptr, I->mesh_records.data(), I->mesh_records.size() * sizeof(MeshRecord), I->stream));
    uint64_t live = 0;
    for (auto& kv : I->mesh_index) live += I->mesh_records[kv.second].tri_count;
    I->n_tris = live;
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
        r.tri_count = (uint32_t)m.tris.size();
        I->mesh_index[kv.first] = (uint32_t)I->mesh_records.size();
        I->mesh_records.push_back(r);
        nodes.insert(nodes.end(), m.bvh.nodes.begin(), m.bvh.nodes.end());
        const size_t p0 = packets.size();
        packets.insert(packets.end(), m.packets.begin(), m.packets.end());
        for (size_t k = p0; k < packets.size(); k++) packets[k].tri_id += r.tri_base; // global triangle id
        tris.insert(tris.end(), m.tris.begin(), m.tris.end());
    }
    uint32_t tri_total = (uint32_t)tris.size(), node_total = (uint32_t)nodes.size();
    int rc;
    if ((rc = layout_derived(I, tri_total, node_total))) return rc;
    assign_logical_ids(I);
    I->n_tris = tri_total;
    I->n_blas_nodes = node_total;
    HIP_TRY(I, I->d_blas_nodes.ensure(node_total)); // room for the skinned copies behind the static meshes
    HIP_TRY(I, follow_copies(I->d_blas_wide, I->d_blas_oct, I->d_blas_nodes, 0, I->stream));
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
    HIP_TRY(I, follow_copies(T->d_tlas_wide, T->d_tlas_oct, T->d_tlas_nodes, 0, T->stream));
    HIP_TRY(I, T->d_tlas_raw.ensure(std::max<size_t>(n_valid, 1)));
    HIP_TRY(I, T->d_node_count.ensure(1));
    hipStream_t s = T->stream;
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
        if ((rc = upload(I, T->d_tlas_nodes, qn.data(), qn.size()))) return rc;
        HIP_TRY(I, follow_copies(T->d_tlas_wide, T->d_tlas_oct, T->d_tlas_nodes, 0, s));
        launch_expand_nodes(s, T->d_tlas_nodes.ptr, copies_of(T->d_tlas_wide, T->d_tlas_oct), 0u, (uint32_t)qn.size());
        if ((rc = upload(I, T->d_tlas_prims, prims.data(), prims.size()))) return rc;
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

CameraParams camera_params(const Instance* I, const rfw_camera_view_3d& v, uint32_t sub = 0)
{
    CameraParams c;
    std::memset(&c, 0, sizeof(c));
    c.pos[0] = v.pos.x; c.pos[1] = v.pos.y; c.pos[2] = v.pos.z;
    c.lens_size = v.lens_size;
    c.right[0] = v.right.x; c.right[1] = v.right.y; c.right[2] = v.right.z;
    c.spread_angle = v.spread_angle;
    c.up[0] = v.up.x; c.up[1] = v.up.y; c.up[2] = v.up.z;
    c.clamp_value = I->clamp_value;
    c.p1[0] = v.p1.x; c.p1[1] = v.p1.y; c.p1[2] = v.p1.z;
    c.width = I->width; c.height = I->height;
    c.sample_count = I->sample_count;
    const Instance* S = scene_of(I);
    c.point_light_count = (uint32_t)S->point_lights.size();
    c.area_light_count = (uint32_t)S->area_lights.size();
    c.spot_light_count = (uint32_t)S->spot_lights.size();
    c.directional_light_count = (uint32_t)S->directional_lights.size();
    c.tile_size = I->tile_size; c.tiles_x = I->tiles_x; c.tiles_y = I->tiles_y;
    c.rank = I->rank * I->substreams + sub; c.world = I->world * I->substreams; c.local_tiles = I->local_tiles_v;
    c.flags = I->flags;
    c.max_path_length = I->max_path_length;
    c.sky[0] = I->sky[0]; c.sky[1] = I->sky[1]; c.sky[2] = I->sky[2];
    c.batch = 1;
    // streaming trades the tail of every wavefront for fewer, longer wavefronts: it pays when other frames fill the chip meanwhile (measured on
    // C4 path traced: 8 frame slots 2880 -> 3190 Mrays/s, 4 slots 2840 -> 3100; but 2 slots 2770 -> 2610, one frame at a time 2270 -> 1670,
    // and batches, whose extension rays are traced in sorted order, 3480 -> 3360).  Unless the option was set by hand it is on for single
    // frames of an instance with four or more frame slots (and for batches, which fill the chip by themselves: do_render)
    const Instance* S_ = scene_of(I);
    c.stream_run = (S_->stream_auto && !(S_->slots.size() + 1 >= 4)) ? 0u : S_->stream_run;
    c.stream_refill = std::max(1u, std::min(64u, scene_of(I)->stream_refill)) | (std::max(1u, std::min(64u, scene_of(I)->stream_leaf_gate)) << 8);
    c.frame_capacity = I->cap_v;
    return c;
}

PathDev path_dev(Instance* I, uint32_t sub = 0)
{
    PathDev p;
    const size_t off = (size_t)sub * I->cap_v;
    for (int h = 0; h < 2; h++) {
        p.ray_o[h] = I->d_ray_o[h].ptr + off;
        p.ray_d[h] = I->d_ray_d[h].ptr + off;
        p.thr[h] = I->d_thr[h].ptr + off;
        p.hit[h] = I->d_hit[h].ptr + off;
    }
    p.sh_o = I->d_sh_o.ptr + off * kShadowBuckets;
    p.sh_d = I->d_sh_d.ptr + off * kShadowBuckets;
    p.sh_e = I->d_sh_e.ptr + off * kShadowBuckets;
    p.acc = I->d_acc_slab.ptr + off;
    p.capacity = I->cap_v;
    return p;
}

inline int ev_index(uint32_t bounce, int kernel, int end) { return EV_KERNEL_BASE + 2 * ((int)bounce * kKernelsPerBounce + kernel) + end; }
constexpr int kEvBlit = EV_KERNEL_BASE + 2 * (kMaxBounces * kKernelsPerBounce);

hipEvent_t* ring_events(Instance* I, int slot, uint32_t sub) { return I->ring.data() + ((size_t)slot * I->substreams + sub) * kNumEvents; }

// k == 1: one sample of the image for views[0].  k > 1 (rfw_hip_render_batch): k independent NEW images, one per view, traced as one
// tall virtual frame — every stage is ONE launch over the paths of all k frames.
// `samples` (rfw_hip_render_samples): the k frames are k consecutive SAMPLES of the one image of views[0] — sample indices sample_count …
// sample_count + k - 1, each traced into its own slab, then summed into slab 0 in sample order.
// ---- sharded frame: what a rank sends, and what it does with what it receives
inline uint64_t slab_words(const Instance* I) // 4-byte words one frame of this rank contributes to the all-gather
{
    const uint64_t c = I->capacity;
    const uint32_t f = scene_of(I)->gather_format;
    return f == 0 ? c * 3u : (f == 1 ? c * 3u / 2u : c);
}
const float* srgb_steps()
{
    static float t[255];
    static std::once_flag once;
    std::call_once(once, [] {
        for (int k = 0; k < 255; k++) {
            const double e = (k + 0.5) / 255.0;
            const double lin = e <= 0.04045 ? e / 12.92 : std::pow((e + 0.055) / 1.055, 2.4);
            float f = (float)lin;
            if ((double)f < lin) f = std::nextafter(f, 2.0f); // smallest float NOT below the exact step
            t[k] = f;
        }
    });
    return t;
}
void pack_slabs(Instance* I, hipStream_t s, void* dst, uint32_t frames)
{
    const uint64_t n = (uint64_t)I->capacity * frames;
    const uint32_t f = scene_of(I)->gather_format;
    if (f == 0) launch_pack_rgb(s, I->d_acc_slab.ptr, (float*)dst, n);
    else launch_pack_finished(s, I->d_acc_slab.ptr, dst, n, std::max(1u, I->sample_count), f, srgb_steps());
}
// gathered = [rank][frame][slab] in the instance's gather format -> the row-major frame(s)
int assemble_gathered(Instance* I, hipStream_t s, const void* gathered, uint32_t k, uint32_t samples)
{
    CameraParams cam = camera_params(I, I->last_view);
    cam.batch = k;
    const uint32_t fmt = scene_of(I)->gather_format;
    if (fmt == 0) {
        launch_assemble(s, cam, gathered, true, false, I->cap_v, I->d_frame_out.ptr, samples);
        I->acc_source = gathered; I->acc_source_rgb = true; I->acc_source_batch = k;
    } else if (fmt == 1) {
        launch_assemble_finished(s, cam, gathered, I->cap_v, 1u, I->d_frame_out.ptr, nullptr);
        I->acc_source = nullptr;
    } else {
        HIP_TRY(I, I->d_present.ensure((size_t)I->width * I->height * I->max_batch));
        launch_assemble_finished(s, cam, gathered, I->cap_v, 2u, nullptr, I->d_present.ptr);
        I->acc_source = nullptr;
        I->presented_valid = true;
    }
    I->deferred = Instance::Deferred();
    HIP_TRY(I, hipGetLastError());
    return RFW_HIP_OK;
}
// does this instance receive other ranks' tiles in the gather format (whoever moves them)?
bool gathers_tiles(const Instance* I) { return scene_of(I)->comm != nullptr || I->external_slab != nullptr || scene_of(I)->p2p.connected; }
bool p2p_timed_out(const Instance* I) { return scene_of(I)->p2p.connected && I->overflow_host && ((volatile const uint32_t*)I->overflow_host)[1] != 0u; }
// The frame's exchange by stores into the peers' buffers (include/rfw_hip.h, rfw_hip_p2p_*).  Destinations: the presenting rank, or all.
int p2p_exchange(Instance* I, hipStream_t s, uint32_t frames)
{
    Instance* C = scene_of(I);
    Instance::P2P& P = C->p2p;
    const uint32_t W = I->world, me = I->rank, slot = I->slot_index;
    const int pr = C->present_rank;
    if (pr >= (int)W) return fail(I, RFW_HIP_E_INVALID, "present_rank is not a rank of this world");
    const uint32_t d0 = pr >= 0 ? (uint32_t)pr : 0u, nd = pr >= 0 ? 1u : W;
    const bool receiver = pr < 0 || (uint32_t)pr == me;
    const uint32_t seq = ++I->p2p_seq;
    uint32_t* timeout_flag = I->overflow_dev + 1;
    uint32_t* my_flags = P.flags + (size_t)slot * 2u * W; // arrived[W], credit[W]
    // 1. the destinations are done with what this slot sent last time
    launch_p2p_wait(s, my_flags + W, d0, nd, seq - 1u, P.timeout_ticks, timeout_flag);
    // 2. this rank's slab(s), packed where the destination's de-tiling reads them: [slot][rank][frame][slab], densely
    const size_t at = (size_t)slot * P.slot_words + (size_t)me * frames * slab_words(I);
    P2PTargets t;
    for (uint32_t d = d0; d < d0 + nd; d++) {
        pack_slabs(I, s, P.peer_data[d] + at, frames);
        t.p[d - d0] = P.peer_flags[d] + (size_t)slot * 2u * W + me;
    }
    // 3. ... and say so (behind the pack kernels on this stream)
    launch_p2p_signal(s, t, nd, seq);
    I->frame_elsewhere = !receiver;
    I->acc_source = nullptr;
    I->presented_valid = false;
    I->deferred = Instance::Deferred();
    if (receiver) {
        launch_p2p_wait(s, my_flags, 0u, W, seq, P.timeout_ticks, timeout_flag);
        const int arc = assemble_gathered(I, s, P.data + (size_t)slot * P.slot_words, frames, std::max(1u, I->sample_count));
        if (arc != RFW_HIP_OK) return arc;
        for (uint32_t q = 0; q < W; q++) t.p[q] = P.peer_flags[q] + (size_t)slot * 2u * W + W + me; // credit[me] at every sender
        launch_p2p_signal(s, t, W, seq);
    }
    HIP_TRY(I, hipGetLastError());
    return RFW_HIP_OK;
}
// after a gather: de-tile now (this rank presents, or every rank does), or remember where the tiles are
int gathered_arrived(Instance* I, hipStream_t s, const void* gathered, uint32_t k)
{
    const uint32_t samples = std::max(1u, I->sample_count);
    const int pr = scene_of(I)->present_rank;
    if (pr < 0 || (uint32_t)pr == I->rank) return assemble_gathered(I, s, gathered, k, samples);
    I->deferred.gathered = gathered; I->deferred.k = k; I->deferred.samples = samples;
    I->acc_source = nullptr;
    I->presented_valid = false;
    return RFW_HIP_OK;
}
int ensure_assembled(Instance* I)
{
    if (I->frame_elsewhere) return fail(I, RFW_HIP_E_STATE, "this rank sent its tiles to the presenting rank (present_rank): the frame exists there only");
    if (p2p_timed_out(I)) return fail(I, RFW_HIP_E_DEVICE, "p2p exchange: a peer's flag did not arrive within p2p_timeout_ms (the frame is incomplete)");
    if (!I->deferred.gathered) return RFW_HIP_OK;
    return assemble_gathered(I, I->stream, I->deferred.gathered, I->deferred.k, I->deferred.samples);
}

int do_render(Instance* I, const rfw_camera_view_3d* views, uint32_t k = 1, bool samples = false)
{
    const rfw_camera_view_3d& view = views[0];
    HIP_TRY(I, hipSetDevice(I->device));
    if (!scene_of(I)->synchronized || tlas_of(I)->d_tlas_nodes.ptr == nullptr) return RFW_HIP_OK; // render before any mesh exists (gpu-rt/src/lib.rs:1686-1688)
    if (I->scene && I->scene->scene_ready && I->waited_version != I->scene->scene_version) { // a slot must not read a scene still being written
        HIP_TRY(I, hipStreamWaitEvent(I->stream, I->scene->scene_ready, 0));
        I->waited_version = I->scene->scene_version;
    }
    {   // the material / light tables this frame reads: wait (on the device) for their upload, once per version
        Instance* S = scene_of(I);
        if (S->tables_ready && I->tables_waited != S->tables_version) {
            HIP_TRY(I, hipStreamWaitEvent(I->stream, S->tables_ready, 0));
            I->tables_waited = S->tables_version;
        }
        I->tables_used = S->tables_version;
        if (I->tables_oldest_pending == ~0ull) I->tables_oldest_pending = S->tables_version;
    }
    if ((I->have_last_view && std::memcmp(&I->last_view, &view, sizeof(view)) != 0) || I->after_batch) I->sample_count = 0;
    I->after_batch = k > 1 && !samples; // the frames of a batch are complete images: whatever follows starts a new one
    I->last_view = view;
    I->have_last_view = true;
    if (k > 1) {
        if (k > I->max_batch || k > (uint32_t)kMaxBatch) return fail(I, RFW_HIP_E_INVALID, "render_batch: more frames than options.max_batch");
        if (I->substreams > 1) return fail(I, RFW_HIP_E_STATE, "render_batch: not available with sub-streams");
        // the frame index rides in bits 24..31 of the path word, next to the PIXEL index of the whole frame (not of this rank's slab)
        if ((uint64_t)I->width * I->height >= (1ull << 24)) return fail(I, RFW_HIP_E_INVALID, "render_batch: frames of 2^24 pixels or more cannot be batched");
        for (uint32_t f = 1; f < k; f++)
            if (views[f].spread_angle != view.spread_angle) return fail(I, RFW_HIP_E_INVALID, "render_batch: the views of a batch must share one spread angle (field of view and height)");
        if (!samples) I->sample_count = 0; // every frame of a batch is a new image
    }

    const uint32_t S = I->substreams;
    const bool count = (I->flags & RFW_HIP_FLAG_COUNT_TRAVERSAL) != 0;
    const bool nee = !(I->flags & RFW_HIP_FLAG_NO_NEE);
    hipStream_t main = I->stream;
    const bool tm = I->timing;
    const int slot = (int)(I->frame_index % kTimingRing);
    I->events = ring_events(I, slot, 0);
    const uint32_t bounces = std::min<uint32_t>(I->max_path_length, kMaxBounces);

    if (tm) (void)hipEventRecord(I->events[EV_FRAME0], main);
    HIP_TRY(I, hipMemsetAsync(I->d_counters.ptr, 0, S * sizeof(QueueCounters), main));
    // The frame's tiles are dealt to S sub-shards, each with its own queues, counters and accumulator slab, each traced on its
    // own stream: the long tail of one sub-shard's trace kernel (the slowest wavefront bounds a launch) overlaps the other
    // sub-shards' kernels.  Fork from / join into the caller's stream with events.
    if (S > 1) HIP_TRY(I, hipEventRecord(I->ev_fork, main));
    SceneDev sc[kMaxSub];
    PathDev p[kMaxSub];
    CameraParams cam[kMaxSub];
    hipStream_t st[kMaxSub];
    for (uint32_t s = 0; s < S; s++) {
        st[s] = S > 1 ? I->sub[s] : main;
        if (S > 1) HIP_TRY(I, hipStreamWaitEvent(st[s], I->ev_fork, 0));
        sc[s] = scene_dev(I);
        sc[s].counters = I->d_counters.ptr + s;
        sc[s].spill = I->d_spill.ptr + (size_t)s * (I->cap_v + kSpillMargin);
        p[s] = path_dev(I, s);
        cam[s] = camera_params(I, view, s);
    }
    BatchViews bv;
    if (k > 1) {
        cam[0].batch = k;
        // a batch fills the chip by itself: its bounces stream whatever the number of frame slots (and are then NOT sorted, see below)
        if (scene_of(I)->stream_auto) cam[0].stream_run = scene_of(I)->stream_run;
        p[0].capacity = I->cap_v * k;
        for (uint32_t f = 0; f < k; f++) {
            cam[0].batch_sample[f] = samples ? I->sample_count + f : 0u;
            FrameView& v = bv.v[f];
            v.pos[0] = views[f].pos.x; v.pos[1] = views[f].pos.y; v.pos[2] = views[f].pos.z; v.lens_size = views[f].lens_size;
            v.right[0] = views[f].right.x; v.right[1] = views[f].right.y; v.right[2] = views[f].right.z; v.pad0 = 0.0f;
            v.up[0] = views[f].up.x; v.up[1] = views[f].up.y; v.up[2] = views[f].up.z; v.pad1 = 0.0f;
            v.p1[0] = views[f].p1.x; v.p1[1] = views[f].p1.y; v.p1[2] = views[f].p1.z; v.pad2 = 0.0f;
        }
    }
    for (uint32_t b = 0; b < bounces; b++) { // gpu-rt/src/lib.rs:1708-1728 without the read-back; stage by stage across the sub-shards
        for (uint32_t s = 0; s < S; s++) {
            hipEvent_t* ev = ring_events(I, slot, s);
            cam[s].path_length = b;
            if (tm) (void)hipEventRecord(ev[ev_index(b, 0, 0)], st[s]);
            if (b == 0 && k > 1) launch_primary_batch(st[s], cam[s], bv, sc[s], p[s], count);
            else if (b == 0) launch_primary(st[s], cam[s], sc[s], p[s], count);
            else {
                const uint32_t* order = nullptr;
                // Measured on C4, max path length 3 (EXPERIMENTS.md): a single 1-spp frame has too few rays per cell and direction for the sort
                // to form coherent wavefronts (+1.5 % with frames in flight, -2.7 % alone: it costs a 2 M-pair sort per bounce); a batch of 8
                // frames — or k samples of one image — sorts 8 x / k x as many rays of the same surfaces together: +17 %
                // Round 3: streaming (traverse_stream) does for a batch what the sort does, without the sort — batches of 8: 3480 sorted, 3490
                // streaming, 3300 both; 4 samples of one image per call: 3070 sorted, 3415 streaming, 3130 both — so "only where it pays"
                // (mode 2) now means: batches whose bounces do NOT stream
                const int mode = scene_of(I)->sort_extension_rays;
                if ((mode == 1 || (mode == 2 && k > 1 && cam[s].stream_run == 0u)) && S == 1) { // (sub-shards keep the queue order: one sort buffer per instance)
                    const size_t n = p[s].capacity;
                    for (int q = 0; q < 2; q++) { HIP_TRY(I, I->d_sort_keys[q].ensure(n)); HIP_TRY(I, I->d_sort_vals[q].ensure(n)); }
                    HIP_TRY(I, I->d_sort_ws.ensure(sort_pairs_workspace_bytes((uint32_t)n)));
                    launch_extension_keys(st[s], sc[s], p[s], b, I->d_sort_keys[0].ptr, I->d_sort_vals[0].ptr);
                    HIP_TRY(I, sort_pairs_u32(st[s], I->d_sort_ws.ptr, I->d_sort_ws.cap, I->d_sort_keys[0].ptr, I->d_sort_keys[1].ptr, I->d_sort_vals[0].ptr,
                                              I->d_sort_vals[1].ptr, (uint32_t)n, 32));
                    order = I->d_sort_vals[1].ptr;
                }
                launch_extend(st[s], cam[s], sc[s], p[s], b, count, order);
            }
            if (tm) (void)hipEventRecord(ev[ev_index(b, 0, 1)], st[s]);
        }
        for (uint32_t s = 0; s < S; s++) {
            hipEvent_t* ev = ring_events(I, slot, s);
            if (tm) (void)hipEventRecord(ev[ev_index(b, 1, 0)], st[s]);
            launch_shade(st[s], cam[s], sc[s], p[s], b);
            if (tm) (void)hipEventRecord(ev[ev_index(b, 1, 1)], st[s]);
        }
        if (nee)
            for (uint32_t s = 0; s < S; s++) {
                hipEvent_t* ev = ring_events(I, slot, s);
                if (tm) (void)hipEventRecord(ev[ev_index(b, 2, 0)], st[s]);
                launch_shadow(st[s], cam[s], sc[s], p[s], b, count);
                if (tm) (void)hipEventRecord(ev[ev_index(b, 2, 1)], st[s]);
            }
    }
    if (S > 1)
        for (uint32_t s = 0; s < S; s++) {
            HIP_TRY(I, hipEventRecord(I->ev_join[s], st[s]));
            HIP_TRY(I, hipStreamWaitEvent(main, I->ev_join[s], 0));
        }
    if (samples && k > 1) { // the k sample slabs -> the image's accumulator (slab 0), in sample order
        launch_sum_batch(main, I->d_acc_slab.ptr, I->cap_v, k);
        cam[0].batch = 1;
    }
    I->sample_count += samples ? k : 1;
    const uint32_t frames_out = samples ? 1u : k; // images this call leaves behind
    if (tm) (void)hipEventRecord(I->events[kEvBlit], main);
    if (Instance* C = scene_of(I); C->comm) {
        // the frame's ONE collective, issued by the library itself: this rank's slab(s) -> all ranks (RCCL over xGMI) -> de-tile
        const uint64_t n_send = slab_words(I) * frames_out; // 4-byte words, whatever they hold
        pack_slabs(I, main, I->d_send.ptr, frames_out);
        if (C->comm_chain && C->comm_chain_pending) HIP_TRY(I, hipStreamWaitEvent(main, C->comm_chain, 0)); // behind the previous slot's collective
        const ncclResult_t nr = g_rccl.all_gather(I->d_send.ptr, I->d_recv.ptr, n_send, ncclFloat, C->comm, main);
        if (nr != ncclSuccess) return fail(I, RFW_HIP_E_DEVICE, std::string("ncclAllGather: ") + g_rccl.error_string(nr));
        if (C->comm_chain) { HIP_TRY(I, hipEventRecord(C->comm_chain, main)); C->comm_chain_pending = true; }
        const int arc = gathered_arrived(I, main, I->d_recv.ptr, frames_out); // gathered = [rank][frame][slab]
        if (arc != RFW_HIP_OK) return arc;
    } else if (C->p2p.connected) {
        const int prc = p2p_exchange(I, main, frames_out);
        if (prc != RFW_HIP_OK) return prc;
    } else if (I->world <= 1) // de-tile the sub-slabs into the linear accumulator / tonemapped frame (blit.comp:15-23)
    {
        launch_assemble(main, cam[0], I->d_acc_slab.ptr, false, false, I->cap_v, I->d_frame_out.ptr, I->sample_count);
        I->acc_source = I->d_acc_slab.ptr; I->acc_source_rgb = false; I->acc_source_batch = frames_out;
    }
    if (I->external_slab) // this rank's contribution to the all-gather, [frame][sub-shard][slot] in the instance's gather format
        pack_slabs(I, main, I->external_slab, frames_out);
    if (tm) (void)hipEventRecord(I->events[kEvBlit + 1], main);
    if (tm) (void)hipEventRecord(I->events[EV_FRAME1], main);
    HIP_TRY(I, hipGetLastError());
    I->last_bounces = bounces;
    I->ring_bounces[slot] = tm ? bounces : 0;
    I->ring_nee[slot] = nee;
    I->frame_index++;
    if (I->frame_index - I->drained_index > kTimingRing) I->drained_index = I->frame_index - kTimingRing;
    I->frame_recorded = tm;
    I->last_count_flag = count;
    if (I->frame_done) HIP_TRY(I, hipEventRecord(I->frame_done, main));
    return RFW_HIP_OK;
}

} // namespace

// ==================================================================== C ABI
#define LOCK(inst)                                    \
    if (!(inst)) return RFW_HIP_E_INVALID;            \
    Instance* I = static_cast<Instance*>(inst);       \
    std::lock_guard<std::mutex> guard_(I->mu)

extern "C" {

uint32_t rfw_hip_abi_version(void) { return RFW_HIP_ABI_VERSION; }

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
    if (out_nodes) *out_nodes = (uint32_t)bvh.nodes.size();
    return errors;
}

void* rfw_hip_create(uint32_t width, uint32_t height, double /*scale*/, const rfw_hip_options* o)
{
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0) {
        g_create_error = std::string("no HIP device available (") + (e == hipSuccess ? "0 devices" : hipGetErrorString(e)) +
                         "); this backend has no CPU fallback";
        return nullptr;
    }
    if (width == 0 || height == 0) {
        g_create_error = "width and height must be non-zero";
        return nullptr;
    }
    Instance* I = new Instance();
    I->width = width;
    I->height = height;
    int dev = -1;
    uint32_t n_slots = 1;
    if (o) {
        dev = o->device;
        if (o->max_path_length) I->max_path_length = std::min<uint32_t>(o->max_path_length, kMaxBounces);
        if (o->clamp_value > 0.0f) I->clamp_value = o->clamp_value;
        I->world = std::max<uint32_t>(o->world, 1);
        I->rank = o->rank;
        if (o->tile_size) I->tile_size = o->tile_size;
        if (o->builder) I->builder = o->builder;
        I->flags = o->flags & 15u; // the public RFW_HIP_FLAG_* bits; the others are internal (set_option)
        if (o->streams) I->substreams = std::min<uint32_t>(o->streams, kMaxSub);
        if (o->struct_size >= offsetof(rfw_hip_options, frames_in_flight) + sizeof(uint32_t)) n_slots = std::min<uint32_t>(std::max<uint32_t>(o->frames_in_flight, 1u), 16u);
        if (o->struct_size >= offsetof(rfw_hip_options, max_batch) + sizeof(uint32_t)) I->max_batch = std::min<uint32_t>(std::max<uint32_t>(o->max_batch, 1u), (uint32_t)kMaxBatch);
    }
    {
        const char* e = getenv("RFW_PACKET_TRACE"); // A/B runs: the default of option "packet_trace"
        const int pt = e ? atoi(e) : kDefaultPacketTrace;
        if (pt & 1) I->flags |= kFlagPacketPrimary;
        if (pt & 2) I->flags |= kFlagPacketShadow;
    }
    if (I->max_batch > 1 && I->substreams > 1) {
        g_create_error = "max_batch > 1 needs streams <= 1 (a batch already fills the device with one launch per stage)";
        delete I;
        return nullptr;
    }
    if (I->rank >= I->world || (I->tile_size % 8) != 0) {
        g_create_error = "invalid shard options (rank >= world, or tile_size not a multiple of 8)";
        delete I;
        return nullptr;
    }
    // AUTO: BLAS by binned SAH on the host cores (built once per mesh change, best traversal quality), TLAS by LBVH on the device
    // (rebuilt every synchronize()).  HOST_SAH / DEVICE_LBVH force one builder for both levels.
    // AUTO: meshes by binned SAH on the device (the host builder's tree quality at ~14x its speed), TLAS and skinned copies by LBVH
    // (no host round trip, so they can be rebuilt every frame without draining the stream)
    I->blas_on_device = I->builder != RFW_HIP_BUILDER_HOST_SAH;
    I->blas_sah_on_device = I->builder == RFW_HIP_BUILDER_DEVICE_SAH || I->builder == RFW_HIP_BUILDER_AUTO;
    I->tlas_on_device = I->builder != RFW_HIP_BUILDER_HOST_SAH;
    if (dev < 0) (void)hipGetDevice(&dev);
    I->device = dev;
    I->build_threads = (int)std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto bail = [&](const char* what, hipError_t err) -> void* {
        g_create_error = std::string(what) + ": " + hipGetErrorString(err);
        delete I;
        return nullptr;
    };
    if ((e = hipSetDevice(dev)) != hipSuccess) return bail("hipSetDevice", e);
    if ((e = hipStreamCreateWithFlags(&I->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    I->stream = I->own_stream;
    I->ring.assign((size_t)kTimingRing * I->substreams * kNumEvents, nullptr);
    for (auto& ev : I->ring)
        if ((e = hipEventCreate(&ev)) != hipSuccess) return bail("hipEventCreate", e);
    I->events = I->ring.data();
    if ((e = hipEventCreateWithFlags(&I->ev_fork, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    // HIP maps streams to a handful of hardware queues in creation order and two streams sharing a queue serialise, so no
    // stream is created that is not used: sub-shard streams only when the frame is actually split
    for (uint32_t k = 0; k < I->substreams && I->substreams > 1; k++) {
        if ((e = hipStreamCreateWithFlags(&I->sub[k], hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
        if ((e = hipEventCreateWithFlags(&I->ev_join[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    }
    for (int k = 0; k < Instance::kStages; k++)
        if ((e = hipEventCreateWithFlags(&I->stage_event[k], hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
    if ((e = hipHostMalloc((void**)&I->overflow_host, 64, hipHostMallocMapped)) != hipSuccess) return bail("hipHostMalloc (overflow flag)", e);
    std::memset(I->overflow_host, 0, 64); // word 0: traversal stack overflow, word 1: p2p timeout; hipHostMalloc does not zero, and a block may be recycled
    if ((e = hipHostGetDevicePointer((void**)&I->overflow_dev, I->overflow_host, 0)) != hipSuccess) return bail("hipHostGetDevicePointer", e);
    if (alloc_paths(I) != RFW_HIP_OK) {
        g_create_error = I->err;
        delete I;
        return nullptr;
    }
    (void)hipStreamSynchronize(I->stream);
    if (n_slots > 1) {
        if ((e = hipEventCreateWithFlags(&I->frame_done, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        if ((e = hipEventCreateWithFlags(&I->scene_ready, hipEventDisableTiming)) != hipSuccess) return bail("hipEventCreate", e);
        rfw_hip_options so;
        std::memset(&so, 0, sizeof(so));
        if (o) std::memcpy(&so, o, std::min<size_t>(o->struct_size ? o->struct_size : sizeof(so), sizeof(so)));
        so.struct_size = sizeof(so);
        so.device = I->device;
        so.frames_in_flight = 1;
        so.max_batch = I->max_batch;
        for (uint32_t k = 1; k < n_slots; k++) {
            Instance* c = static_cast<Instance*>(rfw_hip_create(width, height, 1.0, &so));
            if (!c) { // g_create_error is set
                for (Instance* d : I->slots) rfw_hip_destroy(d);
                I->slots.clear();
                rfw_hip_destroy(I);
                return nullptr;
            }
            c->scene = I;
            c->slot_index = k;
            if (hipEventCreateWithFlags(&c->frame_done, hipEventDisableTiming) != hipSuccess) {
                g_create_error = "hipEventCreate (frame slot)";
                rfw_hip_destroy(c);
                for (Instance* d : I->slots) rfw_hip_destroy(d);
                I->slots.clear();
                rfw_hip_destroy(I);
                return nullptr;
            }
            I->slots.push_back(c);
        }
    }
    return I;
}

void rfw_hip_destroy(void* inst)
{
    if (!inst) return;
    Instance* I = static_cast<Instance*>(inst);
    for (Instance* c : I->slots) rfw_hip_destroy(c); // frame slots first: they read this instance's scene
    I->slots.clear();
    {
        std::lock_guard<std::mutex> g(I->mu);
        (void)hipSetDevice(I->device);
        (void)hipDeviceSynchronize();
        if (I->scene_ready) (void)hipEventDestroy(I->scene_ready);
        if (I->frame_done) (void)hipEventDestroy(I->frame_done);
        if (I->download_done) (void)hipEventDestroy(I->download_done);
        I->d_blas_nodes.release(); I->d_tlas_nodes.release(); I->d_blas_wide.release(); I->d_tlas_wide.release(); I->d_blas_oct.release(); I->d_tlas_oct.release(); I->d_blas_raw.release(); I->d_tlas_raw.release(); I->d_packets.release(); I->d_triangles.release();
        I->d_mesh_records.release(); I->d_matrices.release(); I->d_mesh_of_instance.release(); I->d_tlas_prims.release();
        I->d_xforms.release(); I->d_normals.release();
        for (auto& tb : I->tables) { tb.materials.release(); tb.area.release(); tb.point.release(); tb.spot.release(); tb.dir.release(); }
        if (I->tables_ready) (void)hipEventDestroy(I->tables_ready);
        if (I->upload_stream) (void)hipStreamDestroy(I->upload_stream);
        I->d_spill.release(); I->d_counters.release(); I->d_tex_data.release(); I->d_tex_desc.release(); I->d_blue_noise.release();
        I->d_valid_gids.release(); I->d_tlas_order.release(); I->d_node_count.release(); I->d_inst_boxes.release(); I->d_mesh_local.release();
        I->d_tri_boxes.release(); I->d_lbvh_ws.release(); I->d_blas_order.release();
        I->d_q_o.release(); I->d_q_d.release(); I->d_q_t.release(); I->d_q_h.release(); I->d_q_depth.release(); I->d_q_r.release();
        if (I->comm) { (void)g_rccl.comm_destroy(I->comm); I->comm = nullptr; }
        if (I->comm_chain) (void)hipEventDestroy(I->comm_chain);
        for (auto& ev : I->ev_build)
            if (ev) (void)hipEventDestroy(ev);
        for (auto& L : I->lanes) {
            if (L.s) (void)hipStreamDestroy(L.s);
            if (L.done) (void)hipEventDestroy(L.done);
            L.ws.release(); L.boxes.release();
        }
        p2p_release(I);
        I->d_send.release(); I->d_recv.release();
        for (int q = 0; q < 2; q++) { I->d_sort_keys[q].release(); I->d_sort_vals[q].release(); }
        I->d_sort_ws.release();
        if (I->overflow_host) (void)hipHostFree(I->overflow_host);
        I->pins.release();
        I->d_skin_data.release(); I->d_joints.release(); I->d_bounds_scratch.release(); I->d_sah_ws.release(); I->d_mesh_node_counts.release(); I->d_forest.release(); I->d_refit_parent.release(); I->d_refit_nint.release(); I->d_refit_arrive.release();
        for (int k = 0; k < Instance::kStages; k++) {
            if (I->stage_buf[k]) (void)hipHostFree(I->stage_buf[k]);
            if (I->stage_event[k]) (void)hipEventDestroy(I->stage_event[k]);
        }
        for (int h = 0; h < 2; h++) { I->d_ray_o[h].release(); I->d_ray_d[h].release(); I->d_thr[h].release(); I->d_hit[h].release(); }
        I->d_sh_o.release(); I->d_sh_d.release(); I->d_sh_e.release(); I->d_acc_slab.release(); I->d_frame_acc.release(); I->d_frame_out.release(); I->d_present.release();
        for (auto& ev : I->ring)
            if (ev) (void)hipEventDestroy(ev);
        if (I->ev_fork) (void)hipEventDestroy(I->ev_fork);
        for (int k = 0; k < kMaxSub; k++) {
            if (I->ev_join[k]) (void)hipEventDestroy(I->ev_join[k]);
            if (I->sub[k]) (void)hipStreamDestroy(I->sub[k]);
        }
        if (I->own_stream) (void)hipStreamDestroy(I->own_stream);
    }
    delete I;
}

const char* rfw_hip_last_error(void* inst)
{
    if (!inst) return g_create_error.c_str();
    return static_cast<Instance*>(inst)->err.c_str();
}

int rfw_hip_set_2d_mesh(void* inst, uint32_t, const void*, uint32_t, int32_t) { LOCK(inst); return RFW_HIP_OK; }
int rfw_hip_set_2d_instances(void* inst, uint32_t, const rfw_mat4*, uint32_t) { LOCK(inst); return RFW_HIP_OK; }

int rfw_hip_set_3d_mesh(void* inst, uint32_t id, const rfw_mesh_data_3d* d)
{
    LOCK(inst);
    if (!d || (d->num_triangles && !d->triangles)) return fail(I, RFW_HIP_E_INVALID, "set_3d_mesh: null data");
    if (d->num_triangles > kLeafFirstMask) return fail(I, RFW_HIP_E_INVALID, "set_3d_mesh: more than 2^27 triangles in one mesh");
    MeshHost& m = I->meshes[id];
    m.tris.assign(d->triangles, d->triangles + d->num_triangles); // copy: the borrow ends with this call
    m.skin.clear();
    if (d->skin_data && d->num_skin_data == 3u * d->num_triangles && (d->flags & RFW_MESH_ALLOW_SKINNING))
        m.skin.assign(d->skin_data, d->skin_data + d->num_skin_data);
    m.dirty = true;
    I->meshes_dirty = true;
    return RFW_HIP_OK;
}

int rfw_hip_unload_3d_meshes(void* inst, const uint32_t* ids, uint32_t n)
{
    LOCK(inst);
    if (n && !ids) return fail(I, RFW_HIP_E_INVALID, "unload_3d_meshes: null ids");
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

static int render_impl(Instance* I, const rfw_camera_view_3d* views, uint32_t k, bool samples = false)
{
    if (I->slots.empty()) return do_render(I, views, k, samples);
    // frames in flight: does this call add a sample to the image of the current slot, or start a new image on the next slot?
    Instance* cur = slot_ptr(I, I->cur_slot);
    const bool same_image = (k == 1 || samples) && !I->restart && cur->sample_count > 0 && cur->have_last_view &&
                            std::memcmp(&cur->last_view, views, sizeof(*views)) == 0 && cur->rendered_version == I->scene_version;
    if (!same_image) {
        I->cur_slot = (I->cur_slot + 1) % (uint32_t)(I->slots.size() + 1);
        cur = slot_ptr(I, I->cur_slot);
        cur->sample_count = 0;
    }
    I->restart = false;
    cur->rendered_version = I->scene_version;
    if (cur != I) { // the owner's options apply to every slot
        cur->max_path_length = I->max_path_length; cur->clamp_value = I->clamp_value; cur->flags = I->flags; cur->timing = I->timing;
        for (int c = 0; c < 3; c++) cur->sky[c] = I->sky[c];
    }
    int rc = ensure_slot_tlas(I, cur);
    if (rc == RFW_HIP_OK) rc = do_render(cur, views, k, samples);
    if (rc != RFW_HIP_OK && cur != I && !cur->err.empty()) I->err = cur->err;
    return rc;
}

int rfw_hip_render(void* inst, const rfw_mat4* /*view_2d*/, const rfw_camera_view_3d* view, uint32_t /*mode*/)
{
    LOCK(inst);
    if (!view) return fail(I, RFW_HIP_E_INVALID, "render: null view");
    CHECK_OVERFLOW(I); // of an earlier frame or query (sticky until synchronize() rebuilds the trees)
    return render_impl(I, view, 1);
}

int rfw_hip_render_batch(void* inst, const rfw_camera_view_3d* views, uint32_t count)
{
    LOCK(inst);
    if (!views || count == 0) return fail(I, RFW_HIP_E_INVALID, "render_batch: no views");
    CHECK_OVERFLOW(I);
    if (count == 1) { // a batch of one is still a NEW image
        I->sample_count = 0;
        I->restart = true;
    }
    const int rc = render_impl(I, views, count);
    if (rc == RFW_HIP_OK && count == 1) (I->slots.empty() ? I : slot_ptr(I, I->cur_slot))->after_batch = true;
    return rc;
}

int rfw_hip_render_samples(void* inst, const rfw_camera_view_3d* view, uint32_t count)
{
    LOCK(inst);
    if (!view || count == 0) return fail(I, RFW_HIP_E_INVALID, "render_samples: no view / no samples");
    CHECK_OVERFLOW(I);
    if (count > I->max_batch || count > (uint32_t)kMaxBatch) return fail(I, RFW_HIP_E_INVALID, "render_samples: more samples than options.max_batch");
    if (count == 1) return render_impl(I, view, 1);
    rfw_camera_view_3d views[kMaxBatch];
    for (uint32_t f = 0; f < count; f++) views[f] = *view;
    return render_impl(I, views, count, true);
}

int rfw_hip_comm_unique_id(void* out128)
{
    if (!out128) return RFW_HIP_E_INVALID;
    std::lock_guard<std::mutex> g(g_rccl_mu);
    if (!g_rccl.load()) { g_create_error = g_rccl.error; return RFW_HIP_E_DEVICE; }
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.get_unique_id(&id);
    if (r != ncclSuccess) { g_create_error = std::string("ncclGetUniqueId: ") + g_rccl.error_string(r); return RFW_HIP_E_DEVICE; }
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    std::memcpy(out128, &id, 128);
    return RFW_HIP_OK;
}

int rfw_hip_comm_init(void* inst, const void* id128, uint32_t rank, uint32_t world)
{
    LOCK(inst);
    if (!id128 || world == 0 || rank >= world) return fail(I, RFW_HIP_E_INVALID, "comm_init: bad arguments");
    if (rank != I->rank || world != I->world) return fail(I, RFW_HIP_E_INVALID, "comm_init: rank / world differ from the shard this instance was created with (rfw_hip_options.rank / world)");
    if (I->substreams > 1) return fail(I, RFW_HIP_E_STATE, "comm_init: an instance with sub-streams cannot own a communicator");
    if (I->comm) return fail(I, RFW_HIP_E_STATE, "comm_init: this instance already has a communicator");
    if (I->p2p.data) return fail(I, RFW_HIP_E_STATE, "comm_init: this instance already exchanges by peer stores (rfw_hip_p2p_*)");
    {
        std::lock_guard<std::mutex> g(g_rccl_mu);
        if (!g_rccl.load()) return fail(I, RFW_HIP_E_DEVICE, g_rccl.error);
    }
    HIP_TRY(I, hipSetDevice(I->device));
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    const ncclResult_t r = g_rccl.comm_init_rank(&I->comm, (int)world, id, (int)rank);
    if (r != ncclSuccess) { I->comm = nullptr; return fail(I, RFW_HIP_E_DEVICE, std::string("ncclCommInitRank: ") + g_rccl.error_string(r)); }
    const size_t n = (size_t)I->capacity * I->max_batch * 3u; // (room for the widest format: the option may still change)
    for (uint32_t k = 0; k <= I->slots.size(); k++) { // every frame slot gathers into buffers of its own, on its own stream, through the owner's communicator
        Instance* c = slot_ptr(I, k);
        HIP_TRY(I, c->d_send.ensure(n));
        HIP_TRY(I, c->d_recv.ensure(n * world));
        HIP_TRY(I, hipMemsetAsync(c->d_recv.ptr, 0, n * world * sizeof(float), c->stream));
        c->sample_count = 0;
    }
    if (!I->slots.empty() && !I->comm_chain) HIP_TRY(I, hipEventCreateWithFlags(&I->comm_chain, hipEventDisableTiming));
    I->comm_chain_pending = false;
    return RFW_HIP_OK;
}

int rfw_hip_comm_destroy(void* inst)
{
    LOCK(inst);
    if (!I->comm) return RFW_HIP_OK;
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    for (Instance* c : I->slots) HIP_TRY(I, hipStreamSynchronize(c->stream));
    (void)g_rccl.comm_destroy(I->comm);
    I->comm = nullptr;
    I->acc_source = nullptr;
    return RFW_HIP_OK;
}

// ---- the exchange by peer stores
namespace {
struct P2PHandle { // RFW_HIP_P2P_HANDLE_BYTES on the wire
    uint32_t magic, rank, world, n_slots;
    int64_t pid;
    int32_t device, pad;
    uint64_t data, flags, slot_words, flags_bytes;
    hipIpcMemHandle_t data_ipc, flags_ipc;
};
static_assert(sizeof(P2PHandle) <= RFW_HIP_P2P_HANDLE_BYTES, "P2P handle grew beyond its wire size");
constexpr uint32_t kP2PMagic = 0x70325032u;
void p2p_release(Instance* I)
{
    Instance::P2P& P = I->p2p;
    for (size_t q = 0; q < P.opened.size(); q++) {
        if (P.opened[q] & 1u) (void)hipIpcCloseMemHandle(P.peer_data[q]);
        if (P.opened[q] & 2u) (void)hipIpcCloseMemHandle(P.peer_flags[q]);
    }
    P.peer_data.clear(); P.peer_flags.clear(); P.opened.clear();
    if (P.data) (void)hipFree(P.data);
    if (P.flags) (void)hipFree(P.flags);
    P.data = nullptr; P.flags = nullptr; P.connected = false; P.slot_words = 0; P.n_slots = 0;
    for (uint32_t k = 0; k <= I->slots.size(); k++) // a timeout seen by the old connection says nothing about the next one
        if (slot_ptr(I, k)->overflow_host) ((volatile uint32_t*)slot_ptr(I, k)->overflow_host)[1] = 0u;
}
} // namespace

int rfw_hip_p2p_export(void* inst, void* handle_out)
{
    LOCK(inst);
    if (!handle_out) return fail(I, RFW_HIP_E_INVALID, "p2p_export: null handle");
    if (I->scene) return fail(I, RFW_HIP_E_INVALID, "p2p_export: call it on the instance, not on a frame slot");
    if (I->substreams > 1) return fail(I, RFW_HIP_E_STATE, "p2p_export: not available with sub-streams");
    if (I->comm) return fail(I, RFW_HIP_E_STATE, "p2p_export: this instance already gathers through a communicator");
    if (I->world > 16) return fail(I, RFW_HIP_E_INVALID, "p2p_export: at most 16 ranks");
    if (I->p2p.connected) return fail(I, RFW_HIP_E_STATE, "p2p_export: already connected");
    HIP_TRY(I, hipSetDevice(I->device));
    Instance::P2P& P = I->p2p;
    p2p_release(I);
    P.n_slots = 1u + (uint32_t)I->slots.size();
    P.slot_words = (size_t)I->world * I->max_batch * I->capacity * 3u;
    const size_t flag_bytes = std::max<size_t>((size_t)P.n_slots * 2u * I->world * sizeof(uint32_t), 4096);
    // The receive buffer is written by the PEERS (stores over xGMI) and read by this device's de-tiling kernel.  Ordinary hipMalloc memory is
    // cached in this device's L2, which a remote store does not invalidate: fine-grained (system-scope coherent) memory instead, so that a
    // frame never de-tiles a stale line whatever the kernel-boundary cache policy is.  RFW_P2P_DATA_CACHED=1 keeps round 3's
    // plain allocation (A/B on a multi-GPU node); a device without fine-grained memory falls back to it as well.
    if (getenv("RFW_P2P_DATA_CACHED") || hipExtMallocWithFlags((void**)&P.data, P.n_slots * P.slot_words * sizeof(uint32_t), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        P.data = nullptr;
        HIP_TRY(I, hipMalloc((void**)&P.data, P.n_slots * P.slot_words * sizeof(uint32_t)));
    }
    // flag words are polled while peers write them: uncached, so that a poll never reads a stale line of this device's L2
    // (RFW_P2P_FLAGS_FINEGRAINED=1 forces the fall-back kind of memory, so that tests can take that path)
    if (getenv("RFW_P2P_FLAGS_FINEGRAINED") || hipExtMallocWithFlags((void**)&P.flags, flag_bytes, hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        P.flags = nullptr;
        if (hipExtMallocWithFlags((void**)&P.flags, flag_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            p2p_release(I);
            return fail(I, RFW_HIP_E_DEVICE, "p2p_export: no uncached / fine-grained device memory for the flag words");
        }
    }
    HIP_TRY(I, hipMemsetAsync(P.data, 0, P.n_slots * P.slot_words * sizeof(uint32_t), I->stream));
    HIP_TRY(I, hipMemsetAsync(P.flags, 0, flag_bytes, I->stream));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    P2PHandle hd;
    std::memset(&hd, 0, sizeof(hd));
    hd.magic = kP2PMagic; hd.rank = I->rank; hd.world = I->world; hd.n_slots = P.n_slots;
    hd.pid = (int64_t)getpid(); hd.device = I->device;
    hd.data = (uint64_t)(uintptr_t)P.data; hd.flags = (uint64_t)(uintptr_t)P.flags; hd.slot_words = P.slot_words; hd.flags_bytes = flag_bytes;
    // (a peer of this very process uses the addresses; the IPC handles are for the other processes — a failure to make them only matters there)
    if (hipIpcGetMemHandle(&hd.data_ipc, P.data) != hipSuccess || hipIpcGetMemHandle(&hd.flags_ipc, P.flags) != hipSuccess) {
        (void)hipGetLastError();
        hd.pad = 1; // no IPC handles in this blob
    }
    std::memset(handle_out, 0, RFW_HIP_P2P_HANDLE_BYTES);
    std::memcpy(handle_out, &hd, sizeof(hd));
    return RFW_HIP_OK;
}

int rfw_hip_p2p_connect(void* inst, const void* handles)
{
    LOCK(inst);
    if (!handles) return fail(I, RFW_HIP_E_INVALID, "p2p_connect: null handles");
    Instance::P2P& P = I->p2p;
    if (!P.data || !P.flags) return fail(I, RFW_HIP_E_STATE, "p2p_connect: rfw_hip_p2p_export first");
    if (P.connected) return fail(I, RFW_HIP_E_STATE, "p2p_connect: already connected");
    HIP_TRY(I, hipSetDevice(I->device));
    const uint32_t W = I->world;
    P.peer_data.assign(W, nullptr); P.peer_flags.assign(W, nullptr); P.opened.assign(W, 0);
    for (uint32_t q = 0; q < W; q++) {
        P2PHandle hd;
        std::memcpy(&hd, (const uint8_t*)handles + (size_t)q * RFW_HIP_P2P_HANDLE_BYTES, sizeof(hd));
        if (hd.magic != kP2PMagic || hd.rank != q || hd.world != W || hd.n_slots != P.n_slots || hd.slot_words != P.slot_words) {
            p2p_release(I);
            return fail(I, RFW_HIP_E_INVALID, "p2p_connect: handle " + std::to_string(q) + " is not rank " + std::to_string(q) + "'s handle of an instance of this size, world and number of frame slots");
        }
        if (q == I->rank) {
            P.peer_data[q] = P.data; P.peer_flags[q] = P.flags;
        } else if (hd.pid == (int64_t)getpid()) { // one process driving several devices (or several ranks of one device: the tests)
            if (hd.device != I->device) {
                int can = 0;
                (void)hipDeviceCanAccessPeer(&can, I->device, hd.device);
                if (!can) { p2p_release(I); return fail(I, RFW_HIP_E_DEVICE, "p2p_connect: device " + std::to_string(I->device) + " cannot access device " + std::to_string(hd.device)); }
                const hipError_t pe = hipDeviceEnablePeerAccess(hd.device, 0);
                if (pe != hipSuccess && pe != hipErrorPeerAccessAlreadyEnabled) { p2p_release(I); return fail(I, RFW_HIP_E_DEVICE, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(pe)); }
                (void)hipGetLastError();
            }
            P.peer_data[q] = (uint32_t*)(uintptr_t)hd.data; P.peer_flags[q] = (uint32_t*)(uintptr_t)hd.flags;
        } else {
            if (hd.pad) { p2p_release(I); return fail(I, RFW_HIP_E_DEVICE, "p2p_connect: rank " + std::to_string(q) + " could not export IPC handles (hipIpcGetMemHandle)"); }
            void* pd = nullptr; void* pf = nullptr;
            hipError_t e1 = hipIpcOpenMemHandle(&pd, hd.data_ipc, hipIpcMemLazyEnablePeerAccess);
            if (e1 == hipSuccess) { P.peer_data[q] = (uint32_t*)pd; P.opened[q] |= 1u; }
            hipError_t e2 = e1 == hipSuccess ? hipIpcOpenMemHandle(&pf, hd.flags_ipc, hipIpcMemLazyEnablePeerAccess) : e1;
            if (e2 == hipSuccess) { P.peer_flags[q] = (uint32_t*)pf; P.opened[q] |= 2u; }
            if (e2 != hipSuccess) { (void)hipGetLastError(); p2p_release(I); return fail(I, RFW_HIP_E_DEVICE, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e2)); }
        }
    }
    for (uint32_t k = 0; k <= I->slots.size(); k++) {
        Instance* c = slot_ptr(I, k);
        c->p2p_seq = 0; c->sample_count = 0; c->frame_elsewhere = false;
        if (c->overflow_host) ((volatile uint32_t*)c->overflow_host)[1] = 0u;
    }
    P.connected = true;
    return RFW_HIP_OK;
}

int rfw_hip_p2p_disconnect(void* inst)
{
    LOCK(inst);
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    for (Instance* c : I->slots) HIP_TRY(I, hipStreamSynchronize(c->stream));
    p2p_release(I);
    for (uint32_t k = 0; k <= I->slots.size(); k++) { Instance* c = slot_ptr(I, k); c->frame_elsewhere = false; c->acc_source = nullptr; c->sample_count = 0; }
    return RFW_HIP_OK;
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

int rfw_hip_resize(void* inst, uint32_t w, uint32_t h, double)
{
    LOCK(inst);
    if (w == 0 || h == 0) return fail(I, RFW_HIP_E_INVALID, "resize: zero size");
    if (scene_of(I)->p2p.data && (w != I->width || h != I->height)) return fail(I, RFW_HIP_E_STATE, "resize: disconnect the p2p exchange first (its buffers are sized for the frame)");
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    for (Instance* c : I->slots) {
        const int rc = rfw_hip_resize(c, w, h, 1.0);
        if (rc != RFW_HIP_OK) return fail(I, rc, c->err);
    }
    I->restart = true;
    I->width = w;
    I->height = h;
    // a gathered frame not de-tiled yet belongs to the old size (and d_recv may move below): forget it (each slot passes here for itself)
    I->deferred = Instance::Deferred(); I->acc_source = nullptr; I->presented_valid = false;
    const int arc = alloc_paths(I); // also restarts accumulation (gpu-rt/src/lib.rs:1809)
    if (arc == RFW_HIP_OK && scene_of(I)->comm) { // the gather buffers follow the slab size
        const size_t n = (size_t)I->capacity * I->max_batch * 3u;
        HIP_TRY(I, I->d_send.ensure(n));
        HIP_TRY(I, I->d_recv.ensure(n * I->world));
    }
    return arc;
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

int rfw_hip_reset_accumulation(void* inst)
{
    LOCK(inst);
    I->sample_count = 0;
    I->restart = true; // with frame slots: the next render starts a new image (on the next slot)
    return RFW_HIP_OK;
}

int rfw_hip_set_option(void* inst, const char* key, double value)
{
    LOCK(inst);
    if (!key) return fail(I, RFW_HIP_E_INVALID, "set_option: null key");
    const std::string k(key);
    if (k == "max_path_length") I->max_path_length = std::max<uint32_t>(1, std::min<uint32_t>((uint32_t)value, kMaxBounces));
    else if (k == "clamp_value") I->clamp_value = (float)value;
    else if (k == "nee") I->flags = value != 0.0 ? (I->flags & ~RFW_HIP_FLAG_NO_NEE) : (I->flags | RFW_HIP_FLAG_NO_NEE);
    else if (k == "count_traversal") I->flags = value != 0.0 ? (I->flags | RFW_HIP_FLAG_COUNT_TRAVERSAL) : (I->flags & ~RFW_HIP_FLAG_COUNT_TRAVERSAL);
    else if (k == "shadow_order") { // which end any-hit traversals start from: 0 = default (directional lights far to near, positional near to far), 1 = all near to far, 2 = all far to near
        I->flags &= ~(kFlagNearFirstDirectional | kFlagFarFirstPositional);
        if ((int)value == 1) I->flags |= kFlagNearFirstDirectional;
        else if ((int)value == 2) I->flags |= kFlagFarFirstPositional;
    }
    else if (k == "packet_trace") { // which rays walk the tree as wavefront packets (traverse_packet.h): bit 0 camera rays, bit 1 the camera paths' shadow rays
        I->flags &= ~(kFlagPacketPrimary | kFlagPacketShadow);
        if ((int)value & 1) I->flags |= kFlagPacketPrimary;
        if ((int)value & 2) I->flags |= kFlagPacketShadow;
    }
    else if (k == "sample_count") I->sample_count = (uint32_t)value;
    else if (k == "gather_format") { // 0 f32 accumulator RGB, 1 f16 finished frame, 2 presented BGRA8 (sharded frames only)
        if (value < 0 || value > 2) return fail(I, RFW_HIP_E_INVALID, "set_option: gather_format is 0, 1 or 2");
        I->gather_format = (uint32_t)value;
        for (uint32_t q = 0; q <= I->slots.size(); q++) { // every slot: a frame gathered in the old format must not be de-tiled in the new one
            Instance* c = slot_ptr(I, q);
            c->deferred = Instance::Deferred(); c->acc_source = nullptr; c->presented_valid = false;
        }
    }
    else if (k == "present_rank") I->present_rank = (int)value;
    else if (k == "timing") I->timing = value != 0.0;
    else if (k == "sort_extension_rays") I->sort_extension_rays = std::max(0, std::min(2, (int)value));
    else if (k == "texture_array") { I->texture_array = value != 0.0; I->tex_offsets.clear(); } // applies to textures set from now on (all of them: no partial update across the switch)
    else if (k == "spill_rows") { // tests: exercise the overflow path
        I->spill_rows = std::min<uint32_t>((uint32_t)std::max(0.0, value), (uint32_t)kStackSpill);
        clear_overflow(I); // an overflow seen with another stack size says nothing about this one
    }
    else if (k == "sky_r") I->sky[0] = (float)value;
    else if (k == "sky_g") I->sky[1] = (float)value;
    else if (k == "sky_b") I->sky[2] = (float)value;
    else if (k == "sah_max_leaf") I->sah_max_leaf = std::max(1, std::min((int)value, kMaxLeafTris));
    else if (k == "sah_trav_cost") I->sah_trav_cost = (float)value;
    else if (k == "stream_run") {
        const uint32_t r = (uint32_t)value;
        if (r != 0 && (r > 64 || (r & (r - 1)) != 0)) return fail(I, RFW_HIP_E_INVALID, "set_option: stream_run must be 0 or a power of two up to 64");
        I->stream_run = r;
        I->stream_auto = false;
    } else if (k == "stream_auto") { I->stream_auto = value != 0.0;
    } else if (k == "stream_leaf_gate") I->stream_leaf_gate = (uint32_t)std::max(1.0, std::min(64.0, value));
    else if (k == "stream_refill") I->stream_refill = (uint32_t)std::max(1.0, std::min(64.0, value));
    else if (k == "p2p_timeout_ms") I->p2p.timeout_ticks = (uint64_t)std::max(1.0, value) * 100000ull;
    else if (k == "build_threads") I->build_threads = std::max(1, (int)value);
    else return fail(I, RFW_HIP_E_INVALID, "set_option: unknown key " + k);
    return RFW_HIP_OK;
}

// de-tiles the linear accumulator of the latest frame(s) into d_frame_acc, on the instance's stream (zeros before the first frame)
static int materialize_accumulator(Instance* I)
{
    if (scene_of(I)->gather_format != 0 && gathers_tiles(I))
        return fail(I, RFW_HIP_E_STATE, "read_accumulator: with gather_format 1 / 2 only the finished frame travels; the accumulators stay on the ranks that own the tiles");
    { const int rc = ensure_assembled(I); if (rc != RFW_HIP_OK) return rc; }
    const size_t px = (size_t)I->width * I->height * I->max_batch;
    HIP_TRY(I, I->d_frame_acc.ensure(px));
    if (!I->acc_source) {
        HIP_TRY(I, hipMemsetAsync(I->d_frame_acc.ptr, 0, px * sizeof(float4), I->stream));
        return RFW_HIP_OK;
    }
    CameraParams cam = camera_params(I, I->last_view);
    cam.batch = I->acc_source_batch;
    launch_assemble(I->stream, cam, I->acc_source, I->acc_source_rgb, true, I->cap_v, I->d_frame_acc.ptr, 1u);
    HIP_TRY(I, hipGetLastError());
    return RFW_HIP_OK;
}
static int read_frame_impl(void* inst, uint32_t frame, bool accumulator, float* rgba, uint64_t n);
int rfw_hip_read_framebuffer_at(void* inst, uint32_t frame, float* rgba, uint64_t n) { return read_frame_impl(inst, frame, false, rgba, n); }
int rfw_hip_read_accumulator_at(void* inst, uint32_t frame, float* rgba, uint64_t n) { return read_frame_impl(inst, frame, true, rgba, n); }
static int read_frame_impl(void* inst, uint32_t frame, bool accumulator, float* rgba, uint64_t n)
{
    LOCK(inst);
    if (!rgba || n != (uint64_t)I->width * I->height * 4) return fail(I, RFW_HIP_E_INVALID, "read_*_at: size mismatch");
    if (frame >= I->max_batch) return fail(I, RFW_HIP_E_INVALID, "read_*_at: frame index beyond options.max_batch");
    if (!I->slots.empty() && I->cur_slot != 0) { // frames in flight: the latest batch lives in a slot
        Instance* c = slot_ptr(I, I->cur_slot);
        const int rc = read_frame_impl(c, frame, accumulator, rgba, n);
        if (rc != RFW_HIP_OK) I->err = c->err;
        return rc;
    }
    HIP_TRY(I, hipSetDevice(I->device));
    if (accumulator) {
        const int rc = materialize_accumulator(I);
        if (rc != RFW_HIP_OK) return rc;
    } else {
        if (scene_of(I)->gather_format == 2 && gathers_tiles(I))
            return fail(I, RFW_HIP_E_STATE, "read_framebuffer: gather_format 2 leaves the PRESENTED frame only: rfw_hip_download_frame(what = 2)");
        const int rc = ensure_assembled(I);
        if (rc != RFW_HIP_OK) return rc;
    }
    const float4* src = (accumulator ? I->d_frame_acc.ptr : I->d_frame_out.ptr) + (size_t)frame * I->width * I->height;
    HIP_TRY(I, hipMemcpyAsync(rgba, src, n * sizeof(float), hipMemcpyDeviceToHost, I->stream));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    CHECK_OVERFLOW(I);
    return RFW_HIP_OK;
}

// step k = the smallest linear value whose sRGB encoding rounds to byte k + 1: srgb_to_linear((k + 0.5) / 255), IEC 61966-2-1
void rfw_hip_srgb_steps(float* out255)
{
    if (out255) std::memcpy(out255, srgb_steps(), 255 * sizeof(float));
}
// Pinned host memory for rfw_hip_download_frame (a pageable destination would make the copy synchronous and staged)
void* rfw_hip_host_alloc(uint64_t bytes)
{
    void* p = nullptr;
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void rfw_hip_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}
// Queue the copy of the latest frame (what = 0: finalised frame, 1: accumulator; batch frame `frame`) to host memory behind the
// kernels that produce it, on that frame's stream, and return: with frames in flight the copy of frame k runs on the DMA engines
// while the slots of frames k+1... trace.  The bytes are valid after rfw_hip_wait_downloads.
int rfw_hip_download_frame(void* inst, uint32_t what, uint32_t frame, float* host_rgba, uint64_t n)
{
    LOCK(inst);
    const uint64_t px = (uint64_t)I->width * I->height;
    if (!host_rgba || n != (what == 2 ? px : px * 4)) return fail(I, RFW_HIP_E_INVALID, "download_frame: size mismatch");
    if (what > 2 || frame >= I->max_batch) return fail(I, RFW_HIP_E_INVALID, "download_frame: bad selector");
    Instance* c = I->slots.empty() ? I : slot_ptr(I, I->cur_slot);
    HIP_TRY(I, hipSetDevice(I->device));
    if (!c->download_done) HIP_TRY(I, hipEventCreateWithFlags(&c->download_done, hipEventDisableTiming));
    if (what == 1) {
        const int rc = materialize_accumulator(c);
        if (rc != RFW_HIP_OK) return fail(I, rc, c->err);
    } else {
        const int rc = ensure_assembled(c);
        if (rc != RFW_HIP_OK) return fail(I, rc, c->err);
    }
    const bool sharded_presented = scene_of(c)->gather_format == 2 && gathers_tiles(c);
    if (sharded_presented) { // the gathered frame IS the presented frame: de-tiled into d_present already, nothing to encode
        if (what != 2) return fail(I, RFW_HIP_E_STATE, "download_frame: gather_format 2 leaves the PRESENTED frame only (what = 2)");
        if (!c->presented_valid) return fail(I, RFW_HIP_E_STATE, "download_frame: no gathered frame yet");
        HIP_TRY(I, hipMemcpyAsync(host_rgba, c->d_present.ptr + (size_t)frame * px, px * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(I, hipEventRecord(c->download_done, c->stream));
        c->download_dst.push_back(host_rgba);
        return RFW_HIP_OK;
    }
    const float4* src = (what == 1 ? c->d_frame_acc.ptr : c->d_frame_out.ptr) + (size_t)frame * px;
    // Presented frame into a pinned destination (rfw_hip_host_alloc, or registered by the caller): the encoding kernel stores straight
    // into host memory over the link, no copy command (measured: 0.731 ms per frame against 0.762 with encode + copy, 8 frames in flight).
    // Float frames, and pageable destinations, go through the runtime's copy.
    void* mapped = nullptr;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host_rgba) == hipSuccess && attr.type == hipMemoryTypeHost) mapped = attr.devicePointer;
    else (void)hipGetLastError();
    if (what == 2) { // the swap-chain image: encode on the device, a quarter of the bytes travel
        uint32_t* out = (uint32_t*)mapped;
        if (!out) {
            HIP_TRY(I, c->d_present.ensure(px));
            out = c->d_present.ptr;
        }
        launch_present(c->stream, src, out, px, srgb_steps(), mapped != nullptr);
        HIP_TRY(I, hipGetLastError());
        if (!mapped) HIP_TRY(I, hipMemcpyAsync(host_rgba, c->d_present.ptr, px * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    } else { // 33 MB per 1080p frame: the DMA engine moves it at ~40 GB/s; stores from a kernel reach ~28 GB/s (measured)
        HIP_TRY(I, hipMemcpyAsync(host_rgba, src, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    }
    HIP_TRY(I, hipEventRecord(c->download_done, c->stream));
    c->download_dst.push_back(host_rgba);
    return RFW_HIP_OK;
}
// host_ptr == NULL: every copy queued so far; else the copy into host_ptr (and, being on the same stream, those queued before it on its slot)
int rfw_hip_wait_download(void* inst, const void* host_ptr)
{
    LOCK(inst);
    HIP_TRY(I, hipSetDevice(I->device));
    for (size_t k = 0; k <= I->slots.size(); k++) {
        Instance* c = slot_ptr(I, (uint32_t)k);
        if (c->download_dst.empty()) continue;
        if (host_ptr && std::find(c->download_dst.begin(), c->download_dst.end(), host_ptr) == c->download_dst.end()) continue;
        HIP_TRY(I, hipEventSynchronize(c->download_done)); // the event is behind the LAST copy of this slot
        c->download_dst.clear();
    }
    return RFW_HIP_OK;
}
int rfw_hip_wait_downloads(void* inst) { return rfw_hip_wait_download(inst, nullptr); }

// the latest frame (frame 0 of a batch)
int rfw_hip_read_framebuffer(void* inst, float* rgba, uint64_t n) { return read_frame_impl(inst, 0, false, rgba, n); }
int rfw_hip_read_accumulator(void* inst, float* rgba, uint64_t n) { return read_frame_impl(inst, 0, true, rgba, n); }

static void add_frame_timing(Instance* I, int slot, uint32_t nb, bool nee, rfw_hip_frame_stats* out)
{
    hipEvent_t* e0 = ring_events(I, slot, 0);
    auto el = [&](hipEvent_t a, hipEvent_t b) { float ms = 0.0f; (void)hipEventElapsedTime(&ms, a, b); return ms; };
    out->ms_total += el(e0[EV_FRAME0], e0[EV_FRAME1]);
    out->ms_other += el(e0[kEvBlit], e0[kEvBlit + 1]);
    // per-kernel figures are SUMS over the sub-shard launches (which overlap in time on different streams)
    for (uint32_t s = 0; s < I->substreams; s++) {
        hipEvent_t* ev = ring_events(I, slot, s);
        for (uint32_t b = 0; b < nb; b++) {
            const float tr = el(ev[ev_index(b, 0, 0)], ev[ev_index(b, 0, 1)]);
            if (b == 0) out->ms_trace_primary += tr; else out->ms_trace_extend += tr;
            out->ms_shade += el(ev[ev_index(b, 1, 0)], ev[ev_index(b, 1, 1)]);
            if (nee) out->ms_trace_shadow += el(ev[ev_index(b, 2, 0)], ev[ev_index(b, 2, 1)]);
        }
    }
}

int rfw_hip_get_frame_stats(void* inst, rfw_hip_frame_stats* out)
{
    LOCK(inst);
    if (!out) return fail(I, RFW_HIP_E_INVALID, "get_frame_stats: null out");
    if (!I->slots.empty() && I->cur_slot != 0) { // frames in flight: the latest frame lives in a slot
        Instance* c = slot_ptr(I, I->cur_slot);
        const int rc = rfw_hip_get_frame_stats(c, out);
        if (rc != RFW_HIP_OK) I->err = c->err;
        return rc;
    }
    HIP_TRY(I, hipSetDevice(I->device));
    std::memset(out, 0, sizeof(*out));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    QueueCounters qc[kMaxSub];
    HIP_TRY(I, hipMemcpy(qc, I->d_counters.ptr, I->substreams * sizeof(QueueCounters), hipMemcpyDeviceToHost));
    CHECK_OVERFLOW(I);
    const uint32_t nb = I->last_bounces;
    out->primary_rays = nb ? I->local_pixels : 0;
    for (uint32_t s = 0; s < I->substreams; s++) {
        for (uint32_t b = 0; b + 1 < nb; b++) out->extension_rays += qc[s].ext[b];
        if (!(I->flags & RFW_HIP_FLAG_NO_NEE))
            for (uint32_t b = 0; b < nb; b++)
                for (int k = 0; k < kShadowBuckets; k++) out->shadow_rays += qc[s].shadow[b][k];
        for (int k = 0; k < 3; k++) {
            out->nodes_visited[k] += qc[s].trav[k][0];
            out->tris_tested[k] += qc[s].trav[k][1];
            out->instances_entered[k] += qc[s].trav[k][2];
            out->node_test_executions[k] += qc[s].wave_exec[k][0];
            out->tri_test_executions[k] += qc[s].wave_exec[k][1];
            out->uniform_node_test_executions[k] += qc[s].wave_uniform[k];
            out->wave_max_nodes[k] += qc[s].wave_max_nodes[k];
        }
    }
    out->sample_count = I->sample_count;
    out->bounces = nb;
    out->substreams = I->substreams;
    if (I->frame_recorded && nb) add_frame_timing(I, (int)((I->frame_index - 1) % kTimingRing), nb, !(I->flags & RFW_HIP_FLAG_NO_NEE), out);
    return RFW_HIP_OK;
}

int rfw_hip_drain_timing(void* inst, rfw_hip_frame_stats* sum, uint32_t* frames)
{
    LOCK(inst);
    if (!sum || !frames) return fail(I, RFW_HIP_E_INVALID, "drain_timing: null out");
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    std::memset(sum, 0, sizeof(*sum));
    uint32_t n = 0;
    for (uint64_t f = I->drained_index; f < I->frame_index; f++) {
        const int slot = (int)(f % kTimingRing);
        const uint32_t nb = I->ring_bounces[slot];
        if (!nb) continue;
        add_frame_timing(I, slot, nb, I->ring_nee[slot], sum);
        n++;
    }
    sum->substreams = I->substreams;
    I->drained_index = I->frame_index;
    for (Instance* c : I->slots) { // frames in flight: the timings of every slot's frames
        rfw_hip_frame_stats cs;
        uint32_t cn = 0;
        const int rc = rfw_hip_drain_timing(c, &cs, &cn);
        if (rc != RFW_HIP_OK) return fail(I, rc, c->err);
        sum->ms_total += cs.ms_total; sum->ms_trace_primary += cs.ms_trace_primary; sum->ms_trace_extend += cs.ms_trace_extend;
        sum->ms_trace_shadow += cs.ms_trace_shadow; sum->ms_shade += cs.ms_shade; sum->ms_other += cs.ms_other;
        n += cn;
    }
    *frames = n;
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
    out->triangles = I->n_tris;
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
            (void)hipEventElapsedTime(&I->ms_blas_upload, I->ev_build[0], I->ev_build[1]);
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

int rfw_hip_set_stream(void* inst, void* stream)
{
    LOCK(inst);
    if (!I->slots.empty()) return fail(I, RFW_HIP_E_STATE, "set_stream: an instance with frames in flight launches on its slots' own streams");
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    I->stream = stream ? (hipStream_t)stream : I->own_stream;
    return RFW_HIP_OK;
}

void* rfw_hip_get_stream(void* inst)
{
    if (!inst) return nullptr;
    Instance* I = static_cast<Instance*>(inst);
    std::lock_guard<std::mutex> g(I->mu);
    return (void*)I->stream;
}

int rfw_hip_device_synchronize(void* inst)
{
    LOCK(inst);
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    for (Instance* c : I->slots) HIP_TRY(I, hipStreamSynchronize(c->stream)); // every frame in flight
    return RFW_HIP_OK;
}

int rfw_hip_shard_info(void* inst, uint64_t* slab_floats, uint32_t* local, uint32_t* total)
{
    LOCK(inst);
    if (slab_floats) *slab_floats = slab_words(I); // 4-byte words per frame in the instance's gather format (format 0: RGB of the accumulator as floats)
    if (local) *local = I->local_tiles;
    if (total) *total = I->tiles_x * I->tiles_y;
    return RFW_HIP_OK;
}
int rfw_hip_set_slab_output(void* inst, void* ptr)
{
    LOCK(inst);
    if (!I->slots.empty()) return fail(I, RFW_HIP_E_STATE, "set_slab_output: not available with frames_in_flight > 1 (one instance per frame in flight instead)");
    I->external_slab = ptr;
    I->sample_count = 0;
    return RFW_HIP_OK;
}
static int assemble_impl(void* inst, const void* gathered, uint32_t k)
{
    LOCK(inst);
    if (!gathered) return fail(I, RFW_HIP_E_INVALID, "assemble_frame: null buffer");
    if (k == 0 || k > I->max_batch || (k > 1 && I->substreams > 1)) return fail(I, RFW_HIP_E_INVALID, "assemble_batch: bad frame count");
    HIP_TRY(I, hipSetDevice(I->device));
    // gathered = [world][substreams][cap_v] = [virtual rank][cap_v]; for a batch (one sub-stream): [rank][frame][cap_v]
    return gathered_arrived(I, I->stream, gathered, k);
}
int rfw_hip_assemble_frame(void* inst, const void* gathered) { return assemble_impl(inst, gathered, 1); }
int rfw_hip_assemble_batch(void* inst, const void* gathered, uint32_t count) { return assemble_impl(inst, gathered, count); }

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
    else if (w == "counters") { src = I->d_counters.ptr; avail = sizeof(QueueCounters); }
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

} // extern "C"
