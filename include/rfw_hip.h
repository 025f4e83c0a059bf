/*
 * rfw_hip.h — C ABI of the MI355X-native wavefront path-tracing backend.
 *
 * One entry point per method of the reference plugin trait
 * `rfw_backend::Backend` (crates/rfw-backend/src/lib.rs:35-82) plus
 * `FromWindowHandle::init` (:26-33), in the shape the reference already uses
 * for its one native backend (backends/metal/cpp/src/library.h:119-135: opaque
 * `void* instance`, POD payloads, pointer + count for slices).  The Rust shim
 * that binds these symbols is shown in INTEGRATION.md.
 *
 * Contract:
 *  - every pointer argument is BORROWED for the duration of the call only; the
 *    library copies what it needs before returning and never retains it;
 *  - no call throws; every call except create/destroy/last_error returns 0 on
 *    success or a negative RFW_HIP_E_* code, with a human-readable message kept
 *    per instance (rfw_hip_last_error);
 *  - calls on one instance are serialised internally (one mutex) and may come
 *    from any thread (bevy runs synchronize_system / render_system on arbitrary
 *    workers: rfw/src/system/mod.rs:16-17, rfw/src/lib.rs:411-430);
 *  - `changed` bit slices (bitvec BitSlice<Lsb0, usize> in the trait) arrive as
 *    packed little-endian u32 words, bit i = element i; NULL means "all".
 *  - there is no CPU fallback: without a HIP device rfw_hip_create fails.
 */
#ifndef RFW_HIP_H
#define RFW_HIP_H

#include "rfw_pod.h"

#ifdef __cplusplus
extern "C" {
#endif

#define RFW_HIP_API __attribute__((visibility("default")))
/* 2 (round 5/6): rfw_hip_scene_stats grew by 24 bytes (split_references, accel_bytes, packet_copies) and its `triangles` became the CALLER'S
 * triangle count; a binding compiled against version 1 must not call rfw_hip_get_scene_stats.  Bindings check rfw_hip_abi_version() at load. */
#define RFW_HIP_ABI_VERSION 2

enum {
    RFW_HIP_OK = 0,
    RFW_HIP_E_INVALID = -1,   /* bad argument */
    RFW_HIP_E_DEVICE = -2,    /* HIP runtime error */
    RFW_HIP_E_STATE = -3,     /* call not valid in the current state */
    RFW_HIP_E_NOMEM = -4
};

/* BLAS/TLAS builder selection (rfw_hip_options.builder) */
enum { RFW_HIP_BUILDER_AUTO = 0, RFW_HIP_BUILDER_HOST_SAH = 1, RFW_HIP_BUILDER_DEVICE_LBVH = 2, RFW_HIP_BUILDER_DEVICE_SAH = 3 };

/*
 * Creation options.  Zero-initialise, then set what you need; a NULL pointer
 * means all defaults.  Defaults reproduce gpu-rt: max path length 3
 * (backends/gpu-rt/src/lib.rs:1708), clamp 10 (:205), NEE on.
 */
typedef struct {
    uint32_t struct_size;      /* = sizeof(rfw_hip_options) */
    int32_t device;            /* HIP device ordinal; -1 = current */
    uint32_t max_path_length;  /* 0 = default 3; 1 = primary (+shadow) only */
    float clamp_value;         /* 0 = default 10.0 */
    uint32_t rank;             /* tile shard of this instance (multi-GPU): rank in [0, world) */
    uint32_t world;            /* 0/1 = render the whole frame */
    uint32_t tile_size;        /* shard tile edge in pixels; 0 = default 64 */
    uint32_t builder;          /* RFW_HIP_BUILDER_*: AUTO = meshes by binned SAH on the device, TLAS by LBVH on the device; HOST_SAH = both
                                  levels by binned SAH on the host cores; DEVICE_LBVH = both levels by LBVH; DEVICE_SAH = as AUTO */
    uint32_t flags;            /* RFW_HIP_FLAG_* */
    uint32_t streams;          /* sub-shards (HIP streams) one frame is split into on this GPU; 0 = default 1, max 8 */
    uint32_t frames_in_flight; /* 0/1 = one frame at a time.  N > 1: N frame slots inside this instance (own path state, queues,
                                  accumulator and HIP stream each, ONE scene): a render() that starts a new image (new view, changed
                                  scene, reset) is queued on the next slot while earlier frames still trace; a render() that adds a
                                  sample to the current image stays on its slot.  Reads return the latest frame.  world must be 1.
                                  Export GPU_MAX_HW_QUEUES >= N before the HIP runtime starts. */
    uint32_t max_batch;        /* 0/1 = none.  B > 1 (<= 16): rfw_hip_render_batch may trace up to B independent frames in ONE launch per
                                  stage (buffers are sized for B frames).  Sub-streams must be 1. */
} rfw_hip_options;

enum {
    RFW_HIP_FLAG_NO_NEE = 1u << 0,        /* skip light sampling / shadow rays */
    RFW_HIP_FLAG_COUNT_TRAVERSAL = 1u << 1, /* accumulate node/triangle visit counters (slow; for roofline bytes) */
    /* bits 2 and 3 are set through rfw_hip_set_option("shadow_order", 0 | 1 | 2): which end of a shadow ray the any-hit traversal starts from —
     * 0 (default) directional lights far to near, positional lights near to far; 1 every ray near to far; 2 every ray far to near.  Speed only. */
    RFW_HIP_FLAG_SHADOW_NEAR_FIRST_DIRECTIONAL = 1u << 2,
    RFW_HIP_FLAG_SHADOW_FAR_FIRST_POSITIONAL = 1u << 3
};

/* Per-frame counters and timings of the last rfw_hip_render (extension; no trait equivalent). */
typedef struct {
    uint64_t primary_rays;
    uint64_t extension_rays;
    uint64_t shadow_rays;
    /* traversal work per kernel kind [0 primary, 1 extension, 2 shadow]; only with RFW_HIP_FLAG_COUNT_TRAVERSAL */
    uint64_t nodes_visited[3];
    uint64_t tris_tested[3];
    uint64_t instances_entered[3];
    float ms_total;             /* hipEvent span of the whole frame on the instance's stream (wall time, sub-shards overlap) */
    float ms_trace_primary;
    float ms_trace_extend;
    float ms_trace_shadow;
    float ms_shade;
    float ms_other;
    uint32_t sample_count;      /* samples accumulated so far */
    uint32_t bounces;
    uint32_t substreams;        /* launches per kernel per frame: the ms_* kernel figures are sums over them */
    uint32_t pad;
    /* SIMD efficiency of the traversal per kernel kind, only with RFW_HIP_FLAG_COUNT_TRAVERSAL: how often a wavefront executed the
     * node test / the triangle test (lane utilisation of the node test = nodes_visited / (64 * node_test_executions)), and the sum
     * over wavefronts of the largest per-lane node count (nodes_visited / (64 * wave_max_nodes) = what idle finished lanes cost). */
    uint64_t node_test_executions[3];
    uint64_t tri_test_executions[3];
    uint64_t wave_max_nodes[3];
    /* node-test executions in which every active lane of the wavefront visited the SAME node (coherent rays near the top of a tree) */
    uint64_t uniform_node_test_executions[3];
} rfw_hip_frame_stats;

/* Sizes of the device-resident acceleration structures (for DESIGN.md byte accounting). */
typedef struct {
    uint64_t triangles;
    uint64_t instances;
    uint64_t blas_nodes;
    uint64_t tlas_nodes;
    uint32_t node_bytes;
    uint32_t tri_bytes;
    float ms_blas_build;  /* last synchronize */
    float ms_tlas_build;
    /* the last device build of (all, or the changed) meshes, by HIP events on the instance's stream: the host -> device copy of their
     * triangles, and the kernels behind it (boxes, builder, leaf-ordered packets, quantised nodes); 0 when the host built */
    float ms_blas_upload;
    float ms_blas_kernels;
    uint64_t blas_upload_bytes;   /* triangles x 176 B */
    uint64_t blas_kernel_bytes;   /* ALGORITHMIC bytes of those kernels (what each pass reads and writes per primitive) */
    /* spatial splits (option "spatial_splits"): references the trees hold beyond one per triangle — duplicates of the few triangles whose
     * boxes waste the most (a wall of two triangles across the scene); `triangles` above counts the caller's triangles only */
    uint64_t split_references;
    /* device memory of the acceleration structures as allocated (capacities): quantised nodes, their eight per-octant copies, the packet form
     * of those copies (present only where a packet kernel can run: packet_copies = 1), 48-B triangle packets, TLAS included; the 176-B
     * records the boundary hands over are not counted */
    uint64_t accel_bytes;
    uint32_t packet_copies;
    uint32_t pad;
} rfw_hip_scene_stats;

/* Hit record of the ray-query extension: what ray_gen/ray_extend store per path
 * (backends/gpu-rt/shaders/ray_gen.comp:66-69) before bary quantisation. */
typedef struct {
    int32_t inst;   /* global instance id, -1 = miss */
    int32_t tri;    /* global triangle id (mesh triangle offset + id), -1 = miss */
    float t;
    float u, v;
} rfw_hip_hit;

/* ---- FromWindowHandle::init / Drop  (crates/rfw-backend/src/lib.rs:26-33; metal: library.h:119-120) ---- */
RFW_HIP_API void* rfw_hip_create(uint32_t width, uint32_t height, double scale, const rfw_hip_options* options);
RFW_HIP_API void rfw_hip_destroy(void* instance);
/* instance may be NULL: returns the message of the last failed rfw_hip_create on this thread. */
RFW_HIP_API const char* rfw_hip_last_error(void* instance);
RFW_HIP_API uint32_t rfw_hip_abi_version(void);
/* Host-only self test of the CPU-side builder + node quantiser (no GPU needed): boxes6 = n x (lo.xyz, hi.xyz); returns the
 * number of structural violations (0 = pass). */
/* Host-only self test of the spatial splits set_3d_mesh computes (no GPU needed): returns the number of references (n + duplicates), -1 on
 * bad arguments.  pieces7 = n_pieces x (mesh-local index, lo.xyz, hi.xyz): the box of every reference of a split triangle (index < n: the
 * triangle's own entry; >= n: a duplicate); duplicate_of[j] = the triangle duplicate n + j stands for.  split_tau as option "spatial_splits". */
RFW_HIP_API int64_t rfw_hip_selftest_splits(const rfw_rt_triangle* tris, uint32_t n, float split_tau, uint32_t threads, float* pieces7, uint32_t pieces_cap,
                                            uint32_t* duplicate_of, uint32_t duplicates_cap, uint32_t* n_pieces);
/* Host-only: the reciprocal the kernels use for `n / d` in their index arithmetic (pixel index / frame width, tile / tiles per row):
 * m with n / d == (n * m) >> 32 for EVERY n <= n_max, or 0 when no such 32-bit constant is exact that far (the kernels then divide). */
RFW_HIP_API uint32_t rfw_hip_selftest_index_magic(uint32_t d, uint64_t n_max);
RFW_HIP_API int64_t rfw_hip_selftest_bvh(const float* boxes6, uint32_t n, uint32_t max_leaf, uint32_t threads, uint32_t* out_nodes);

/* ---- Backend trait, in declaration order (crates/rfw-backend/src/lib.rs:36-81) ---- */
/* :36 set_2d_mesh / :39 set_2d_instances — accepted and ignored; gpu-rt `unimplemented!()`s them
 * (backends/gpu-rt/src/lib.rs:1131-1137). */
RFW_HIP_API int rfw_hip_set_2d_mesh(void* instance, uint32_t id, const void* vertices, uint32_t num_vertices, int32_t tex_id);
RFW_HIP_API int rfw_hip_set_2d_instances(void* instance, uint32_t mesh, const rfw_mat4* matrices, uint32_t num_matrices);
/* :41 set_3d_mesh */
RFW_HIP_API int rfw_hip_set_3d_mesh(void* instance, uint32_t id, const rfw_mesh_data_3d* data);
/* :43 unload_3d_meshes */
RFW_HIP_API int rfw_hip_unload_3d_meshes(void* instance, const uint32_t* ids, uint32_t num_ids);
/* :46 set_3d_instances — a zero matrix marks a removed slot (crates/rfw-scene/src/instances_3d.rs:79-86). */
RFW_HIP_API int rfw_hip_set_3d_instances(void* instance, uint32_t mesh, const rfw_instances_data_3d* data);
/* :49 set_materials — `changed` (here and in every call below that takes one): bit k of word k / 32 = element k, NULL = everything; the
 * array must cover all `num` elements ((num + 31) / 32 words) — a clear bit means the element is not looked at. */
RFW_HIP_API int rfw_hip_set_materials(void* instance, const rfw_device_material* materials, uint32_t num, const uint32_t* changed);
/* :53 set_textures — sampled by shade for diffuse and normal maps (the two maps shade.comp reads).  `changed` (bit k = texture k, may be NULL =
 * all): with the same number of textures as before, textures whose bit is clear are not read at all, and synchronize() uploads only the
 * changed ones. */
RFW_HIP_API int rfw_hip_set_textures(void* instance, const rfw_texture_data* textures, uint32_t num, const uint32_t* changed);
/* :57 synchronize — builds/refits acceleration structures for what changed. */
RFW_HIP_API int rfw_hip_synchronize(void* instance);
/* :60 render — one sample per pixel per call, accumulating (gpu-rt/src/lib.rs:1685-1731). */
RFW_HIP_API int rfw_hip_render(void* instance, const rfw_mat4* view_2d, const rfw_camera_view_3d* view_3d, uint32_t mode);
/* :63 resize */
RFW_HIP_API int rfw_hip_resize(void* instance, uint32_t width, uint32_t height, double scale);
/* :66-75 lights */
RFW_HIP_API int rfw_hip_set_point_lights(void* instance, const rfw_point_light* lights, uint32_t num, const uint32_t* changed);
RFW_HIP_API int rfw_hip_set_spot_lights(void* instance, const rfw_spot_light* lights, uint32_t num, const uint32_t* changed);
RFW_HIP_API int rfw_hip_set_area_lights(void* instance, const rfw_area_light* lights, uint32_t num, const uint32_t* changed);
RFW_HIP_API int rfw_hip_set_directional_lights(void* instance, const rfw_directional_light* lights, uint32_t num, const uint32_t* changed);
/* :78 set_skybox */
RFW_HIP_API int rfw_hip_set_skybox(void* instance, const rfw_texture_data* skybox);
/* :81 set_skins */
RFW_HIP_API int rfw_hip_set_skins(void* instance, const rfw_skin_data* skins, uint32_t num, const uint32_t* changed);

/* ---- extensions: the trait presents to a swap chain and has no read-back, options or queries ---- */
/* gpu-rt's RenderMode::Reset (gpu-rt/src/lib.rs:1690-1692): restart accumulation. */
RFW_HIP_API int rfw_hip_reset_accumulation(void* instance);
/* Options (unknown keys are an error).  The trait has none: these are the knobs a host outside the trait may turn.
 *   rendering      "max_path_length" (1 = primary + shadow), "clamp_value", "nee" (0 / 1), "sample_count", "sky_r" / "sky_g" / "sky_b",
 *                  "texture_array" (gpu-rt's 1024^2 x 5 texture array, default 1)
 *   measurement    "count_traversal" (node / triangle / instance counters of the next frames), "timing" (HIP events per kernel)
 *   ray order      "shadow_order" 0 | 1 | 2 (which end any-hit traversals start from; the image is the same under every order),
 *                  "sort_extension_rays" 0 never | 1 always | 2 batches whose bounces do not stream (default),
 *                  "stream_run" r (0 or a power of two <= 64: a wavefront of the bounces' trace kernels owns r x 64 queue entries and refills
 *                  its idle lanes; setting it forces, the default 8 applies by "stream_auto" = 1 to single frames of an instance with >= 4 frame
 *                  slots and to batches), "stream_refill" (idle lanes that trigger a refill, 12), "stream_leaf_gate" (lanes at a leaf that
 *                  start the packet loop, 16) — images never depend on any of these
 *   builders       "sah_max_leaf", "sah_trav_cost", "build_threads" (host builder), "spill_rows" (test hook: rows of the HBM stack spill),
 *                  "spatial_splits" (threshold; 0 = off), "packet_trace" (bit 0 camera rays, bit 1 / 2 shadow rays as wavefront packets),
 *                  "shade_group" 0 | 256 | 512, "tlas_fused" 0 | 1 | 2 (the TLAS of <= 16 384 instances by one workgroup + one finishing launch:
 *                  never | always | where the instance has frame slots (default) — csrc/lbvh.hip, k_tlas_fused)
 *   multi-GPU      "gather_format" 0 | 1 | 2, "present_rank" r (see rfw_hip_shard_info), "p2p_timeout_ms" (see rfw_hip_p2p_*) */
RFW_HIP_API int rfw_hip_set_option(void* instance, const char* key, double value);
/* tonemapped frame, RGBA32F, sqrt(acc/samples) (backends/gpu-rt/shaders/blit.comp:15-23); n_floats = w*h*4 */
RFW_HIP_API int rfw_hip_read_framebuffer(void* instance, float* rgba, uint64_t n_floats);
/* raw accumulator (acPixels), RGBA32F sums */
RFW_HIP_API int rfw_hip_read_accumulator(void* instance, float* rgba, uint64_t n_floats);
RFW_HIP_API int rfw_hip_get_frame_stats(void* instance, rfw_hip_frame_stats* out);
/* Sums the per-kernel HIP-event timings of every frame rendered since the previous drain (at most 64 frames are kept)
 * into `sum` (ms_* fields only) and reports how many frames that was.  One stream synchronisation per call, none per frame. */
RFW_HIP_API int rfw_hip_drain_timing(void* instance, rfw_hip_frame_stats* sum, uint32_t* frames);
RFW_HIP_API int rfw_hip_get_scene_stats(void* instance, rfw_hip_scene_stats* out);
/* launch all work on this hipStream_t (NULL = the instance's own stream) */
RFW_HIP_API int rfw_hip_set_stream(void* instance, void* hip_stream);
/* the hipStream_t all work of this instance is launched on (its own stream unless rfw_hip_set_stream replaced it) */
RFW_HIP_API void* rfw_hip_get_stream(void* instance);
RFW_HIP_API int rfw_hip_device_synchronize(void* instance);

/* Multi-GPU tile sharding.  With world > 1 an instance renders only the tiles dealt to `rank`.  Its contribution to
 * the frame is a compact slab of `slab_floats` floats: the RGB of the accumulator for each of its slab elements (alpha is never
 * written by the path tracer, so 12 B per pixel travel instead of 16).  After tracing, render() packs that slab into the device
 * buffer given to rfw_hip_set_slab_output (typically this rank's slice of the all-gather buffer); the caller all-gathers the
 * slabs (RCCL) into a buffer of world * slab_floats and hands it to rfw_hip_assemble_frame, which de-tiles it into the full
 * frame.  Accumulation over samples happens in the instance's own slab, not in the caller's buffer.  The linear accumulator of the
 * assembled frame (rfw_hip_read_accumulator*) is de-tiled on demand from the gathered buffer, which therefore has to stay valid
 * until the next assemble if that call is used.
 * rfw_hip_set_option("gather_format", f) chooses WHAT travels: 0 (default) the accumulator's RGB as floats, as described; 1 the FINISHED frame
 * sqrt(acc / samples) as three halves per pixel (6 B); 2 the PRESENTED frame, B, G, R, A bytes as rfw_hip_download_frame(what = 2) encodes it
 * (4 B: what the reference draws onto its swap chain).  `slab_floats` is then the number of 4-byte words per frame in that format.  With 1 and 2
 * the accumulators stay on the ranks that own the tiles: rfw_hip_read_accumulator* fails on a sharded instance, and with 2 the frame is
 * available through rfw_hip_download_frame(what = 2) only.
 * rfw_hip_set_option("present_rank", r): only rank r de-tiles a gathered frame at once (it presents); the other ranks keep the gathered tiles
 * and de-tile when a frame is read.  Default -1: every rank de-tiles every frame. */
RFW_HIP_API int rfw_hip_shard_info(void* instance, uint64_t* slab_floats, uint32_t* num_tiles_local, uint32_t* num_tiles_total);
/* device buffer (slab_floats floats; count * slab_floats for render_batch) render() leaves this rank's slab in; NULL = none */
RFW_HIP_API int rfw_hip_set_slab_output(void* instance, void* device_ptr);
RFW_HIP_API int rfw_hip_assemble_frame(void* instance, const void* gathered_device_ptr);

/* The collective inside the library (one process per GPU): a RCCL communicator owned by the instance.  Rank 0 obtains a
 * 128-byte id (rfw_hip_comm_unique_id = ncclGetUniqueId) and hands it to every rank by whatever means the host has; every rank creates
 * its instance with options.rank / options.world and calls rfw_hip_comm_init (collective: = ncclCommInitRank on the instance's device).
 * From then on rfw_hip_render / rfw_hip_render_batch leave the COMPLETE frame(s) on every rank: the rank's tiles are traced, the RGB of
 * its slab is packed, ONE ncclAllGather per call runs on the instance's own stream (RCCL over xGMI) and the gathered slabs are de-tiled —
 * no buffer, stream or collective on the host's side (rfw_hip_set_slab_output / rfw_hip_assemble_* stay for hosts that bring their own
 * collective, e.g. torch.distributed).  world = 1 is allowed (a one-rank communicator).  librccl is opened at run time, on first use.
 * Frame slots (options.frames_in_flight) share the instance's communicator: every slot gathers into buffers of its own on its own stream and the
 * collectives are chained on the device, so a rank pipelines sharded frames with ONE scene copy.  Not available with sub-streams. */
RFW_HIP_API int rfw_hip_comm_unique_id(void* out128);
RFW_HIP_API int rfw_hip_comm_init(void* instance, const void* id128, uint32_t rank, uint32_t world);
RFW_HIP_API int rfw_hip_comm_destroy(void* instance);
/* TEST TRANSPORT.  `world` instances of ONE process on ONE device, created with options.rank = 0 .. world - 1, join the hub `hub_key` instead of
 * a communicator: their render() calls then exchange tiles as with rfw_hip_comm_init — same packing, same per-slot gather buffers, same
 * ordering of the frame slots' collectives, same de-tiling — but the bytes are moved by device copies the LAST rank to call enqueues.  Every
 * rank must call render() for a frame before anyone waits for that frame; a rank that never does shows up as RFW_HIP_E_DEVICE ("a peer's
 * flag did not arrive", option p2p_timeout_ms) from the others' reads.  rfw_hip_comm_destroy leaves the hub.  No RCCL is involved. */
RFW_HIP_API int rfw_hip_comm_init_loopback(void* instance, uint64_t hub_key, uint32_t rank, uint32_t world);

/* The same exchange WITHOUT a collective library (SURVEY.md §8e's alternative): every rank stores its tiles straight into its peers'
 * receive buffers over xGMI.  xGMI is point to point, so the 7 links of a GPU carry the 7 peers' tiles side by side, where a ring
 * all-gather is bound by one link; and with present_rank = r the tiles travel to rank r ONLY (1 / world of the all-gather's bytes).
 *   rfw_hip_p2p_export   allocates this instance's receive buffers (per frame slot, per rank) and its flag words and writes a
 *                        RFW_HIP_P2P_HANDLE_BYTES handle: process id, device, the buffers' addresses and their hipIpcMemHandles
 *   (the host hands every rank's handle to every rank: one all-gather of 256 bytes, by whatever means it has)
 *   rfw_hip_p2p_connect  handles = world x RFW_HIP_P2P_HANDLE_BYTES in rank order.  A peer of this process is reached through its address
 *                        (hipDeviceEnablePeerAccess when it is on another device), a peer of another process through hipIpcOpenMemHandle
 *   (a host barrier: nobody renders before every rank has connected)
 * From then on rfw_hip_render / rfw_hip_render_batch pack this rank's slab(s) in the gather format DIRECTLY into the destinations'
 * buffers, raise the destinations' arrival flags (system-scope release stores), and a destination waits (one wavefront polling its OWN
 * memory) for all world flags before it de-tiles; afterwards it returns a credit to every sender, which a sender waits for before it
 * overwrites that slot's buffer.  All remote traffic is stores.  A wait gives up after option "p2p_timeout_ms" (default 5000): the
 * next read then fails instead of the device hanging.  On a rank that is not a destination the frame does not exist: reads fail.
 * rfw_hip_p2p_disconnect (after a host barrier) unmaps and frees.  The instance's size cannot change while connected. */
#define RFW_HIP_P2P_HANDLE_BYTES 256
RFW_HIP_API int rfw_hip_p2p_export(void* instance, void* handle_out);
RFW_HIP_API int rfw_hip_p2p_connect(void* instance, const void* handles);
RFW_HIP_API int rfw_hip_p2p_disconnect(void* instance);

/* Ray queries against the synchronized scene — the C form of the reference's CPU query
 * interface TIntersector::{intersect, occludes} (crates/rfw-scene/src/intersector.rs:45-75,
 * 21-43).  Host pointers; origins/directions are n x 3 floats. */
RFW_HIP_API int rfw_hip_intersect(void* instance, const float* origins, const float* directions,
                                  float t_min, float t_max, uint64_t n, rfw_hip_hit* hits);
RFW_HIP_API int rfw_hip_occludes(void* instance, const float* origins, const float* directions,
                                 float t_min, const float* t_max, uint64_t n, uint8_t* occluded);
/* TIntersector::depth_test (intersector.rs:103-127): the closest hit and, per ray, how many BVH nodes the query visited
 * (4-wide nodes of THIS backend's trees, top level and meshes; like the reference's number it depends on the builder). */
RFW_HIP_API int rfw_hip_depth_test(void* instance, const float* origins, const float* directions,
                                   float t_min, float t_max, uint64_t n, rfw_hip_hit* hits, uint32_t* depth);

/* TIntersector::intersect4 / occludes4 (intersector.rs:129-166): the reference's 4-wide CPU packets.  SoA as rtbvh's RayPacket4 —
 * origin_xyz4 = x[4] y[4] z[4], direction_xyz4 likewise; t4 = far limits on entry, hit distances on return; ids -1 for a miss.  On this
 * backend a packet is four single-ray queries (each lane may have its own interval); batches of rays belong in rfw_hip_intersect /
 * rfw_hip_occludes.  The reference's occludes4 is a stub that answers `true` four times (intersector.rs:129-131); this one answers.
 * TIntersector::get_hit_record has no counterpart: the mesh side of it is commented out in the reference (crates/rfw-backend/src/structs.rs:1253). */
RFW_HIP_API int rfw_hip_intersect4(void* instance, const float* origin_xyz4, const float* direction_xyz4, const float* t_min4, float* t4,
                                   int32_t* instance_ids4, int32_t* prim_ids4);
RFW_HIP_API int rfw_hip_occludes4(void* instance, const float* origin_xyz4, const float* direction_xyz4, const float* t_min4, const float* t_max4,
                                  uint8_t* occluded4);

/* Debug read-back of the wavefront queues after the last render() bounce `bounce`
 * (test-only; enabled by option "keep_queues"=1).  Layout documented in DESIGN.md. */
RFW_HIP_API int rfw_hip_debug_read(void* instance, const char* what, void* dst, uint64_t bytes, uint64_t* written);

/* occludes() that also reports how many 4-wide nodes each any-hit traversal visited (the any-hit counterpart of rfw_hip_depth_test).  For
 * the planning probes under tools/probes (wave-occupancy models from real per-ray traversal lengths); not part of the trait. */
RFW_HIP_API int rfw_hip_debug_occludes_depth(void* instance, const float* origins, const float* directions, float t_min, const float* t_max,
                                             uint64_t num_rays, uint8_t* occluded, uint32_t* depth);
/* Test-only: stress test of the device LBVH builder that rebuilds the TLAS every frame (its bottom-up fit hands a subtree's box from one
 * thread to another without fences: csrc/lbvh.hip, k_fit).  `iterations` times: num_boxes jittered boxes -> tree -> exact structural check
 * on the device (every child box equals the union of what lies below it; every box in exactly one leaf), all queued on the instance's
 * stream; *errors = mismatches found, *checked = child boxes compared.  Environment RFW_LBVH_FENCED=1 selects the fenced fit. */
RFW_HIP_API int rfw_hip_debug_lbvh_stress(void* instance, uint32_t num_boxes, uint32_t iterations, uint32_t seed, uint64_t* errors, uint64_t* checked);

/* Test-only: the device functions the shade kernel is made of, evaluated one by one on caller-supplied inputs, so that each can be held
 * against an independent formulation (tests/test_shading_kat.py compares with numpy float64).  Host pointers; per case 48 input floats:
 *   [0,24) one rfw_device_material (its 96 bytes)  [24,27) N  [27,30) wo (op 3: D; op 4: the shaded point I)  [30,33) wi  [33,36) T
 *   [36,39) B  [39] t  [40] backfacing (0/1)  [41] r3 (op 4: r0)  [42] r4  [43] light area (op 3)
 * and 12 output floats.  op 0: BSDFEval -> rgb (gpu-rt/shaders/disney.glsl:110-195); 1: BSDFPdf -> pdf (:89-108); 2: BSDFSample -> wi.xyz,
 * pdf, type (:197-263); 3: CalculateLightPDF -> pdf (shade.comp:325-328); 4: RandomPointOnLight with the lights set on this instance
 * (synchronize first) -> P.xyz, pickProb, lightPdf, colour.rgb, picked light (shade.comp:413-528); 5: RandomBarycentrics(r0 = [41]) -> barycentrics
 * (shade.comp:371-411). */
RFW_HIP_API int rfw_hip_debug_eval_shading(void* instance, int op, uint64_t n, const float* in48, float* out12);

/* A batch of `count` independent NEW images, one per view, traced as one tall virtual frame: every stage of the wavefront loop is ONE
 * launch over the paths of all frames (bigger launches, fewer of them; with a sharded frame also ONE all-gather per batch).  Each
 * frame is exactly what rfw_hip_render of that view on a freshly reset instance produces.  count <= options.max_batch; the views must
 * share one spread angle (same field of view and height).  Afterwards frame f is read with the _at functions; with world > 1 the slab
 * written is [frame][slab], the gathered buffer handed to rfw_hip_assemble_batch is [rank][frame][slab]. */
RFW_HIP_API int rfw_hip_render_batch(void* instance, const rfw_camera_view_3d* views, uint32_t count);
RFW_HIP_API int rfw_hip_assemble_batch(void* instance, const void* gathered_device_ptr, uint32_t count);
/* `count` consecutive SAMPLES of the one image of `view` in one launch per stage (count <= options.max_batch): sample indices
 * n .. n + count - 1 where n = samples accumulated so far (0 after a view / scene change or a reset, exactly like rfw_hip_render), each
 * traced into its own slab and then added to the image's accumulator in sample order.  The same paths, random numbers and contributions
 * as `count` rfw_hip_render calls (gpu-rt's loop of one sample per render(), gpu-rt/src/lib.rs:1685-1731); the only difference is where
 * the per-pixel partial sums are rounded: sum_f(sample f) instead of sequential accumulation — equal to a few ulp (tests: <= 1e-6 rel.
 * L2), and bit-identical to the sum of the per-sample images.  C4's "4 spp" is one call. */
RFW_HIP_API int rfw_hip_render_samples(void* instance, const rfw_camera_view_3d* view, uint32_t count);
/* The tables of the reference's blue-noise sampler (gpu-rt/shaders/ray_gen.comp:72-91, shade.comp:530-545), which it uses for the
 * first 256 samples of every image (ray_gen.comp:109-122, shade.comp:189-227): `table` = the 5 * 65536 words
 * gpu_rt::blue_noise::create_blue_noise_buffer() returns (backends/gpu-rt/src/blue_noise.rs:40970-41005; every word a byte value),
 * copied before the call returns.  The Rust shim passes them once after rfw_hip_create; n_words = 0 clears them.
 * Without tables every sample draws from the xorshift generator (the reference's branch for samples >= 256).  Restarts accumulation. */
RFW_HIP_API int rfw_hip_set_blue_noise(void* instance, const uint32_t* table, uint32_t n_words);
RFW_HIP_API int rfw_hip_read_framebuffer_at(void* instance, uint32_t frame, float* rgba, uint64_t n_floats);
RFW_HIP_API int rfw_hip_read_accumulator_at(void* instance, uint32_t frame, float* rgba, uint64_t n_floats);

/* Frames to host memory without stalling the pipeline.  rfw_hip_download_frame queues the copy of the latest frame (what = 0: the
 * finalised frame, 1: the accumulator; `frame` = index inside the last batch, 0 otherwise) behind the kernels that produce it, on
 * that frame's stream, and returns at once; with frames in flight the DMA of frame k overlaps the tracing of the following frames.
 * what = 2 is the PRESENTED frame: what gpu-rt's final pass leaves on its Bgra8UnormSrgb swap chain (gpu-rt/src/lib.rs:373,560-585,
 * shaders/quad.frag) — the finalised frame clamped, sRGB-encoded and quantised, one B,G,R,A byte quadruple per pixel (alpha 255),
 * n_floats = width * height 32-bit words; a quarter of the bytes of the float frame on the PCIe link.  rfw_hip_srgb_steps returns
 * the 255 linear values at which the encoded byte steps (the encoder compares against them, so it is exact everywhere).
 * The destination should come from rfw_hip_host_alloc (pinned; a pageable buffer makes the copy synchronous).  The bytes are valid
 * after rfw_hip_wait_downloads, which waits for every copy queued so far (and for nothing else). */
RFW_HIP_API void rfw_hip_srgb_steps(float* out255);
RFW_HIP_API void* rfw_hip_host_alloc(uint64_t bytes);
RFW_HIP_API void rfw_hip_host_free(void* ptr);
RFW_HIP_API int rfw_hip_download_frame(void* instance, uint32_t what, uint32_t frame, float* host_rgba, uint64_t n_floats);
RFW_HIP_API int rfw_hip_wait_downloads(void* instance);
/* waits only for the copy into host_ptr (e.g. before a ring of host buffers hands that buffer out again) */
RFW_HIP_API int rfw_hip_wait_download(void* instance, const void* host_ptr);

/* Measured HBM roofline for this device in this job: a float4 device-to-device copy of `bytes` bytes
 * (rounded down to 16), repeated `iterations` times on the instance's stream and timed with HIP events.
 * *gb_per_s = 2 * bytes * iterations / time (bytes read + bytes written).  Allocates and frees 2 * bytes of HBM. */
RFW_HIP_API int rfw_hip_bandwidth_probe(void* instance, uint64_t bytes, uint32_t iterations, double* gb_per_s);

/* Measured vector-issue ceiling of this device in this job, for the roofline of the (issue-bound) trace kernels: 8 wavefronts per SIMD on
 * every CU run `trips` x 32 independent instructions each, timed with HIP events; *g_instructions_per_s = wave64 instructions retired per
 * second, chip-wide, in units of 1e9.  mix 0: v_fma_f32 alone (what the FP32 peak of the data sheet is quoted on).  mix 1: the
 * instruction mix of one child of the PER-LANE 4-wide node test (6 byte->float conversions, 3 packed FMAs, max3, min3, min, 2 compares), most of
 * which issue at half the FMA rate or less on gfx950.  mix 2: one node step of the PACKET kernel the camera rays run (per child 6 v_fma_f32
 * with a scalar operand, max3, min3, min, max, one compare into a scalar register pair) with the step's 27 scalar instructions issued beside
 * the 44 vector ones; the VECTOR instructions per second are returned.  Nothing of the scene is touched. */
RFW_HIP_API int rfw_hip_issue_probe(void* instance, int mix, uint32_t trips, double* g_instructions_per_s);

#ifdef __cplusplus
}
#endif
#endif /* RFW_HIP_H */
