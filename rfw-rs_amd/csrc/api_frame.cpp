// api_frame.cpp — instance life cycle and the per-frame launch sequence of the C ABI (backends/gpu-rt/src/lib.rs:1685-1780 render): frame slots,
// batches, samples, reads and downloads, options, timing.
#include "api_internal.h"

using namespace rfwapi;

namespace rfwapi {
thread_local std::string g_create_error;
} // namespace rfwapi
namespace rfwhip {
static EnvSwitches g_env;
const EnvSwitches& env_switches() { return g_env; }
void read_env_switches()
{
    EnvSwitches e;
    auto set = [](const char* name) { const char* v = getenv(name); return v != nullptr && std::strcmp(v, "0") != 0; };
    e.build_trace = set("RFW_BUILD_TRACE");
    e.no_forest = set("RFW_NO_FOREST");
    e.lbvh_fenced = set("RFW_LBVH_FENCED");
    e.p2p_data_cached = set("RFW_P2P_DATA_CACHED");
    e.p2p_flags_finegrained = set("RFW_P2P_FLAGS_FINEGRAINED");
    if (const char* v = getenv("RFW_PACKET_AUTO_MAX_TRIANGLES")) e.packet_auto_max_triangles = strtoull(v, nullptr, 10);
    if (const char* v = getenv("RFW_SPATIAL_SPLITS")) { e.has_spatial_splits = true; e.spatial_splits = (float)std::max(0.0, atof(v)); }
    if (const char* v = getenv("RFW_PACKET_TRACE")) e.packet_trace = std::max(0, atoi(v));
    if (const char* v = getenv("RFW_NODE_ORDER")) e.node_order = std::max(0, atoi(v));
    static std::mutex mu; // (instances may be created from several threads)
    std::lock_guard<std::mutex> g(mu);
    g_env = e;
}
} // namespace rfwhip
namespace rfwapi {
// m with n / d == (n * m) >> 32 for every n <= n_max, or 0 when the round-up reciprocal m = ceil(2^32 / d) is not exact that far
// (n * m = n * 2^32 / d + n * e / d with e = m * d - 2^32 < d: exact while n_max * e < 2^32).  The kernels divide when they get 0.
uint32_t index_magic(const uint32_t d, const uint64_t n_max)
{
    if (d < 2u) return 0u;
    const uint64_t m = ((1ull << 32) + d - 1u) / d, e = m * d - (1ull << 32);
    if (m > 0xffffffffull || n_max >= (1ull << 32)) return 0u;
    return (e == 0 || n_max < (1ull << 32) / e) ? (uint32_t)m : 0u;
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
    HIP_TRY(I, I->d_counters.ensure(2 * kMaxSub));
    HIP_TRY(I, hipMemsetAsync(I->d_counters.ptr, 0, 2 * kMaxSub * sizeof(QueueCounters), I->stream));
    I->sample_count = 0;
    return RFW_HIP_OK;
}

uint32_t spill_stride(const Instance* I) { return (uint32_t)(I->d_spill.cap / kStackSpill); }

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
    s.tlas_wide_stride = (uint32_t)(TL->d_tlas_oct.cap / kPacketNodeCopies); // (both forms of the copies share the stride; the packet form may be absent)
    s.blas_wide_stride = (uint32_t)(S->d_blas_oct.cap / kPacketNodeCopies);
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
    s.counters = I->d_counters.ptr + (size_t)I->counter_phase * kMaxSub; // (the latest frame's block)
    s.counters_next = nullptr;
    return s;
}

CameraParams camera_params(const Instance* I, const rfw_camera_view_3d& v, uint32_t sub)
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
    c.tile_shift = (I->tile_size & (I->tile_size - 1u)) == 0u ? (uint32_t)__builtin_ctz(I->tile_size) : 0xffffffffu;
    c.width_magic = index_magic(I->width, (uint64_t)I->width * I->height);
    // (slab indices run a little past the last tile: idx < capacity of the rank's slab, tile = lt * world + rank)
    c.tiles_x_magic = index_magic(I->tiles_x, (uint64_t)I->tiles_x * I->tiles_y + 4096ull * std::max<uint64_t>((uint64_t)I->world * I->substreams, 1ull));
    // (api_internal.h: far outside the caches one ray per lane wins)
    if (S->packet_auto && S->n_tris > packet_auto_limit()) c.flags &= ~kFlagPacketPrimary;
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

PathDev path_dev(Instance* I, uint32_t sub)
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

int do_render(Instance* I, const rfw_camera_view_3d* views, uint32_t k, bool samples)
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
    // queue counters: this frame takes the block the previous frame's k_primary cleared (alloc_paths cleared both), and clears the other
    I->counter_phase ^= 1u;
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
        sc[s].counters = I->d_counters.ptr + (size_t)I->counter_phase * kMaxSub + s;
        sc[s].counters_next = I->d_counters.ptr + (size_t)(I->counter_phase ^ 1u) * kMaxSub + s;
        sc[s].spill = I->d_spill.ptr + (size_t)s * (I->cap_v + kSpillMargin);
        p[s] = path_dev(I, s);
        cam[s] = camera_params(I, view, s);
        {   // k_shade's workgroup size (kernels.hip, k_shade): small where other frames' kernels share the chip with this call
            const Instance* O = scene_of(I);
            const int g = O->shade_group;
            if (g == 256 || (g == 0 && k == 1 && !O->slots.empty())) cam[s].flags |= kFlagShadeSmallGroups;
        }
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
    if (Instance* C = scene_of(I); C->comm || C->loop) {
        // the frame's ONE collective, issued by the library itself: this rank's slab(s) -> all ranks (RCCL over xGMI) -> de-tile
        const uint64_t n_send = slab_words(I) * frames_out; // 4-byte words, whatever they hold
        pack_slabs(I, main, I->d_send.ptr, frames_out);
        if (C->comm_chain && C->comm_chain_pending) HIP_TRY(I, hipStreamWaitEvent(main, C->comm_chain, 0)); // behind the previous slot's collective
        if (C->loop) { // (the test transport: same buffers, same ordering around it)
            const int lrc = loop_all_gather(I, main, n_send);
            if (lrc != RFW_HIP_OK) return lrc;
        } else {
            const ncclResult_t nr = g_rccl.all_gather(I->d_send.ptr, I->d_recv.ptr, n_send, ncclFloat, C->comm, main);
            if (nr != ncclSuccess) return fail(I, RFW_HIP_E_DEVICE, std::string("ncclAllGather: ") + g_rccl.error_string(nr));
        }
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

} // namespace rfwapi

extern "C" {

uint32_t rfw_hip_abi_version(void) { return RFW_HIP_ABI_VERSION; }
uint32_t rfw_hip_selftest_index_magic(uint32_t d, uint64_t n_max) { return index_magic(d, n_max); }


void* rfw_hip_create(uint32_t width, uint32_t height, double /*scale*/, const rfw_hip_options* o)
{
    read_env_switches(); // (the one place the library asks the environment: env_switches.h)
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
        if (o->struct_size >= offsetof(rfw_hip_options, frames_in_flight) + sizeof(uint32_t)) n_slots = std::min<uint32_t>(std::max<uint32_t>(o->frames_in_flight, 1u), 24u);
        if (o->struct_size >= offsetof(rfw_hip_options, max_batch) + sizeof(uint32_t)) I->max_batch = std::min<uint32_t>(std::max<uint32_t>(o->max_batch, 1u), (uint32_t)kMaxBatch);
    }
    {
        const EnvSwitches& env = env_switches();
        if (env.has_spatial_splits) I->split_tau = env.spatial_splits; // A/B runs: the default of option "spatial_splits"
        const int pt = env.packet_trace >= 0 ? env.packet_trace : kDefaultPacketTrace; // A/B runs: the default of option "packet_trace"
        I->packet_auto = env.packet_trace < 0;
        if (pt & 1) I->flags |= kFlagPacketPrimary;
        if (pt & 2) I->flags |= kFlagPacketShadow;
        if (pt & 4) I->flags |= kFlagPacketShadowFar;
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
        I->d_valid_gids.release(); I->d_stage_dev.release(); I->d_tlas_order.release(); I->d_node_count.release(); I->d_inst_boxes.release(); I->d_mesh_local.release();
        I->d_tri_boxes.release(); I->d_lbvh_ws.release(); I->d_blas_order.release();
        I->d_q_o.release(); I->d_q_d.release(); I->d_q_t.release(); I->d_q_h.release(); I->d_q_depth.release(); I->d_q_r.release();
        if (I->comm) { (void)g_rccl.comm_destroy(I->comm); I->comm = nullptr; }
        if (I->loop) loop_leave(I);
        if (I->loop_sent) { (void)hipEventDestroy(I->loop_sent); I->loop_sent = nullptr; }
        if (I->comm_chain) (void)hipEventDestroy(I->comm_chain);
        for (auto& ev : I->ev_build)
            if (ev) (void)hipEventDestroy(ev);
        if (I->records_stream) (void)hipStreamSynchronize(I->records_stream);
        if (I->ev_heads) (void)hipEventDestroy(I->ev_heads);
        if (I->ev_records) (void)hipEventDestroy(I->ev_records);
        if (I->records_stream) (void)hipStreamDestroy(I->records_stream);
        I->d_heads.release();
        I->meshes.clear(); // (unregisters the pinned host copies while the runtime is still up)
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
    if (arc == RFW_HIP_OK && (scene_of(I)->comm || scene_of(I)->loop)) { // the gather buffers follow the slab size
        const size_t n = (size_t)I->capacity * I->max_batch * 3u;
        HIP_TRY(I, I->d_send.ensure(n));
        HIP_TRY(I, I->d_recv.ensure(n * I->world));
    }
    return arc;
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
        I->flags &= ~(kFlagPacketPrimary | kFlagPacketShadow | kFlagPacketShadowFar);
        I->packet_auto = false; // an explicit choice holds whatever the scene's size
        if ((int)value & 1) I->flags |= kFlagPacketPrimary;
        if ((int)value & 2) I->flags |= kFlagPacketShadow;
        if ((int)value & 4) I->flags |= kFlagPacketShadowFar; // bit 2: packets only for the buckets traced far to near (the directional lights)
        if (blas_wide_wanted(I, I->n_tris) && !I->d_blas_wide.ptr && I->n_tris) {
            // packets asked for on a scene that was built without their copies of the nodes: the next synchronize() rebuilds (until then the
            // camera rays keep going one per lane)
            for (auto& kv : I->meshes) kv.second.dirty = true;
            I->meshes_dirty = true;
            I->layout_valid = false;
        }
    }
    else if (k == "shade_group") { // threads per k_shade workgroup: 0 = automatic (do_render), 256, 512
        if ((int)value != 0 && (int)value != 256 && (int)value != 512) return fail(I, RFW_HIP_E_INVALID, "set_option: shade_group is 0, 256 or 512");
        I->shade_group = (int)value;
    }
    else if (k == "tlas_fused") I->tlas_fused = std::max(0, std::min(2, (int)value)); // 0: always the launch chain (lbvh_build); 1: always the one-workgroup build (up to 16 384 instances); 2: that where frames overlap (frame slots)
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
    else if (k == "spatial_splits") I->split_tau = (float)std::max(0.0, value); // meshes handed over from now on: a part of a triangle is cut while its box wastes more than this x the mesh box's area; 0 = off
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
    HIP_TRY(I, hipMemcpy(qc, I->d_counters.ptr + (size_t)I->counter_phase * kMaxSub, I->substreams * sizeof(QueueCounters), hipMemcpyDeviceToHost));
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

} // extern "C"
