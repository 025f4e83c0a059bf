// api_exchange.cpp — the sharded frame: what a rank sends and what it does with what it receives — packing, the RCCL all-gather
// inside the library (rfw_hip_comm_*), the exchange by peer stores (rfw_hip_p2p_*), de-tiling.
#include "api_internal.h"

using namespace rfwapi;

namespace rfwapi {
Rccl g_rccl;
std::mutex g_rccl_mu;
// k == 1: one sample of the image for views[0].  k > 1 (rfw_hip_render_batch): k independent NEW images, one per view, traced as one
// tall virtual frame — every stage is ONE launch over the paths of all k frames.
// `samples` (rfw_hip_render_samples): the k frames are k consecutive SAMPLES of the one image of views[0] — sample indices sample_count …
// sample_count + k - 1, each traced into its own slab, then summed into slab 0 in sample order.
// ---- sharded frame: what a rank sends, and what it does with what it receives
uint64_t slab_words(const Instance* I) // 4-byte words one frame of this rank contributes to the all-gather
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
bool gathers_tiles(const Instance* I) { return scene_of(I)->comm != nullptr || scene_of(I)->loop != nullptr || I->external_slab != nullptr || scene_of(I)->p2p.connected; }
bool p2p_timed_out(const Instance* I) { return (scene_of(I)->p2p.connected || scene_of(I)->loop) && I->overflow_host && ((volatile const uint32_t*)I->overflow_host)[1] != 0u; }
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
    if (p2p_timed_out(I)) return fail(I, RFW_HIP_E_DEVICE, "exchange: a peer's flag did not arrive within p2p_timeout_ms (a rank did not take part: the frame is incomplete)");
    if (!I->deferred.gathered) return RFW_HIP_OK;
    return assemble_gathered(I, I->stream, I->deferred.gathered, I->deferred.k, I->deferred.samples);
}

} // namespace rfwapi

extern "C" {

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
    if (I->loop) {
        HIP_TRY(I, hipSetDevice(I->device));
        HIP_TRY(I, hipStreamSynchronize(I->stream));
        for (Instance* c : I->slots) HIP_TRY(I, hipStreamSynchronize(c->stream));
        loop_leave(I);
        I->acc_source = nullptr;
        return RFW_HIP_OK;
    }
    if (!I->comm) return RFW_HIP_OK;
    HIP_TRY(I, hipSetDevice(I->device));
    HIP_TRY(I, hipStreamSynchronize(I->stream));
    for (Instance* c : I->slots) HIP_TRY(I, hipStreamSynchronize(c->stream));
    (void)g_rccl.comm_destroy(I->comm);
    I->comm = nullptr;
    I->acc_source = nullptr;
    return RFW_HIP_OK;
}

} // extern "C"

// ---- the loop-back transport (rfw_hip_comm_init_loopback): W instances of ONE process, one per rank, exchange their tiles through a hub
// instead of a communicator.  It exists so that everything around the collective — the packing, the per-slot gather buffers, the event chain
// that orders the slots' collectives (comm_chain), de-tiling on the presenting rank, a rank that never arrives — runs where there is one
// GPU and no second process.  Semantics of ncclAllGather on a stream: the call returns at once; what follows it on the
// caller's stream runs when every rank's contribution is in every rank's receive buffer.  Here: the caller records "my tiles are packed",
// enqueues a wait for its own flag word, and the LAST rank to arrive (host side) enqueues the W x W copies on the hub's stream behind
// every rank's event, then sets every rank's flag.
namespace rfwapi {
struct LoopHub {
    std::mutex mu;
    uint32_t world = 0, n_slots = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    uint32_t* flags = nullptr; // [slot][rank]: gathers completed
    std::vector<Instance*> owner; // per rank (nullptr: not joined / left)
    struct Call { Instance* inst; uint64_t words; };
    std::vector<std::vector<Call>> pending; // [slot]
    std::vector<uint32_t> done;             // [slot]
};
static std::mutex g_hubs_mu;
static std::map<uint64_t, LoopHub*> g_hubs;

int loop_all_gather(Instance* I, hipStream_t s, uint64_t n_words)
{
    Instance* C = scene_of(I);
    LoopHub* H = C->loop;
    const uint32_t slot = I->slot_index;
    if (slot >= H->n_slots) return fail(I, RFW_HIP_E_STATE, "loop-back exchange: this instance has more frame slots than the hub was made for");
    if (!I->loop_sent) HIP_TRY(I, hipEventCreateWithFlags(&I->loop_sent, hipEventDisableTiming));
    HIP_TRY(I, hipEventRecord(I->loop_sent, s)); // (behind pack_slabs)
    const uint32_t want = ++I->loop_seq;
    launch_p2p_wait(s, H->flags + (size_t)slot * H->world, I->rank, 1u, want, C->p2p.timeout_ticks, I->overflow_dev + 1);
    std::lock_guard<std::mutex> g(H->mu);
    auto& calls = H->pending[slot];
    for (const auto& c : calls)
        if (c.inst->rank == I->rank) return fail(I, RFW_HIP_E_STATE, "loop-back exchange: this rank joined the slot's gather twice before the others arrived");
    calls.push_back({I, n_words});
    if (calls.size() < H->world) return RFW_HIP_OK;
    // the last rank of this gather: move everything, on the hub's stream, behind every rank's packing
    for (const auto& c : calls) {
        if (c.words != n_words) { calls.clear(); return fail(I, RFW_HIP_E_STATE, "loop-back exchange: the ranks contribute slabs of different sizes"); }
        HIP_TRY(I, hipStreamWaitEvent(H->stream, c.inst->loop_sent, 0));
    }
    for (const auto& dst : calls)
        for (const auto& src : calls)
            HIP_TRY(I, hipMemcpyAsync(dst.inst->d_recv.ptr + (size_t)src.inst->rank * n_words, src.inst->d_send.ptr, n_words * sizeof(float), hipMemcpyDeviceToDevice, H->stream));
    P2PTargets t;
    for (uint32_t r = 0; r < H->world; r++) t.p[r] = H->flags + (size_t)slot * H->world + r;
    launch_p2p_signal(H->stream, t, H->world, ++H->done[slot]);
    HIP_TRY(I, hipGetLastError());
    calls.clear();
    return RFW_HIP_OK;
}
void loop_leave(Instance* owner)
{
    LoopHub* H = owner->loop;
    if (!H) return;
    owner->loop = nullptr;
    bool last = true;
    {
        std::lock_guard<std::mutex> g(H->mu);
        H->owner[owner->rank] = nullptr;
        for (auto& calls : H->pending) // a gather this rank had joined and the others had not: it will never complete
            calls.erase(std::remove_if(calls.begin(), calls.end(), [&](const LoopHub::Call& c) { return scene_of(c.inst) == owner; }), calls.end());
        for (Instance* o : H->owner) last = last && o == nullptr;
    }
    if (!last) return;
    std::lock_guard<std::mutex> g(g_hubs_mu);
    for (auto it = g_hubs.begin(); it != g_hubs.end(); ++it)
        if (it->second == H) { g_hubs.erase(it); break; }
    (void)hipSetDevice(H->device);
    if (H->stream) { (void)hipStreamSynchronize(H->stream); (void)hipStreamDestroy(H->stream); }
    if (H->flags) (void)hipFree(H->flags);
    delete H;
}
} // namespace rfwapi

extern "C" {
int rfw_hip_comm_init_loopback(void* inst, uint64_t hub_key, uint32_t rank, uint32_t world)
{
    LOCK(inst);
    if (world == 0 || world > 16 || rank >= world) return fail(I, RFW_HIP_E_INVALID, "comm_init_loopback: bad arguments (1 <= world <= 16, rank < world)");
    if (rank != I->rank || world != I->world) return fail(I, RFW_HIP_E_INVALID, "comm_init_loopback: rank / world differ from the shard this instance was created with (rfw_hip_options.rank / world)");
    if (I->substreams > 1) return fail(I, RFW_HIP_E_STATE, "comm_init_loopback: an instance with sub-streams cannot exchange through a hub");
    if (I->comm || I->loop || I->p2p.data) return fail(I, RFW_HIP_E_STATE, "comm_init_loopback: this instance already has an exchange");
    HIP_TRY(I, hipSetDevice(I->device));
    LoopHub* H = nullptr;
    {
        std::lock_guard<std::mutex> g(g_hubs_mu);
        auto it = g_hubs.find(hub_key);
        if (it == g_hubs.end()) {
            H = new LoopHub();
            H->world = world; H->n_slots = (uint32_t)I->slots.size() + 1u; H->device = I->device;
            H->owner.assign(world, nullptr); H->pending.resize(H->n_slots); H->done.assign(H->n_slots, 0u);
            const size_t fb = (size_t)H->n_slots * world * sizeof(uint32_t);
            hipError_t e = hipStreamCreateWithFlags(&H->stream, hipStreamNonBlocking);
            if (e == hipSuccess && hipExtMallocWithFlags((void**)&H->flags, fb, hipDeviceMallocUncached) != hipSuccess) {
                (void)hipGetLastError();
                e = hipMalloc((void**)&H->flags, fb);
            }
            if (e == hipSuccess) e = hipMemset(H->flags, 0, fb);
            if (e != hipSuccess) {
                if (H->stream) (void)hipStreamDestroy(H->stream);
                if (H->flags) (void)hipFree(H->flags);
                delete H;
                return fail(I, RFW_HIP_E_DEVICE, std::string("comm_init_loopback: ") + hipGetErrorString(e));
            }
            g_hubs[hub_key] = H;
        } else H = it->second;
    }
    std::lock_guard<std::mutex> g(H->mu);
    if (H->world != world || H->device != I->device || H->n_slots != I->slots.size() + 1u)
        return fail(I, RFW_HIP_E_INVALID, "comm_init_loopback: the hub was made for another world size, device or number of frame slots");
    if (H->owner[rank]) return fail(I, RFW_HIP_E_STATE, "comm_init_loopback: this rank of the hub is taken");
    const size_t n = (size_t)I->capacity * I->max_batch * 3u; // (room for the widest format: the option may still change)
    for (uint32_t k = 0; k <= I->slots.size(); k++) { // as rfw_hip_comm_init: every slot gathers into buffers of its own
        Instance* c = slot_ptr(I, k);
        HIP_TRY(I, c->d_send.ensure(n));
        HIP_TRY(I, c->d_recv.ensure(n * world));
        HIP_TRY(I, hipMemsetAsync(c->d_recv.ptr, 0, n * world * sizeof(float), c->stream));
        c->sample_count = 0;
        c->loop_seq = 0;
    }
    if (!I->slots.empty() && !I->comm_chain) HIP_TRY(I, hipEventCreateWithFlags(&I->comm_chain, hipEventDisableTiming));
    I->comm_chain_pending = false;
    H->owner[rank] = I;
    I->loop = H;
    return RFW_HIP_OK;
}
} // extern "C"

// ---- the exchange by peer stores
namespace rfwapi {
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
} // namespace rfwapi

extern "C" {

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
    if (env_switches().p2p_data_cached || hipExtMallocWithFlags((void**)&P.data, P.n_slots * P.slot_words * sizeof(uint32_t), hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        P.data = nullptr;
        HIP_TRY(I, hipMalloc((void**)&P.data, P.n_slots * P.slot_words * sizeof(uint32_t)));
    }
    // flag words are polled while peers write them: uncached, so that a poll never reads a stale line of this device's L2
    // (RFW_P2P_FLAGS_FINEGRAINED=1 forces the fall-back kind of memory, so that tests can take that path)
    if (env_switches().p2p_flags_finegrained || hipExtMallocWithFlags((void**)&P.flags, flag_bytes, hipDeviceMallocUncached) != hipSuccess) {
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

} // extern "C"
