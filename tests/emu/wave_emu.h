// wave_emu.h — TEST INFRASTRUCTURE: a lockstep-free emulator of one gfx950 workgroup on CPU threads, for running the DEVICE SOURCE of a
// wave-synchronous kernel (the text of rfw-rs_amd/csrc/sah_build.hip's phase 2, cut out by tests/test_builder_emulated.py) under g++.
//
// A lane is a fiber of its wavefront's OS thread; with -DEMU_THREADS (for the sanitizers) an OS thread of its own.  Everything that crosses lanes
// on the GPU is a collective here — the lanes of a wavefront publish their operand, meet, read their source lane's: __ballot, __shfl,
// __shfl_xor / _up / _down, readlane / readfirstlane, the DPP row operations (update_dpp with row_shr / row_shl / row_bcast:15 / row_bcast:31,
// the controls the builders use), and the wave barrier of wave_sync().  LDS is a function-local static (one workgroup at a time), LDS and
// global atomics are __atomic builtins, __syncthreads is a barrier over the workgroup.
// Fibers: a lane runs until its next meeting, then the next lane does — a fixed schedule, fast (no kernel sleeps inside a wavefront), good
// for what a kernel COMPUTES.  Threads: the lanes run free between collectives, so an LDS hand-over between lanes that the kernel forgot to
// fence with wave_sync() (on the GPU: left to instruction order and the compiler's mercy) is a data race -fsanitize=thread reports.
// Limits: a lane that waits at a collective for lanes that have returned is an error (the device would go on); a kernel is run for one
// blockIdx at a time.
#pragma once
#include <algorithm>
#include <barrier>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <thread>
#include <vector>

#ifndef EMU_THREADS
#include <sys/mman.h>
#include <ucontext.h>
#endif

#define __device__
#define __global__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)

namespace emu {
struct Dim { unsigned x = 0, y = 0, z = 0; };

#ifdef EMU_THREADS
// ---- one OS thread per lane (for the sanitizers: -fsanitize=thread sees every unfenced hand-over between lanes as a race)
struct WaveCtx {
    std::barrier<> bar{64};
    uint32_t slot[2][64]; // two sets, used in turn: ONE meeting per collective (a lane can only be writing set p again after every lane has
                          // arrived at the collective in between, i.e. has finished reading set p)
};
struct GroupCtx {
    std::unique_ptr<std::barrier<>> bar;
    std::vector<std::unique_ptr<WaveCtx>> waves;
};
inline thread_local Dim t_thread, t_block;
inline thread_local WaveCtx* t_wave = nullptr;
inline thread_local unsigned t_phase = 0; // the same in every lane of a wavefront: every lane takes part in every collective
inline thread_local GroupCtx* t_group = nullptr;
inline Dim& thread_idx() { return t_thread; }
inline Dim& block_idx() { return t_block; }
inline thread_local unsigned t_linear = 0;
inline unsigned lane() { return t_linear & 63u; }
inline uint32_t* next_slots() { return t_wave->slot[t_phase++ & 1u]; }
inline unsigned long long g_meetings = 0;
inline void meet() { t_wave->bar.arrive_and_wait(); }          // the lanes of my wavefront
inline void group_sync() { t_group->bar->arrive_and_wait(); }  // the threads of my workgroup

// run kernel() for one workgroup of `threads` threads with blockIdx.x = block
template <class K> inline void run_group(unsigned threads, unsigned block, K kernel, Dim block_dim = Dim{0, 1, 1}, Dim block_idx3 = Dim{0, 0, 0})
{
    GroupCtx g;
    if (block_dim.x == 0) { block_dim.x = threads; block_idx3.x = block; }
    g.bar = std::make_unique<std::barrier<>>((std::ptrdiff_t)threads);
    for (unsigned w = 0; w < (threads + 63) / 64; w++) g.waves.push_back(std::make_unique<WaveCtx>());
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            t_linear = t; t_thread.x = t % block_dim.x; t_thread.y = t / block_dim.x % block_dim.y; t_thread.z = t / (block_dim.x * block_dim.y);
            t_block = block_idx3; t_group = &g; t_wave = g.waves[t / 64].get(); t_phase = 0;
            kernel();
        });
    for (auto& x : th) x.join();
}
#else
// ---- the default: one OS thread per WAVEFRONT, its lanes as fibers (ucontext) that hand the thread round at every meeting — no kernel
// sleeps inside a wavefront, ~10 x the speed of a thread per lane.  A lane that returns from the kernel no longer counts at meetings and
// barriers (as on the device); a wavefront whose lanes have all returned leaves the workgroup's barrier.
struct WaveCtx;
struct Fiber {
    ucontext_t ctx;
    Dim thread, block;
    unsigned linear = 0;   // x + y * blockDim.x + ...: the lane is its low six bits
    unsigned phase = 0;
    bool done = false;
    WaveCtx* wave = nullptr;
};
struct GroupCtx { std::unique_ptr<std::barrier<>> bar; };
struct WaveCtx {
    Fiber fib[64];
    ucontext_t main_ctx;
    int n = 0, cur = 0, alive = 0;
    int arrived = 0; unsigned gen = 0;     // meetings of the wavefront
    int s_arrived = 0; unsigned sgen = 0;  // __syncthreads
    uint32_t slot[2][64];
    GroupCtx* group = nullptr;
    std::function<void()> kernel;
    char* stacks = nullptr;
};
constexpr size_t kFiberStack = 256 * 1024;
inline thread_local Fiber* t_cur = nullptr;
inline Dim& thread_idx() { return t_cur->thread; }
inline Dim& block_idx() { return t_cur->block; }
inline unsigned lane() { return t_cur->linear & 63u; }
inline uint32_t* next_slots() { return t_cur->wave->slot[t_cur->phase++ & 1u]; }
inline int next_alive(WaveCtx& w, int from)
{
    for (int k = 1; k <= w.n; k++) { const int i = (from + k) % w.n; if (!w.fib[i].done) return i; }
    return -1;
}
inline void yield()
{
    WaveCtx& w = *t_cur->wave;
    const int from = w.cur, to = next_alive(w, from);
    if (to < 0 || to == from) { std::fprintf(stderr, "wave_emu: a lane waits for lanes that have returned\n"); std::abort(); }
    w.cur = to; t_cur = &w.fib[to];
    swapcontext(&w.fib[from].ctx, &w.fib[to].ctx);
}
inline unsigned long long g_meetings = 0; // wave-level collectives executed, all wavefronts (a measure of a kernel's dependent cross-lane steps)
inline void meet()
{
    WaveCtx& w = *t_cur->wave;
    const unsigned my = w.gen;
    if (++w.arrived == w.alive) { w.arrived = 0; w.gen++; __atomic_fetch_add(&g_meetings, 1ull, __ATOMIC_RELAXED); }
    else while (w.gen == my) yield();
}
inline void group_sync()
{
    WaveCtx& w = *t_cur->wave;
    const unsigned my = w.sgen;
    if (++w.s_arrived == w.alive) { w.s_arrived = 0; w.group->bar->arrive_and_wait(); w.sgen++; }
    else while (w.sgen == my) yield();
}
inline void fiber_entry(unsigned lo, unsigned hi)
{
    WaveCtx& w = *reinterpret_cast<WaveCtx*>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    w.kernel();
    // this lane has returned: the others may have been waiting for nobody else
    Fiber& me = w.fib[w.cur];
    me.done = true;
    w.alive--;
    if (w.alive > 0) {
        if (w.arrived == w.alive) { w.arrived = 0; w.gen++; }
        if (w.s_arrived == w.alive) { w.s_arrived = 0; w.group->bar->arrive_and_wait(); w.sgen++; }
        const int to = next_alive(w, w.cur);
        w.cur = to; t_cur = &w.fib[to];
        setcontext(&w.fib[to].ctx);
    }
    setcontext(&w.main_ctx);
}
template <class K> inline void run_group(unsigned threads, unsigned block, K kernel, Dim block_dim = Dim{0, 1, 1}, Dim block_idx3 = Dim{0, 0, 0})
{
    GroupCtx g;
    const unsigned n_waves = (threads + 63) / 64;
    if (block_dim.x == 0) { block_dim.x = threads; block_idx3.x = block; }
    g.bar = std::make_unique<std::barrier<>>((std::ptrdiff_t)n_waves);
    std::vector<std::unique_ptr<WaveCtx>> waves;
    for (unsigned w = 0; w < n_waves; w++) waves.push_back(std::make_unique<WaveCtx>());
    auto run_wave = [&](unsigned wi) {
        WaveCtx& w = *waves[wi];
        w.n = w.alive = (int)std::min(64u, threads - 64u * wi);
        w.group = &g;
        w.kernel = kernel;
        w.stacks = static_cast<char*>(mmap(nullptr, kFiberStack * (size_t)w.n, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0));
        if (w.stacks == MAP_FAILED) { std::perror("wave_emu: mmap"); std::abort(); }
        for (int l = 0; l < w.n; l++) {
            Fiber& f = w.fib[l];
            f.linear = 64u * wi + (unsigned)l; f.wave = &w; f.block = block_idx3;
            f.thread.x = f.linear % block_dim.x; f.thread.y = f.linear / block_dim.x % block_dim.y; f.thread.z = f.linear / (block_dim.x * block_dim.y);
            getcontext(&f.ctx);
            f.ctx.uc_stack.ss_sp = w.stacks + kFiberStack * (size_t)l;
            f.ctx.uc_stack.ss_size = kFiberStack;
            f.ctx.uc_link = nullptr;
            const uintptr_t p = reinterpret_cast<uintptr_t>(&w);
            makecontext(&f.ctx, reinterpret_cast<void (*)()>(fiber_entry), 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
        }
        w.cur = 0; t_cur = &w.fib[0];
        swapcontext(&w.main_ctx, &w.fib[0].ctx);
        t_cur = nullptr;
        g.bar->arrive_and_drop(); // (no lane of this wavefront will come to a barrier again)
        munmap(w.stacks, kFiberStack * (size_t)w.n);
    };
    if (n_waves == 1) { run_wave(0); return; }
    std::vector<std::thread> th;
    for (unsigned wi = 0; wi < n_waves; wi++) th.emplace_back(run_wave, wi);
    for (auto& x : th) x.join();
}
#endif

// every lane publishes v and gets the value of lane src(lane) — or keeps `old` when src is negative
template <class F> inline uint32_t exchange(uint32_t v, uint32_t old, F src)
{
    uint32_t* const slot = next_slots();
    slot[lane()] = v;
    meet();
    const int s = src((int)lane());
    return s < 0 ? old : slot[s & 63];
}
} // namespace emu

#define threadIdx (emu::thread_idx())
#define blockIdx (emu::block_idx())

inline void __syncthreads() { emu::group_sync(); }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __float_as_int(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float __int_as_float(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __ffsll(long long v) { return __builtin_ffsll(v); }

inline uint32_t atomicAdd(uint32_t* p, uint32_t v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline unsigned long long atomicAdd(unsigned long long* p, unsigned long long v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
inline uint32_t atomicMin(uint32_t* p, uint32_t v)
{
    uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
inline uint32_t atomicMax(uint32_t* p, uint32_t v)
{
    uint32_t o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}

inline unsigned long long __ballot(bool p)
{
    uint32_t* const slot = emu::next_slots();
    slot[emu::lane()] = p ? 1u : 0u;
    emu::meet();
    unsigned long long m = 0;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)(slot[i] & 1u) << i;
    return m;
}
inline int __shfl(int v, int src) { return (int)emu::exchange((uint32_t)v, 0u, [&](int) { return src & 63; }); }
inline float __shfl(float v, int src) { return __int_as_float(__shfl(__float_as_int(v), src)); }
inline int __shfl_xor(int v, int m) { return (int)emu::exchange((uint32_t)v, 0u, [&](int l) { return l ^ m; }); }
inline float __shfl_xor(float v, int m) { return __int_as_float(__shfl_xor(__float_as_int(v), m)); }
inline int __builtin_amdgcn_readlane(int v, int l) { return (int)emu::exchange((uint32_t)v, 0u, [&](int) { return l; }); }
inline int __builtin_amdgcn_readfirstlane(int v) { return __builtin_amdgcn_readlane(v, 0); }
inline uint32_t __builtin_amdgcn_mbcnt_lo(uint32_t mask, uint32_t add)
{
    const unsigned l = emu::lane();
    return add + (uint32_t)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u)));
}
inline uint32_t __builtin_amdgcn_mbcnt_hi(uint32_t mask, uint32_t add)
{
    const unsigned l = emu::lane();
    return add + (l <= 32 ? 0u : (uint32_t)__builtin_popcount(mask & ((1u << (l - 32)) - 1u)));
}
// DPP: row_shr:n = 0x110 + n, row_shl:n = 0x100 + n (inside the 16-lane row; a lane without a source keeps `old`: bound_ctrl = false),
// row_bcast:15 = 0x142 (lane 15 of a row to the whole next row), row_bcast:31 = 0x143 (lane 31 to lanes 32..63)
inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl)
{
    (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    return (int)emu::exchange((uint32_t)src, (uint32_t)old, [&](int l) {
        const int row = l >> 4, in = l & 15;
        if (ctrl > 0x110 && ctrl < 0x120) { const int n = ctrl - 0x110; return in >= n ? l - n : -1; }
        if (ctrl > 0x100 && ctrl < 0x110) { const int n = ctrl - 0x100; return in + n < 16 ? l + n : -1; }
        if (ctrl == 0x142) return row >= 1 ? row * 16 - 1 : -1;
        if (ctrl == 0x143) return l >= 32 ? 31 : -1;
        std::abort();
    });
}
inline int __shfl_down(int v, int off) { return (int)emu::exchange((uint32_t)v, (uint32_t)v, [&](int l) { return l + off < 64 ? l + off : -1; }); }
inline float __shfl_down(float v, int off) { return __int_as_float(__shfl_down(__float_as_int(v), off)); }
inline uint32_t __shfl_up(uint32_t v, int off) { return emu::exchange(v, v, [&](int l) { return l - off >= 0 ? l - off : -1; }); }
inline uint32_t __shfl_down(uint32_t v, int off) { return (uint32_t)__shfl_down((int)v, off); }
inline unsigned long long __shfl_down(unsigned long long v, int off)
{
    const uint32_t lo = __shfl_down((uint32_t)v, off), hi = __shfl_down((uint32_t)(v >> 32), off);
    return ((unsigned long long)hi << 32) | lo;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v)
{
    unsigned long long o = __atomic_load_n(p, __ATOMIC_SEQ_CST);
    while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {}
    return o;
}
inline int __clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((unsigned long long)a * b) >> 32); }
inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; } // (the device's v_rcp_f32 is within 1 ulp of this; the traversal only uses it for conservative box tests)
inline unsigned long long __builtin_amdgcn_ballot_w64(bool p) { return __ballot(p); }
inline bool __builtin_amdgcn_inverse_ballot_w64(unsigned long long m) { return (m >> emu::lane()) & 1ull; }
inline void __builtin_amdgcn_s_sleep(int) { std::this_thread::yield(); }
inline unsigned long long wall_clock64() { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count() / 10ull; } // 100 MHz
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
struct alignas(8) float2 { float x, y; };
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct float3 { float x, y, z; };
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
struct alignas(8) uint2 { uint32_t x, y; };
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct alignas(16) uint4 { uint32_t x, y, z, w; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct alignas(16) int4 { int x, y, z, w; };
inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
// scoped atomics: the scope is dropped (one process), the order kept
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), (order))
#define __hip_atomic_load(p, order, scope) __uint_as_float(__atomic_load_n(reinterpret_cast<const uint32_t*>(p), (order)))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { emu::meet(); }
