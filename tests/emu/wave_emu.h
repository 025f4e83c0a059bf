// wave_emu.h — TEST INFRASTRUCTURE: a lockstep-free emulator of one gfx950 workgroup on CPU threads, for running the DEVICE SOURCE of a
// wave-synchronous kernel (the text of rfw-rs_amd/csrc/sah_build.hip's phase 2, cut out by tests/test_builder_emulated.py) under g++.
//
// One OS thread per lane.  Everything that crosses lanes on the GPU is a collective here — the lanes of a wavefront meet at a barrier, publish
// their operand, read their source lane's: __ballot, __shfl, __shfl_xor, readlane / readfirstlane, the DPP row operations
// (update_dpp with row_shr / row_shl / row_bcast:15 / row_bcast:31, the controls the builder uses), and the wave barrier of wave_sync().
// Between collectives the lanes run free, so an LDS hand-over between lanes that the kernel forgot to fence with wave_sync() (on the GPU: left
// to instruction order and the compiler's mercy) shows up here as a race.  LDS is a function-local static (one workgroup at a time), LDS and
// global atomics are __atomic builtins, __syncthreads is a barrier over the workgroup.
// Limits: every lane of a wavefront must reach every collective (true for the builder: its wave-level branches are uniform); a kernel is
// run for one blockIdx at a time.
#pragma once
#include <barrier>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <thread>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __shared__ static
#define __launch_bounds__(...)

namespace emu {
struct Dim { unsigned x = 0, y = 0, z = 0; };
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
inline unsigned lane() { return t_thread.x & 63u; }

// every lane publishes v and gets the value of lane src(lane) — or keeps `old` when src is negative
template <class F> inline uint32_t exchange(uint32_t v, uint32_t old, F src)
{
    WaveCtx& w = *t_wave;
    uint32_t* const slot = w.slot[t_phase++ & 1u];
    slot[lane()] = v;
    w.bar.arrive_and_wait();
    const int s = src((int)lane());
    return s < 0 ? old : slot[s & 63];
}

// run kernel(args...) for one workgroup of `threads` threads with blockIdx.x = block
template <class K> inline void run_group(unsigned threads, unsigned block, K kernel)
{
    GroupCtx g;
    g.bar = std::make_unique<std::barrier<>>((std::ptrdiff_t)threads);
    for (unsigned w = 0; w < (threads + 63) / 64; w++) g.waves.push_back(std::make_unique<WaveCtx>());
    std::vector<std::thread> th;
    for (unsigned t = 0; t < threads; t++)
        th.emplace_back([&, t] {
            t_thread.x = t; t_block.x = block; t_group = &g; t_wave = g.waves[t / 64].get(); t_phase = 0;
            kernel();
        });
    for (auto& x : th) x.join();
}
} // namespace emu

#define threadIdx (emu::t_thread)
#define blockIdx (emu::t_block)

inline void __syncthreads() { emu::t_group->bar->arrive_and_wait(); }
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
    emu::WaveCtx& w = *emu::t_wave;
    uint32_t* const slot = w.slot[emu::t_phase++ & 1u];
    slot[emu::lane()] = p ? 1u : 0u;
    w.bar.arrive_and_wait();
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
inline int __clz(uint32_t v) { return v ? __builtin_clz(v) : 32; }
struct alignas(16) float4 { float x, y, z, w; };
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
// scoped atomics: the scope is dropped (one process), the order kept
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
#define __hip_atomic_fetch_add(p, v, order, scope) __atomic_fetch_add((p), (v), (order))
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), (order))
#define __hip_atomic_load(p, order, scope) __uint_as_float(__atomic_load_n(reinterpret_cast<const uint32_t*>(p), (order)))
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_fence(...) ((void)0)
inline void __builtin_amdgcn_wave_barrier() { emu::t_wave->bar.arrive_and_wait(); }
