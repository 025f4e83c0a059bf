// <hip/hip_runtime.h> for the CPU emulation of tests/emu — TEST INFRASTRUCTURE.  With tests/emu/fake_hip in front of the include path a .hip file of
// the product compiles under g++ as it stands (g++ -x c++): its kernels run under wave_emu.h (a thread per lane), and the few runtime calls its
// host side makes are these: a launch runs the workgroups of the grid one after the other; the stream is the calling thread, so the
// "asynchronous" copies and fills are done when they return.
#pragma once
#include "wave_emu.h"

#include <cstddef>
#include <cstdio>

enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct ihipStream_t;
typedef ihipStream_t* hipStream_t;
struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
namespace emu {
inline thread_local Dim t_grid, t_dim;
inline unsigned long long g_groups_run = 0;
template <class K, class... A> inline void launch(K kernel, dim3 grid, dim3 block, A... args)
{
    for (unsigned b = 0; b < grid.x; b++) {
        run_group(block.x, b, [&] { t_grid.x = grid.x; t_dim.x = block.x; kernel(args...); });
        g_groups_run++;
    }
}
} // namespace emu
#define gridDim (emu::t_grid)
#define blockDim (emu::t_dim)
#define HIP_KERNEL_NAME(...) __VA_ARGS__
#define hipLaunchKernelGGL(kernel, grid, block, shared_bytes, stream, ...) emu::launch(kernel, grid, block, __VA_ARGS__)
inline uint32_t min(uint32_t a, uint32_t b) { return a < b ? a : b; }
inline uint32_t max(uint32_t a, uint32_t b) { return a > b ? a : b; }
inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
