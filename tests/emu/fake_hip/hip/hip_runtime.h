// <hip/hip_runtime.h> for the CPU emulation of tests/emu — TEST INFRASTRUCTURE.  With tests/emu/fake_hip in front of the include path a .hip file of
// the product compiles under g++ as it stands (g++ -x c++): its kernels run under wave_emu.h (a thread per lane), and the few runtime calls its
// host side makes are these: a launch runs the workgroups of the grid one after the other; the stream is the calling thread, so the
// "asynchronous" copies and fills are done when they return.
#pragma once
#include "wave_emu.h"

#include <cstddef>
#include <cstdio>
#include <mutex>

enum hipError_t { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorInvalidDevice = 101, hipErrorNotSupported = 801, hipErrorPeerAccessAlreadyEnabled = 704 };
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
inline std::mutex g_launch_mutex; // one kernel at a time: "LDS" is a static array per kernel, whichever host thread launches
template <class K, class... A> inline void launch(K kernel, dim3 grid, dim3 block, A... args)
{
    std::lock_guard<std::mutex> one_at_a_time(g_launch_mutex);
    const Dim gd{grid.x, grid.y, grid.z}, bd{block.x, block.y, block.z};
    for (unsigned bz = 0; bz < grid.z; bz++)
        for (unsigned by = 0; by < grid.y; by++)
            for (unsigned bx = 0; bx < grid.x; bx++) {
                run_group(block.x * block.y * block.z, bx, [&] { t_grid = gd; t_dim = bd; kernel(args...); }, bd, Dim{bx, by, bz});
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
// ---- the rest of the runtime the library's host side uses: ONE device whose memory is the host's, streams that are the calling thread
// (a call is done when it returns), events that are timestamps, no peers, no IPC
#include <chrono>
#include <cstdlib>
struct ihipEvent_t { std::chrono::steady_clock::time_point t; };
typedef ihipEvent_t* hipEvent_t;
struct ihipStream_t { int unused; };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocMapped = 2, hipHostRegisterDefault = 0, hipDeviceMallocFinegrained = 1,
       hipDeviceMallocUncached = 3, hipIpcMemLazyEnablePeerAccess = 1 };
enum hipMemoryType { hipMemoryTypeHost = 0, hipMemoryTypeDevice = 1, hipMemoryTypeUnregistered = 3 };
struct hipPointerAttribute_t { hipMemoryType type; int device; void* devicePointer; void* hostPointer; };
struct hipDeviceProp_t { char name[256]; size_t totalGlobalMem; int multiProcessorCount; char gcnArchName[256]; };
struct hipIpcMemHandle_t { char reserved[64]; };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "an error of the emulated runtime"; }
inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidDevice; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::memset(p, 0, sizeof(*p)); std::strcpy(p->name, "wave_emu"); std::strcpy(p->gcnArchName, "x86"); p->totalGlobalMem = (size_t)8 << 30; p->multiProcessorCount = 8; return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 0; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipErrorNotSupported; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = static_cast<T*>(std::aligned_alloc(256, (n + 255) / 256 * 256 + 256)); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipExtMallocWithFlags(T** p, size_t n, unsigned) { return hipMalloc(p, n); }
template <class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
inline hipError_t hipHostUnregister(void*) { return hipSuccess; }
template <class T> inline hipError_t hipHostGetDevicePointer(T** d, void* h, unsigned) { *d = static_cast<T*>(h); return hipSuccess; }
inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) { a->type = hipMemoryTypeUnregistered; a->device = 0; a->devicePointer = nullptr; a->hostPointer = const_cast<void*>(p); return hipErrorInvalidValue; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* p, int v, size_t n) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = new ihipStream_t{0}; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new ihipEvent_t{std::chrono::steady_clock::now()}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t*, void*) { return hipErrorNotSupported; }
inline hipError_t hipIpcOpenMemHandle(void**, hipIpcMemHandle_t, unsigned) { return hipErrorNotSupported; }
inline hipError_t hipIpcCloseMemHandle(void*) { return hipErrorNotSupported; }
inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { std::memset(p, v, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memmove(d, s, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipGetLastError() { return hipSuccess; }
