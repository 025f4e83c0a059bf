// Probe (planning numbers, not a test): how fast does a 185 MB host -> device copy run from hipHostMalloc'ed / hipHostRegister'ed memory,
// alone and while kernels keep the device busy on another stream, right after CPU threads rewrote the source?
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 tools/probes/h2d_overlap.cpp -o /tmp/h2d -pthread && /tmp/h2d
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_busy(float4* a, const float4* b, size_t n, int rounds)
{
    for (int r = 0; r < rounds; r++)
        for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = b[i]; v.x += 1.0f; a[i] = v; }
}
static void rewrite(char* p, const char* src, size_t bytes, int nt)
{
    std::vector<std::thread> pool;
    const size_t per = (bytes + nt - 1) / nt;
    for (int k = 0; k < nt; k++) pool.emplace_back([=] { const size_t a = std::min(bytes, per * k), b = std::min(bytes, a + per); if (a < b) memcpy(p + a, src + a, b - a); });
    for (auto& t : pool) t.join();
}
int main()
{
    const size_t bytes = (size_t)1048568 * 176;
    char* src = (char*)malloc(bytes); memset(src, 1, bytes);
    char *pinned = nullptr, *reg = (char*)aligned_alloc(4096, (bytes + 4095) & ~(size_t)4095);
    CK(hipHostMalloc((void**)&pinned, bytes, hipHostMallocDefault));
    memset(reg, 0, bytes);
    CK(hipHostRegister(reg, bytes, hipHostRegisterDefault));
    char* dev; CK(hipMalloc((void**)&dev, bytes));
    float4 *ba, *bb; const size_t nb = (size_t)64 << 20; CK(hipMalloc((void**)&ba, nb * 16)); CK(hipMalloc((void**)&bb, nb * 16));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const char* kind : {"hipHostMalloc", "hipHostRegister"}) {
        char* h = kind[7] == 'M' ? pinned : reg;
        for (int busy = 0; busy < 2; busy++)
            for (int fresh = 0; fresh < 2; fresh++)
                for (int rep = 0; rep < 3; rep++) {
                    if (fresh) rewrite(h, src, bytes, 16);
                    if (busy) hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, s2, ba, bb, nb, 4);
                    CK(hipEventRecord(e0, s1));
                    CK(hipMemcpyAsync(dev, h, bytes, hipMemcpyHostToDevice, s1));
                    CK(hipEventRecord(e1, s1));
                    CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
                    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("%-16s device %s source %s: %.3f ms = %.1f GB/s\n", kind, busy ? "busy" : "idle", fresh ? "just rewritten by 16 threads" : "untouched", ms, bytes / ms / 1e6);
                }
    }
    return 0;
}
