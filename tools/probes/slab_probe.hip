// slab_probe.hip — cycles per 4-wide quantised slab test (the node block of traverse.h, copied) with all 64 lanes active and the node in
// registers: how close the compiler's schedule of that block gets to the 2-cycles-per-instruction issue peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ inline float bitsf(uint32_t u) { return __uint_as_float(u); }
__device__ inline uint32_t fbits(float f) { return __float_as_uint(f); }
__global__ void k(const uint4* node, float* out, int iters)
{
    const uint4 w0 = node[0], w1 = node[1], w2 = node[2], ch = node[3];
    float ox = out[threadIdx.x] * 1e-9f + 0.1f, oy = 0.2f, oz = -3.0f, t = 1e30f;
    const float ix = 1.0f / (0.3f + threadIdx.x * 1e-3f), iy = 1.0f / 0.5f, iz = 1.0f / 0.8f;
    uint32_t acc = 0;
    for (int it = 0; it < iters; it++) {
        const float Ax = bitsf(w0.w) * ix, Ay = bitsf(w2.z) * iy, Az = bitsf(w2.w) * iz;
        const float Bx = (bitsf(w0.x) - ox) * ix, By = (bitsf(w0.y) - oy) * iy, Bz = (bitsf(w0.z) - oz) * iz;
        const bool mx = ix < 0.0f, my = iy < 0.0f, mz = iz < 0.0f;
        const uint32_t nxw = mx ? w1.w : w1.x, fxw = mx ? w1.x : w1.w;
        const uint32_t nyw = my ? w2.x : w1.y, fyw = my ? w1.y : w2.x;
        const uint32_t nzw = mz ? w2.y : w1.z, fzw = mz ? w1.z : w2.y;
        const v2f Ax2 = {Ax, Ax}, Ay2 = {Ay, Ay}, Az2 = {Az, Az}, Bx2 = {Bx, Bx}, By2 = {By, By}, Bz2 = {Bz, Bz};
        uint32_t nhit = 0, first = 0;
#define SLAB(i, CH)                                                                                                   \
    {                                                                                                                 \
        const v2f qx = {(float)((nxw >> (8 * i)) & 0xffu), (float)((fxw >> (8 * i)) & 0xffu)};                        \
        const v2f qy = {(float)((nyw >> (8 * i)) & 0xffu), (float)((fyw >> (8 * i)) & 0xffu)};                        \
        const v2f qz = {(float)((nzw >> (8 * i)) & 0xffu), (float)((fzw >> (8 * i)) & 0xffu)};                        \
        const v2f tx = __builtin_elementwise_fma(qx, Ax2, Bx2), ty = __builtin_elementwise_fma(qy, Ay2, By2),         \
                  tz = __builtin_elementwise_fma(qz, Az2, Bz2);                                                       \
        const float tn = __builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), tz.x);                                          \
        const float tf = __builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y);                                          \
        const bool h = (tf >= tn) & (tn <= t) & (tf >= 0.0f) & (CH != 0xffffffffu);                                   \
        nhit += h ? 1u : 0u;                                                                                          \
        first = h ? CH : first;                                                                                       \
    }
        SLAB(0, ch.x) SLAB(1, ch.y) SLAB(2, ch.z) SLAB(3, ch.w)
        acc += nhit + first;
        ox += (float)(acc & 1u) * 1e-7f; // the next test depends on this one, like a traversal step
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc;
}
int main()
{
    uint32_t h[16] = {0, 0, 0, 0x3c000000u, 0x20100804u, 0x20100804u, 0x20100804u, 0x40302010u, 0x40302010u, 0x40302010u, 0x3c000000u, 0x3c000000u, 1, 2, 3, 0xffffffffu};
    uint4* dn; float* d;
    (void)hipMalloc(&dn, 64); (void)hipMemcpy(dn, h, 64, hipMemcpyHostToDevice);
    (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4); (void)hipMemset(d, 0, 256 * 4 * 8 * 64 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w : {1, 2, 4, 6, 8}) {
        const int blocks = 256 * 4 * w, iters = 20000;
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dn, d, 100);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, dn, d, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%d waves/SIMD: %.3f ms, %.1f cycles per slab test per SIMD (at 2.4 GHz)\n", w, ms, ms * 1e-3 * 2.4e9 / ((double)iters * w));
    }
    return 0;
}
