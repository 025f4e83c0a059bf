// valu_issue_probe.hip — what a wave64 vector instruction costs on gfx950, as a function of instruction-level parallelism and of the
// number of active lanes.  Build and run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o probe valu_issue_probe.hip
// Measured on MI355X (2.4 GHz, 1024 SIMDs), G wave-instructions/s:
//   independent FMA chains (ILP 2 / 4):            880-1000   (2.45-2.8 cycles per instruction and SIMD: FP32 runs 32 lanes per clock,
//                                                              the 157 TFLOP/s vector peak = 1229 G wave-instructions/s)
//   every instruction depends on the one before:   517        (4.75 cycles)
//   the same with <= 16 active lanes:              119        (20.6 cycles!  17 lanes and more: 530; other waves do not hide it)
//   Moeller-Trumbore-like code, 64 / 17 / 16 / 11 active lanes: 17.4 / 19.3 / 19.7 / 19.8 G triangles/s (no penalty in real code)
// DESIGN.md section 5 uses these numbers for the instruction-issue roofline of the trace kernels.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP> __global__ void k(float* out, unsigned long long mask, int iters)
{
    const int lane = threadIdx.x & 63;
    const bool active = (mask >> lane) & 1ull;
    float a[ILP];
    for (int j = 0; j < ILP; j++) a[j] = out[threadIdx.x + j];
    const float b = 1.0001f, c = 0.5f;
    if (active) {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int r = 0; r < 12 / ILP; r++)
#pragma unroll
                for (int j = 0; j < ILP; j++) a[j] = __builtin_fmaf(a[j], b, c);
        }
    }
    float s = 0;
    for (int j = 0; j < ILP; j++) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// Moeller-Trumbore-like dependent arithmetic (cross, dot, compare, select), one "triangle" per iteration
__global__ void kmt(float* out, unsigned long long mask, int iters)
{
    const int lane = threadIdx.x & 63;
    const bool active = (mask >> lane) & 1ull;
    float ox = out[threadIdx.x], oy = ox + 1, oz = ox + 2, dx = 0.3f, dy = 0.5f, dz = 0.8f, acc = 0;
    if (active) {
        for (int i = 0; i < iters; i++) {
            const float e1x = 1 + acc * 1e-9f, e1y = 0.1f, e1z = 0.2f, e2x = 0.1f, e2y = 1, e2z = 0.3f, v0x = 0.5f, v0y = 0.25f, v0z = 3 + acc * 1e-9f;
            const float hx = dy * e2z - dz * e2y, hy = dz * e2x - dx * e2z, hz = dx * e2y - dy * e2x;
            const float a = e1x * hx + e1y * hy + e1z * hz;
            const float f = __builtin_amdgcn_rcpf(a);
            const float sx = ox - v0x, sy = oy - v0y, sz = oz - v0z;
            const float u = f * (sx * hx + sy * hy + sz * hz);
            const float qx = sy * e1z - sz * e1y, qy = sz * e1x - sx * e1z, qz = sx * e1y - sy * e1x;
            const float v = f * (dx * qx + dy * qy + dz * qz);
            const float t = f * (e2x * qx + e2y * qy + e2z * qz);
            acc += (u >= 0 && v >= 0 && u + v <= 1) ? t : 0.25f;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <typename K> void run(K kern, float* d, const char* name, unsigned long long m, double per_iter)
{
    const int blocks = 256 * 4 * 6, threads = 64, iters = 10000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, m, 100);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d, m, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.3f ms  (%.1f G iter-units/s)\n", name, ms, (double)blocks * iters * per_iter / ms / 1e6);
}
int main()
{
    float* d;
    (void)hipMalloc(&d, 256 * 4 * 16 * 64 * sizeof(float) + 64);
    (void)hipMemset(d, 0, 256 * 4 * 16 * 64 * sizeof(float) + 64);
    const unsigned long long m64 = ~0ull, m17 = 0x1ffffull, m16 = 0xffffull, m11 = 0x7ffull;
    run(k<1>, d, "ILP1 64", m64, 12); run(k<1>, d, "ILP1 17", m17, 12); run(k<1>, d, "ILP1 16", m16, 12);
    run(k<2>, d, "ILP2 64", m64, 12); run(k<2>, d, "ILP2 17", m17, 12); run(k<2>, d, "ILP2 16", m16, 12);
    run(k<3>, d, "ILP3 64", m64, 12); run(k<3>, d, "ILP3 17", m17, 12); run(k<3>, d, "ILP3 16", m16, 12);
    run(k<4>, d, "ILP4 64", m64, 12); run(k<4>, d, "ILP4 16", m16, 12);
    run(kmt, d, "MT-like 64", m64, 1); run(kmt, d, "MT-like 17", m17, 1); run(kmt, d, "MT-like 16", m16, 1); run(kmt, d, "MT-like 11", m11, 1);
    return 0;
}
