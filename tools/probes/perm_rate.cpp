// Probe (round 5 planning): do v_perm_b32 / v_and_or_b32 issue at the full rate?  (If they do, the per-lane node test's 24 half-rate
// v_cvt_f32_ubyte per visit could become 24 full-rate byte permutes building 0x4B0000qq = 2^23 + q, the 2^23 folded into the plane offset.)
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 tools/probes/perm_rate.cpp -o /tmp/perm_rate && /tmp/perm_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int kIters = 4000;
#define B8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define B32(I) B8(I) B8(I) B8(I) B8(I)
template <int KIND> __global__ __launch_bounds__(256) void k(float* out, float seed)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + (float)(threadIdx.x + i);
    const float a = seed * 1.0001f, b = seed * 0.5f;
    const uint32_t u = __float_as_uint(seed) | 0x01020304u, magic = 0x4B000000u, sel = 0x07060501u;
    for (int it = 0; it < kIters; it++) {
        if (KIND == 0) {
#define I(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            B32(I)
#undef I
        } else if (KIND == 1) {
#define I(k) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[k]) : "v"(u));
            B32(I)
#undef I
        } else if (KIND == 2) {
#define I(k) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v[k]) : "v"(magic), "v"(u), "v"(sel));
            B32(I)
#undef I
        } else if (KIND == 3) {
#define I(k) asm volatile("v_and_or_b32 %0, %1, %3, %2" : "=v"(v[k]) : "v"(u), "v"(magic), "v"(sel));
            B32(I)
#undef I
        } else if (KIND == 4) {
#define I(k) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(v[k]) : "v"(magic), "v"(u), "s"(sel));
            B32(I)
#undef I
        } else if (KIND == 5) { // the proposed child: 6 permutes + 6 fma + max3 + min3 + min + max + cmp
            asm volatile("v_perm_b32 %0, %6, %7, %8\n\tv_perm_b32 %1, %6, %7, %8\n\tv_perm_b32 %2, %6, %7, %8\n\tv_perm_b32 %3, %6, %7, %8\n\tv_perm_b32 %4, %6, %7, %8\n\tv_perm_b32 %5, %6, %7, %8"
                         : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]) : "v"(magic), "v"(u), "v"(sel));
            asm volatile("v_fma_f32 %0, %0, %6, %7\n\tv_fma_f32 %1, %1, %6, %7\n\tv_fma_f32 %2, %2, %6, %7\n\tv_fma_f32 %3, %3, %6, %7\n\tv_fma_f32 %4, %4, %6, %7\n\tv_fma_f32 %5, %5, %6, %7"
                         : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]) : "v"(a), "v"(b));
            asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %5, %6, %7" : "=v"(v[6]), "=v"(v[7]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]));
            asm volatile("v_min_f32 %0, %0, %2\n\tv_max_f32 %1, %1, 0\n\tv_cmp_ge_f32 vcc, %0, %1" : "+v"(v[7]), "+v"(v[6]) : "v"(a) : "vcc");
        } else { // today's child: 6 cvt + 3 pk_fma + max3 + min3 + min + max + cmp
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p0 = {v[0], v[1]}, p1 = {v[2], v[3]}, p2 = {v[4], v[5]};
            const v2 pa = {a, a}, pb = {b, b};
            asm volatile("v_cvt_f32_ubyte0 %0, %6\n\tv_cvt_f32_ubyte1 %1, %6\n\tv_cvt_f32_ubyte2 %2, %6\n\tv_cvt_f32_ubyte3 %3, %6\n\tv_cvt_f32_ubyte0 %4, %6\n\tv_cvt_f32_ubyte1 %5, %6"
                         : "=v"(p0.x), "=v"(p0.y), "=v"(p1.x), "=v"(p1.y), "=v"(p2.x), "=v"(p2.y) : "v"(u));
            asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n\tv_pk_fma_f32 %1, %1, %3, %4\n\tv_pk_fma_f32 %2, %2, %3, %4" : "+v"(p0), "+v"(p1), "+v"(p2) : "v"(pa), "v"(pb));
            asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %5, %6, %7" : "=v"(v[6]), "=v"(v[7]) : "v"(p0.x), "v"(p1.x), "v"(p2.x), "v"(p0.y), "v"(p1.y), "v"(p2.y));
            asm volatile("v_min_f32 %0, %0, %2\n\tv_max_f32 %1, %1, 0\n\tv_cmp_ge_f32 vcc, %0, %1" : "+v"(v[7]), "+v"(v[6]) : "v"(a) : "vcc");
            v[0] = p0.x; v[1] = p0.y; v[2] = p1.x; v[3] = p1.y; v[4] = p2.x; v[5] = p2.y;
        }
    }
    float r = 0; for (int i = 0; i < 8; i++) r += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND> int run(const char* name, double per_iter, int cus)
{
    float* out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<KIND>, dim3(cus * 8), dim3(256), 0, 0, out, 1.5f);
    CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<KIND>, dim3(cus * 8), dim3(256), 0, 0, out, 1.5f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double insts = (double)cus * 8 * 4 * kIters * per_iter;
    printf("%-44s %8.1f G wave64 instructions/s   (%.1f children/ns-chip)\n", name, insts / (ms * 1e-3) / 1e9, per_iter < 32 ? insts / per_iter / (ms * 1e-3) / 1e9 : 0.0);
    CK(hipFree(out)); return 0;
}
int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); const int cus = p.multiProcessorCount;
    run<0>("v_fma_f32", 32, cus); run<1>("v_cvt_f32_ubyte1", 32, cus); run<2>("v_perm_b32 (vgpr selector)", 32, cus); run<4>("v_perm_b32 (sgpr selector)", 32, cus);
    run<3>("v_and_or_b32", 32, cus); run<5>("child: 6 perm + 6 fma + 5 (17 instr)", 17, cus); run<6>("child today: 6 cvt + 3 pk_fma + 5 (14 instr)", 14, cus);
    return 0;
}
