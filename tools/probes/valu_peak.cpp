// Probe (planning / roofline numbers, not a test): what does one SIMD of gfx950 actually issue per cycle for the instruction kinds the trace
// and shade kernels are made of?  bench.py prices vector issue at 1 wave64 instruction per 2 cycles per SIMD (the FP32 FMA peak).  This
// measures, with 1 / 2 / 4 / 8 resident wavefronts per SIMD, shader cycles (s_memtime) per instruction for streams of independent
// instructions of one kind, and chip-wide G instructions / s from HIP events.
// build + run on the GPU box: hipcc -O2 --offload-arch=gfx950 tools/probes/valu_peak.cpp -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int kUnroll = 32, kIters = 2000;
// 8 independent accumulators, kUnroll instructions per trip
#define BODY8(INSTR) INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7)
#define BODY32(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR) BODY8(INSTR)

template <int KIND> __global__ __launch_bounds__(256) void k_stream(float* out, uint64_t* cycles, float seed)
{
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + (float)(threadIdx.x + i);
    float a = seed * 1.0001f, b = seed * 0.5f;
    uint32_t u = __float_as_uint(seed) | 1u;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < kIters; it++) {
        if (KIND == 0) {
#define I(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            BODY32(I)
#undef I
        } else if (KIND == 1) {
#define I(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            BODY32(I)
#undef I
        } else if (KIND == 2) {
#define I(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a) : );
            BODY32(I)
#undef I
        } else if (KIND == 3) {
#define I(k) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 4) {
#define I(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
            BODY32(I)
#undef I
        } else if (KIND == 5) {
#define I(k) asm volatile("v_cmp_ge_f32 vcc, %0, %1" : : "v"(v[k]), "v"(a) : "vcc");
            BODY32(I)
#undef I
        } else if (KIND == 6) {
#define I(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 7) {
#define I(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            BODY32(I)
#undef I
        } else if (KIND == 8) { // packed: two FMAs per lane per instruction
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p[4] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}, {v[6], v[7]}};
            const v2 pa = {a, a}, pb = {b, b};
#define I(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k & 3]) : "v"(pa), "v"(pb));
            BODY32(I)
#undef I
            v[0] = p[0].x + p[1].y + p[2].x + p[3].y;
        } else if (KIND == 9) {
#define I(k) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[k]));
            BODY32(I)
#undef I
        } else if (KIND == 10) { // scalar ALU
            uint32_t s = u;
#define I(k) asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) : : "scc");
            BODY32(I)
#undef I
            u = s;
        } else if (KIND == 11) { // a node-test like mix: cvt, fma, max3/min3, cmp, cndmask
#define I(k) asm volatile("v_cvt_f32_ubyte0 %0, %3\n\tv_fma_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %1, %2\n\tv_cmp_ge_f32 vcc, %0, %1" : "+v"(v[k]) : "v"(a), "v"(b), "v"(u) : "vcc");
            BODY8(I)
#undef I
        } else if (KIND == 13) { // select with the mask in an SGPR pair set before the loop
            const uint64_t m = 0x5555aaaa5555aaaaull ^ (uint64_t)u;
#define I(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "s"(m));
            BODY32(I)
#undef I
        } else if (KIND == 14) { // compare + select pairs (the usual form)
#define I(k) asm volatile("v_cmp_ge_f32 vcc, %0, %2\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a), "v"(b) : "vcc");
            BODY8(I) BODY8(I)
#undef I
        } else if (KIND == 15) { // select, vcc initialised by a scalar move before the loop, distinct source and destination
            asm volatile("s_mov_b64 vcc, 0x55" : : : "vcc");
            float w[8];
#define I(k) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(w[k]) : "v"(v[k]), "v"(a));
            BODY32(I)
#undef I
            v[0] += w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        } else if (KIND == 16) {
#define I(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[k]) : "v"(a));
            BODY32(I)
#undef I
        } else if (KIND == 17) {
#define I(k) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 18) { // SDWA byte select + convert in one instruction
#define I(k) asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1" : "=v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 40) { // byte -> float WITHOUT a conversion: 0x4B000000 | byte = 2^23 + byte, as a VOP2 or with an SDWA byte select (full rate?)
            const uint32_t magic = 0x4B000000u;
#define I(k) asm volatile("v_or_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1" : "=v"(v[k]) : "v"(magic), "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 41) { // the same as one child's planes: 6 or_sdwa + 6 plain FMAs + max3/min3 ... (14 instructions like KIND 38, conversions exchanged)
            const uint32_t magic = 0x4B000000u;
            float q[6];
#define I(k) asm volatile("v_or_b32_sdwa %0, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_or_b32_sdwa %1, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
                          "v_or_b32_sdwa %2, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n\tv_or_b32_sdwa %3, %8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n\t" \
                          "v_or_b32_sdwa %4, %8, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\tv_or_b32_sdwa %5, %8, %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n\t" \
                          "v_fma_f32 %0, %0, %10, %11\n\tv_fma_f32 %1, %1, %10, %11\n\tv_fma_f32 %2, %2, %10, %11\n\tv_fma_f32 %3, %3, %10, %11\n\tv_fma_f32 %4, %4, %10, %11\n\tv_fma_f32 %5, %5, %10, %11\n\t" \
                          "v_max_f32 %6, %0, %2\n\tv_min_f32 %7, %1, %3\n\tv_min3_f32 %7, %7, %5, %11\n\tv_max3_f32 %6, %6, %4, 0\n\tv_cmp_ge_f32 vcc, %7, %6" \
                          : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(magic), "v"(u), "v"(a), "v"(b), "v"(u ^ 0x3c003c00u) : "vcc");
            I(0) I(1)
#undef I
            v[0] += q[0];
        } else if (KIND == 19) {
#define I(k) asm volatile("v_lshrrev_b32 %0, 8, %1" : "=v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 20) {
#define I(k) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 21) { // ballot-like: compare into an SGPR pair
            uint64_t m;
#define I(k) asm volatile("v_cmp_ge_f32_e64 %0, %1, %2" : "=s"(m) : "v"(v[k]), "v"(a));
            BODY32(I)
#undef I
            u += (uint32_t)m;
        } else if (KIND == 22) {
#define I(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 23) {
#define I(k) asm volatile("v_readlane_b32 %0, %1, 5" : "=s"(u) : "v"(v[k]));
            BODY32(I)
#undef I
        } else if (KIND == 25) { // a mask combined on the scalar unit into an SGPR pair, then a VOP3 select on it
            const uint64_t m1 = 0x5555aaaa5555aaaaull ^ (uint64_t)u, m2 = 0x0f0f0f0ff0f0f0f0ull | (uint64_t)u;
            uint64_t m3;
#define I(k) asm volatile("s_and_b64 %1, %3, %4\n\tv_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(v[k]), "=&s"(m3) : "v"(a), "s"(m1), "s"(m2) : "scc");
            BODY8(I) BODY8(I)
#undef I
        } else if (KIND == 26) { // compare into vcc, select on vcc 4 instructions later (other VALU work in between)
#define I(k) asm volatile("v_cmp_ge_f32 vcc, %0, %2\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[k]) : "v"(a), "v"(b) : "vcc");
            BODY8(I)
#undef I
        } else if (KIND == 27) { // 64-bit address arithmetic of the per-lane node fetch
            uint64_t q[4] = {u, u + 1u, u + 2u, u + 3u};
#define I(k) asm volatile("v_lshl_add_u64 %0, %0, 6, %1" : "+v"(q[k & 3]) : "v"(q[(k + 1) & 3]));
            BODY32(I)
#undef I
            u += (uint32_t)(q[0] + q[1] + q[2] + q[3]);
        } else if (KIND == 28) {
            uint64_t q[4] = {u, u + 1u, u + 2u, u + 3u};
#define I(k) asm volatile("v_lshlrev_b64 %0, 6, %0" : "+v"(q[k & 3]));
            BODY32(I)
#undef I
            u += (uint32_t)(q[0] + q[1] + q[2] + q[3]);
        } else if (KIND == 29) {
            uint64_t q[4] = {u, u + 1u, u + 2u, u + 3u};
#define I(k) asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(q[k & 3]) : "v"(u) : "vcc");
            BODY32(I)
#undef I
            u += (uint32_t)(q[0] + q[1] + q[2] + q[3]);
        } else if (KIND == 30) {
            uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = u + i;
#define I(k) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(w[k]) : "v"(u) : "vcc");
            BODY32(I)
#undef I
            u += w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        } else if (KIND == 31) {
            uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = u + i;
#define I(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(w[k]) : "v"(u));
            BODY32(I)
#undef I
            u += w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        } else if (KIND == 32) {
            uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = u + i;
#define I(k) asm volatile("v_mov_b32 %0, %1" : "=v"(w[k]) : "v"(u));
            BODY32(I)
#undef I
            u += w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        } else if (KIND == 33) {
            uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = u + i;
#define I(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(w[k]) : "v"(u));
            BODY32(I)
#undef I
            u += w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        } else if (KIND == 34) {
            uint32_t w[8]; for (int i = 0; i < 8; i++) w[i] = u + i;
#define I(k) asm volatile("v_lshl_add_u32 %0, %0, 6, %1" : "+v"(w[k]) : "v"(u));
            BODY32(I)
#undef I
            u += w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7];
        } else if (KIND == 35) { // mixed-precision FMA: source 0 an f16 half of a register, converted on the fly
#define I(k) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(v[k]) : "v"(u), "v"(a));
            BODY32(I)
#undef I
        } else if (KIND == 36) { // ... the high half
#define I(k) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(v[k]) : "v"(u), "v"(a));
            BODY32(I)
#undef I
        } else if (KIND == 37) {
#define I(k) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(v[k]) : "v"(u));
            BODY32(I)
#undef I
        } else if (KIND == 38) { // one child of the per-lane node test as it is: 6 cvt, 3 pk_fma, max, min, min3, max3, cmp (14 instructions)
            typedef float v2 __attribute__((ext_vector_type(2)));
            v2 p[3]; const v2 pa = {a, a}, pb = {b, b};
            for (int r = 0; r < 2; r++) {
                asm volatile("v_cvt_f32_ubyte0 %0, %6\n\tv_cvt_f32_ubyte1 %1, %6\n\tv_cvt_f32_ubyte2 %2, %6\n\tv_cvt_f32_ubyte3 %3, %6\n\tv_cvt_f32_ubyte0 %4, %6\n\tv_cvt_f32_ubyte1 %5, %6"
                             : "=v"(p[0].x), "=v"(p[0].y), "=v"(p[1].x), "=v"(p[1].y), "=v"(p[2].x), "=v"(p[2].y) : "v"(u));
                asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n\tv_pk_fma_f32 %1, %1, %3, %4\n\tv_pk_fma_f32 %2, %2, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]) : "v"(pa), "v"(pb));
                asm volatile("v_max_f32 %0, %2, %4\n\tv_min_f32 %1, %3, %5\n\tv_min3_f32 %1, %1, %7, %8\n\tv_max3_f32 %0, %0, %6, 0\n\tv_cmp_ge_f32 vcc, %1, %0"
                             : "=&v"(v[6]), "=&v"(v[7]) : "v"(p[0].x), "v"(p[0].y), "v"(p[1].x), "v"(p[1].y), "v"(p[2].x), "v"(p[2].y), "v"(a) : "vcc");
            }
            v[0] += p[0].x;
        } else if (KIND == 39) { // ... with the planes as f16 halves read by v_fma_mix_f32: 6 fma_mix, max, min, min3, max3, cmp (11 instructions)
            float q[6];
#define I(k) asm volatile("v_fma_mix_f32 %0, %8, %9, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %8, %9, %10 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                          "v_fma_mix_f32 %2, %11, %9, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %3, %11, %9, %10 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                          "v_fma_mix_f32 %4, %8, %10, %9 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %5, %11, %10, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                          "v_max_f32 %6, %0, %2\n\tv_min_f32 %7, %1, %3\n\tv_min3_f32 %7, %7, %5, %9\n\tv_max3_f32 %6, %6, %4, 0\n\tv_cmp_ge_f32 vcc, %7, %6" \
                          : "=&v"(q[0]), "=&v"(q[1]), "=&v"(q[2]), "=&v"(q[3]), "=&v"(q[4]), "=&v"(q[5]), "=&v"(v[6]), "=&v"(v[7]) : "v"(u), "v"(a), "v"(b), "v"(u ^ 0x3c003c00u) : "vcc");
            I(0) I(1)
#undef I
            v[0] += q[0];
        } else if (KIND == 12) { // VALU and SALU interleaved 1 : 1 (do they share an issue slot?)
            uint32_t s = u;
#define I(k) asm volatile("v_fma_f32 %0, %0, %2, %3\n\ts_add_u32 %1, %1, 3" : "+v"(v[k]), "+s"(s) : "v"(a), "v"(b) : "scc");
            BODY8(I) BODY8(I)
#undef I
            u = s;
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    float r = 0.0f;
    for (int i = 0; i < 8; i++) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + __uint_as_float(u);
    if ((threadIdx.x & 63u) == 0) cycles[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int KIND> int run(const char* name, int per_trip, float* d_out, uint64_t* d_cyc, hipEvent_t e0, hipEvent_t e1)
{
    for (int waves_per_simd : {1, 4, 8}) {
        const int blocks = 256 * waves_per_simd; // 256 CUs x (waves_per_simd x 4 SIMDs) wavefronts = one 256-thread block per CU per wave slot
        hipLaunchKernelGGL(k_stream<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.5f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_stream<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 1.5f);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<uint64_t> cyc((size_t)blocks * 4);
        CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        double mean = 0; for (auto c : cyc) mean += (double)c; mean /= (double)cyc.size();
        const double instr_per_wave = (double)kIters * per_trip;
        const double total = instr_per_wave * blocks * 4;
        printf("%-34s %d waves/SIMD: %6.2f s_memtime ticks per instruction per wave, %6.2f per instruction per SIMD; chip %8.1f G wave-instructions/s (%.3f ms)\n", name, waves_per_simd,
               mean / instr_per_wave, mean / instr_per_wave / waves_per_simd, total / ms / 1e6, ms);
    }
    return 0;
}

int main()
{
    float* d_out; uint64_t* d_cyc;
    CK(hipMalloc((void**)&d_out, (size_t)2048 * 256 * 4));
    CK(hipMalloc((void**)&d_cyc, (size_t)2048 * 4 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int wall_clock_rate = 0; CK(hipDeviceGetAttribute(&wall_clock_rate, hipDeviceAttributeWallClockRate, 0));
    int clk = 0; CK(hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0));
    printf("clock rate attribute %d kHz, wall clock rate %d kHz\n", clk, wall_clock_rate);
    run<0>("v_fma_f32", 32, d_out, d_cyc, e0, e1);
    run<1>("v_add_f32", 32, d_out, d_cyc, e0, e1);
    run<7>("v_mul_f32", 32, d_out, d_cyc, e0, e1);
    run<2>("v_cndmask_b32", 32, d_out, d_cyc, e0, e1);
    run<3>("v_cvt_f32_ubyte1", 32, d_out, d_cyc, e0, e1);
    run<4>("v_max3_f32", 32, d_out, d_cyc, e0, e1);
    run<5>("v_cmp_ge_f32", 32, d_out, d_cyc, e0, e1);
    run<6>("v_and_b32", 32, d_out, d_cyc, e0, e1);
    run<8>("v_pk_fma_f32", 32, d_out, d_cyc, e0, e1);
    run<9>("v_rcp_f32", 32, d_out, d_cyc, e0, e1);
    run<13>("v_cndmask_b32_e64 (sgpr mask)", 32, d_out, d_cyc, e0, e1);
    run<15>("v_cndmask_b32 (vcc set, dst!=src)", 32, d_out, d_cyc, e0, e1);
    run<14>("v_cmp + v_cndmask (x16 pairs)", 32, d_out, d_cyc, e0, e1);
    run<25>("s_and_b64 sN + v_cndmask_e64 sN (x16)", 32, d_out, d_cyc, e0, e1);
    run<26>("v_cmp vcc, add, mul, v_cndmask vcc (x8)", 32, d_out, d_cyc, e0, e1);
    run<16>("v_min_f32", 32, d_out, d_cyc, e0, e1);
    run<17>("v_cvt_f32_u32", 32, d_out, d_cyc, e0, e1);
    run<18>("v_cvt_f32_u32_sdwa BYTE_1", 32, d_out, d_cyc, e0, e1);
    run<19>("v_lshrrev_b32", 32, d_out, d_cyc, e0, e1);
    run<20>("v_bfe_u32", 32, d_out, d_cyc, e0, e1);
    run<21>("v_cmp_ge_f32_e64 -> sgpr", 32, d_out, d_cyc, e0, e1);
    run<22>("v_mad_u32_u24", 32, d_out, d_cyc, e0, e1);
    run<23>("v_readlane_b32", 32, d_out, d_cyc, e0, e1);
    run<27>("v_lshl_add_u64", 32, d_out, d_cyc, e0, e1);
    run<28>("v_lshlrev_b64", 32, d_out, d_cyc, e0, e1);
    run<29>("v_mad_u64_u32", 32, d_out, d_cyc, e0, e1);
    run<30>("v_add_co_u32", 32, d_out, d_cyc, e0, e1);
    run<31>("v_add_u32", 32, d_out, d_cyc, e0, e1);
    run<32>("v_mov_b32", 32, d_out, d_cyc, e0, e1);
    run<33>("v_mul_lo_u32", 32, d_out, d_cyc, e0, e1);
    run<34>("v_lshl_add_u32", 32, d_out, d_cyc, e0, e1);
    run<35>("v_fma_mix_f32 (f16 lo source)", 32, d_out, d_cyc, e0, e1);
    run<36>("v_fma_mix_f32 (f16 hi source)", 32, d_out, d_cyc, e0, e1);
    run<37>("v_cvt_f32_f16", 32, d_out, d_cyc, e0, e1);
    run<38>("node test child, bytes (14 instr x2)", 28, d_out, d_cyc, e0, e1);
    run<39>("node test child, f16 + fma_mix (11 instr x2)", 22, d_out, d_cyc, e0, e1);
    run<40>("v_or_b32_sdwa BYTE_1 (0x4B000000 | byte)", 32, d_out, d_cyc, e0, e1);
    run<41>("node test child, or_sdwa + 6 plain fma (17 instr x2)", 34, d_out, d_cyc, e0, e1);
    run<10>("s_add_u32", 32, d_out, d_cyc, e0, e1);
    run<11>("cvt+fma+max3+cmp (x8)", 32, d_out, d_cyc, e0, e1);
    run<12>("v_fma_f32 + s_add_u32 (x16 pairs)", 32, d_out, d_cyc, e0, e1);
    return 0;
}
