// mem_probe.hip — what the memory side of an MI355X delivers to 16-B-per-lane loads, the access shape of every hot load on the path
// (nodes, triangle packets, queue entries).  Build: hipcc --offload-arch=gfx950 -O3 -o mem_probe mem_probe.hip ; prints one JSON object.
//   copy_*      device-to-device float4 copies (read + written bytes / time): the job's measured HBM roofline; variants of the kernel
//   l1_stream   every wave re-reads its own 4 KiB (resident in the CU's vector L1): bytes = 1 KiB per wave-instruction
//   l1_same     all 64 lanes of a wave read the SAME 64-B node with 4 dwordx4 loads (coherent rays at the top of a tree)
//   l1_gather   every lane reads a different 64-B node out of an L1-resident 8 KiB set (divergent lanes, node-sized accesses)
//   l2_gather   the same out of 2 MiB per workgroup set shared by the chip (L1 misses, L2 hits)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                     \
    do {                                                                                          \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

__global__ __launch_bounds__(256) void k_copy_stride(const float4* __restrict__ s, float4* __restrict__ d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (size_t)gridDim.x * 256u) d[i] = s[i];
}
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT> __global__ __launch_bounds__(256) void k_copy_unroll(const float4* __restrict__ s4, float4* __restrict__ d4, size_t n)
{
    const v4f* __restrict__ s = reinterpret_cast<const v4f*>(s4);
    v4f* __restrict__ d = reinterpret_cast<v4f*>(d4);
    // each workgroup owns consecutive chunks of U * 256 elements; U loads in flight per lane before the first store
    const size_t chunk = (size_t)U * 256u;
    for (size_t base = (size_t)blockIdx.x * chunk; base < n; base += (size_t)gridDim.x * chunk) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256u + threadIdx.x;
            if (i < n) v[u] = NT ? __builtin_nontemporal_load(s + i) : s[i];
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const size_t i = base + (size_t)u * 256u + threadIdx.x;
            if (i < n) {
                if (NT) __builtin_nontemporal_store(v[u], d + i);
                else d[i] = v[u];
            }
        }
    }
}

// MODE 0: own 4 KiB per wave, streamed; 1: the same 64 B for all lanes; 2: per-lane random 64-B node of an 8 KiB set; 3: of a 2 MiB set
template <int MODE> __global__ __launch_bounds__(256) void k_read(const uint4* __restrict__ buf, uint32_t* __restrict__ out, int iters, uint32_t set_nodes)
{
    const uint32_t lane = threadIdx.x & 63u, wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    uint4 acc = make_uint4(0, 0, 0, 0);
    uint32_t rnd = wave * 2654435761u + lane * 40503u + 17u;
    for (int it = 0; it < iters; it++) {
        const uint4* p;
        if (MODE == 0) {
            p = buf + (size_t)(wave & 1023u) * 256u + lane;                         // 4 loads x 1 KiB, lane-contiguous
            asm volatile("" : "+v"(p)); // the address is the same every trip: without this the compiler loads once, outside the loop (round 2 printed 13.7 PB/s)
        }
        else if (MODE == 1) p = buf + (size_t)((wave + it) & 127u) * 4u;           // one 64-B node for the whole wave
        else {
            rnd = rnd * 1664525u + 1013904223u;
            p = buf + (size_t)((rnd >> 8) % set_nodes) * 4u;
        }
        if (MODE == 0) {
            const uint4 a = p[0], b = p[64], c = p[128], d = p[192];
            acc.x += a.x ^ b.y; acc.y += c.z ^ d.w; acc.z += a.w + c.x; acc.w += b.z + d.y;
        } else {
            const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
            acc.x += a.x ^ b.y; acc.y += c.z ^ d.w; acc.z += a.w + c.x; acc.w += b.z + d.y;
            if (MODE >= 2) rnd ^= acc.x & 1u; // the next address depends on the data, as a child pointer does
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wave] = acc.x; // keep the loads
}

template <typename F> static double time_ms(F f, int reps)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) f();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main()
{
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    float4 *a, *b;
    CK(hipMalloc(&a, bytes));
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0x3c, bytes));
    CK(hipMemset(b, 0, bytes));
    printf("{");
    auto rep = [&](const char* name, double ms) { printf("\"%s_GBps\": %.1f, ", name, 2.0 * bytes / (ms * 1e-3) / 1e9); };
    for (int blocks : {2048, 4096, 8192, 16384, 65536}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy_stride_%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL(k_copy_stride, dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
    }
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, sizeof nm, "copy_u4_%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((k_copy_unroll<4, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
        snprintf(nm, sizeof nm, "copy_u4nt_%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((k_copy_unroll<4, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
        snprintf(nm, sizeof nm, "copy_u8_%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((k_copy_unroll<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
        snprintf(nm, sizeof nm, "copy_u8nt_%d", blocks);
        rep(nm, time_ms([&] { hipLaunchKernelGGL((k_copy_unroll<8, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
    }
    {   // exactly one chunk per workgroup (no loop): n / (8 * 256) workgroups
        const int blocks = (int)(n / (8 * 256));
        rep("copy_u8_one_chunk_per_wg", time_ms([&] { hipLaunchKernelGGL((k_copy_unroll<8, false>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
        rep("copy_u8nt_one_chunk_per_wg", time_ms([&] { hipLaunchKernelGGL((k_copy_unroll<8, true>), dim3(blocks), dim3(256), 0, 0, a, b, n); }, 10));
    }
    rep("hipMemcpyDtoD", time_ms([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }, 10));
    // read-rate probes: 256 CUs x 8 workgroups of 4 waves, 2000 iterations of 4 dwordx4 loads (4 KiB per wave and iteration at the TA)
    uint32_t* out;
    CK(hipMalloc(&out, 1 << 20));
    const int wgs = 256 * 8, iters = 2000;
    const double kib = (double)wgs * 4 * iters * 4.0; // wave-instructions x 1 KiB
    auto rd = [&](const char* name, double ms) { printf("\"%s_TA_GBps\": %.1f, \"%s_Ginst_per_s\": %.2f, ", name, kib * 1024.0 / (ms * 1e-3) / 1e9, name, kib / (ms * 1e-3) / 1e9); };
    rd("l1_stream", time_ms([&] { hipLaunchKernelGGL(k_read<0>, dim3(wgs), dim3(256), 0, 0, (const uint4*)a, out, iters, 0u); }, 5));
    rd("l1_same", time_ms([&] { hipLaunchKernelGGL(k_read<1>, dim3(wgs), dim3(256), 0, 0, (const uint4*)a, out, iters, 0u); }, 5));
    rd("l1_gather", time_ms([&] { hipLaunchKernelGGL(k_read<2>, dim3(wgs), dim3(256), 0, 0, (const uint4*)a, out, iters, 128u); }, 5));
    rd("l2_gather", time_ms([&] { hipLaunchKernelGGL(k_read<3>, dim3(wgs), dim3(256), 0, 0, (const uint4*)a, out, iters, 32768u); }, 5));
    rd("mall_gather", time_ms([&] { hipLaunchKernelGGL(k_read<3>, dim3(wgs), dim3(256), 0, 0, (const uint4*)a, out, iters, 1500000u); }, 5));
    printf("\"note\": \"copy: read + written bytes over time, 1 GiB buffers; reads: 1 KiB per wave-instruction at the texture-address unit\"}\n");
    return 0;
}
