// kernels.hip — the wavefront path-tracing kernels for gfx950 (wave64).
//
// One kernel per stage of the reference's per-bounce loop (backends/gpu-rt/src/lib.rs:1708-1728):
//   k_primary  = ray_gen.comp   (generate + trace primary)      k_extend = ray_extend.comp
//   k_shade    = shade.comp                                       k_shadow = ray_shadow.comp
//   k_blit     = blit.comp
// with the queues compacted by wavefront ballot + one atomic per wave (shade.comp:250,261 issue one
// global atomic per thread) and all queue counters resident on the device, one slot per bounce, so a
// frame is a fixed sequence of launches with no host read-back (gpu-rt/src/lib.rs:2052-2069 blocks
// on a map_async per bounce).
#include "kernels.h"
#include "shade_device.h"
#include "traverse.h"
#include "traverse_packet.h"

// waves per SIMD the trace kernels are compiled for (register budget 512 / waves), chosen by measurement with frames in flight:
// closest hit (k_primary, k_extend) 6 (5: -6 %, 7: -2 %, 8: -11 %); any hit (k_shadow: fewer live values) 8 (7: -1.4 %, 6: -2.8 %)
// the streaming flavours alike; the packet kernels 8 (6 -> 6755, 7 -> 6785, 8 -> 6855 Mrays/s: 16 spilled registers at 8 cost less than the two extra waves hide)
constexpr int kTraceWaves = 6, kTraceWavesAny = 8, kStreamWaves = 6, kStreamWavesAny = 8, kPacketWaves = 8;

namespace rfwhip {

// ---------------------------------------------------------------- shard / slab indexing
// Local path index -> pixel.  Tiles of tile_size^2 pixels are dealt round-robin to ranks; inside a tile, pixels are
// ordered in 8x8 blocks so that one wavefront = one 8x8 pixel block (coherent primary rays), and the blocks of a tile follow a Z
// curve when the tile has a power-of-two number of them per row (tile sizes 8 ... 128): consecutive blocks then stay close in the image
// (the 8 blocks of one shade workgroup cover 32 x 16 pixels instead of a 64 x 8 strip; measured +1.9 % frame rate, mostly cache locality).
// Other tile sizes keep the blocks row by row.  rfw-rs_amd/dist.py has the numpy twin of these two functions.
RFW_DI bool morton_tile(uint32_t ts) { const uint32_t bpr = ts >> 3; return bpr <= 16u && (bpr & (bpr - 1u)) == 0u; }
RFW_DI uint32_t morton_even_bits(uint32_t v) { return (v & 1u) | ((v >> 1) & 2u) | ((v >> 2) & 4u) | ((v >> 3) & 8u); }       // bits 0,2,4,6 -> 0..3
RFW_DI uint32_t morton_spread_bits(uint32_t v) { return (v & 1u) | ((v & 2u) << 1) | ((v & 4u) << 2) | ((v & 8u) << 3); }    // bits 0..3 -> 0,2,4,6
// The index arithmetic below runs once per path and kernel; a 32-bit division is ~25 instructions on this device.  Tile sizes are powers of two
// in practice (shifts: CameraParams::tile_shift), and the divisions by the frame's width / tiles per row are multiplications by a reciprocal the
// host has proved exact for every index that can occur (index_magic in api_frame.cpp; 0 = no such constant: divide)
RFW_DI uint32_t div_magic(const uint32_t n, const uint32_t d, const uint32_t magic) { return magic ? __umulhi(n, magic) : n / d; }
RFW_DI bool slab_to_pixel(const CameraParams& c, uint32_t idx, uint32_t& px, uint32_t& py)
{
    const uint32_t ts = c.tile_size, per_tile = ts * ts;
    const uint32_t lt = c.tile_shift != 0xffffffffu ? idx >> (2u * c.tile_shift) : idx / per_tile, within = idx - lt * per_tile;
    const uint32_t tile = lt * c.world + c.rank;
    if (tile >= c.tiles_x * c.tiles_y) return false;
    const uint32_t ty = div_magic(tile, c.tiles_x, c.tiles_x_magic), tx = tile - ty * c.tiles_x;
    const uint32_t block = within >> 6, lane = within & 63u, bpr = ts >> 3;
    uint32_t by, bx;
    if (morton_tile(ts)) {
        bx = morton_even_bits(block);
        by = morton_even_bits(block >> 1);
    } else {
        by = block / bpr;
        bx = block - by * bpr;
    }
    px = tx * ts + bx * 8u + (lane & 7u);
    py = ty * ts + by * 8u + (lane >> 3);
    return px < c.width && py < c.height;
}
RFW_DI uint32_t pixel_to_slab(const CameraParams& c, uint32_t px, uint32_t py, uint32_t& owner)
{
    const uint32_t ts = c.tile_size;
    const bool pow2 = c.tile_shift != 0xffffffffu;
    const uint32_t tx = pow2 ? px >> c.tile_shift : px / ts, ty = pow2 ? py >> c.tile_shift : py / ts;
    const uint32_t tile = ty * c.tiles_x + tx;
    uint32_t lt = tile;
    owner = 0;
    if (c.world != 1u) { // (uniform)
        lt = tile / c.world;
        owner = tile - lt * c.world;
    }
    const uint32_t ix = px - tx * ts, iy = py - ty * ts;
    const uint32_t block = morton_tile(ts) ? (morton_spread_bits(ix >> 3) | (morton_spread_bits(iy >> 3) << 1)) : (iy >> 3) * (ts >> 3) + (ix >> 3);
    return lt * ts * ts + block * 64u + ((iy & 7u) << 3) + (ix & 7u);
}

RFW_DI SceneView scene_view(const SceneDev& sc)
{
    SceneView v;
    v.tlas_nodes = sc.tlas_nodes;
    v.tlas_prims = sc.tlas_prims;
    v.instances = sc.instances;
    v.blas_nodes = sc.blas_nodes;
    v.tlas_oct = sc.tlas_oct;
    v.blas_oct = sc.blas_oct;
    v.tlas_oct_stride = sc.tlas_wide_stride;
    v.blas_oct_stride = sc.blas_wide_stride;
    v.tri_packets = sc.tri_packets;
    v.spill = sc.spill;
    v.spill_stride = sc.spill_stride;
    v.spill_rows = sc.spill_rows;
    v.overflow_flag = sc.overflow_flag;
    v.counters = sc.counters;
    return v;
}

// the first workgroup of a frame's first kernel clears the counter block the NEXT frame of this stream will use (SceneDev::counters_next)
RFW_DI void clear_next_counters(const SceneDev& sc)
{
    static_assert(sizeof(QueueCounters) % 8 == 0, "QueueCounters is cleared in 8-byte words");
    if (blockIdx.x != 0 || !sc.counters_next) return;
    unsigned long long* z = reinterpret_cast<unsigned long long*>(sc.counters_next);
    for (uint32_t i = threadIdx.x; i < sizeof(QueueCounters) / 8u; i += kTraceBlock) z[i] = 0ull;
}
template <bool COUNT> RFW_DI void flush_counters(QueueCounters* qc, const TravCounters& tc, const int kind)
{
    if (!COUNT) return;
    // wave reduction, then one atomic per wave
    unsigned long long n = tc.nodes, t = tc.tris, i = tc.insts, wn = tc.wave_nodes, wt = tc.wave_tris, wu = tc.wave_uniform;
    uint32_t mx = tc.nodes;
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        n += __shfl_down(n, off);
        wn += __shfl_down(wn, off);
        wt += __shfl_down(wt, off);
        wu += __shfl_down(wu, off);
        t += __shfl_down(t, off);
        i += __shfl_down(i, off);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(&qc->trav[kind][0], n);
        atomicAdd(&qc->trav[kind][1], t);
        atomicAdd(&qc->trav[kind][2], i);
        atomicAdd(&qc->wave_max_nodes[kind], (unsigned long long)mx);
        atomicMax(&qc->max_nodes[kind], (unsigned long long)mx);
        atomicAdd(&qc->wave_exec[kind][0], wn);
        atomicAdd(&qc->wave_exec[kind][1], wt);
        atomicAdd(&qc->wave_uniform[kind], wu);
    }
}

// ---------------------------------------------------------------- instance descriptors (gpu-rt/src/lib.rs:1589-1615)
// matrix -> inverse (explicit cofactor expansion, every term left to right) and normal = transpose(inverse)
RFW_DI void prepare_instance(const uint32_t i, const rfw_mat4* __restrict__ matrices, const uint32_t* __restrict__ mesh_of_instance,
                             const MeshRecord* __restrict__ meshes, InstanceXform* __restrict__ xf, InstanceNormal* __restrict__ nm)
{
    float m[16], inv[16];
    bool zero = true;
    for (int k = 0; k < 16; k++) {
        m[k] = matrices[i].m[k];
        zero = zero && (m[k] == 0.0f);
    }
    const uint32_t mesh = mesh_of_instance[i];
    InstanceXform x;
    InstanceNormal nn;
    const bool valid = !zero && mesh != 0xffffffffu && meshes[mesh].tri_count > 0;
    inv[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] + m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    inv[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] - m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    inv[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] + m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    inv[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] - m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    inv[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] - m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    inv[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] + m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    inv[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] - m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    inv[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] + m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    inv[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] + m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    inv[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] - m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    inv[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] + m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    inv[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] - m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    inv[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] - m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    inv[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] + m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    inv[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] - m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    inv[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] + m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    det = 1.0f / det;
    for (int k = 0; k < 16; k++) inv[k] = inv[k] * det;
    // traversal rows: row i of the inverse = (inv[i], inv[4+i], inv[8+i], inv[12+i])
    for (int c = 0; c < 4; c++) {
        x.inv_r0[c] = inv[4 * c + 0];
        x.inv_r1[c] = inv[4 * c + 1];
        x.inv_r2[c] = inv[4 * c + 2];
    }
    // normal matrix N = transpose(inverse): N[col c][row r] = inv[col r][row c]; row i of N = (inv[4i], inv[4i+1], inv[4i+2], inv[4i+3])
    for (int c = 0; c < 4; c++) {
        nn.n_r0[c] = inv[0 + c];
        nn.n_r1[c] = inv[4 + c];
        nn.n_r2[c] = inv[8 + c];
    }
    x.node_base = valid ? meshes[mesh].node_base : 0u;
    x.tri_base = valid ? meshes[mesh].tri_base : 0u;
    // bit 1: the rows of the inverse are EXACTLY (1 0 0 0) (0 1 0 0) (0 0 1 0), bit for bit (+0, not -0): the transform of a ray into this
    // instance's space is then v * 1 + (+0) + (+0) + (+0) per component, i.e. v + 0.0f — the traversal adds the zero and skips the matrix
    bool ident = valid;
    for (int c = 0; c < 4; c++)
        ident = ident && fbits(x.inv_r0[c]) == (c == 0 ? 0x3f800000u : 0u) && fbits(x.inv_r1[c]) == (c == 1 ? 0x3f800000u : 0u) && fbits(x.inv_r2[c]) == (c == 2 ? 0x3f800000u : 0u);
    x.flags = (valid ? 1u : 0u) | (ident ? kInstanceIdentity : 0u);
    x.mesh = mesh;
    xf[i] = x;
    nm[i] = nn;
}
__global__ void k_prepare_instances(const rfw_mat4* __restrict__ matrices, const uint32_t* __restrict__ mesh_of_instance,
                                    const MeshRecord* __restrict__ meshes, uint32_t n, InstanceXform* __restrict__ xf,
                                    InstanceNormal* __restrict__ nm)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) prepare_instance(i, matrices, mesh_of_instance, meshes, xf, nm);
}

// ---------------------------------------------------------------- ray_gen.comp:72-91 == shade.comp:530-545
// The blue-noise sampler of the first 256 samples.  Its tables are run-time input (rfw_hip_set_blue_noise: the 5 x 65536 words of
// gpu_rt::blue_noise::create_blue_noise_buffer(), kept as bytes): [0, 65536) Sobol bytes, [65536, ...) scrambling tile, [3 * 65536, ...)
// ranking tile.  A read past the end (the ranking lookup of dimensions 8..15 in the last pixel of a tile) returns 0.
RFW_DI int bn_at(const uint8_t* __restrict__ t, int idx) { return (uint32_t)idx < kBlueNoiseWords ? (int)t[idx] : 0; }
RFW_DI float blueNoiseSampler(const uint8_t* __restrict__ t, uint32_t sample_count, int x, int y, int sampleDimension)
{
    x &= 127;
    y &= 127;
    const int sampleIdx = (int)((sample_count + 1u) & 255u);
    sampleDimension &= 255;
    const int rankedSampleIndex = sampleIdx ^ bn_at(t, sampleDimension + (x + y * 128) * 8 + 65536 * 3);
    int value = bn_at(t, sampleDimension + rankedSampleIndex * 256);
    value ^= bn_at(t, (sampleDimension & 7) + (x + y * 128) * 8 + 65536);
    return (0.5f + (float)value) * (1.0f / 256.0f);
}

// ---------------------------------------------------------------- ray_gen.comp:103-146
RFW_DI void generate_eye_ray(const CameraParams& cam, f3& O, f3& D, uint32_t sx, uint32_t sy, uint32_t& seed, const uint8_t* __restrict__ blue_noise,
                             const uint32_t sample_count)
{
    float r0, r1, r2, r3;
    if (blue_noise != nullptr && sample_count < 256u) { // ray_gen.comp:109-115 (uniform branch)
        r0 = blueNoiseSampler(blue_noise, sample_count, (int)sx, (int)sy, 0);
        r1 = blueNoiseSampler(blue_noise, sample_count, (int)sx, (int)sy, 1);
        r2 = blueNoiseSampler(blue_noise, sample_count, (int)sx, (int)sy, 2);
        r3 = blueNoiseSampler(blue_noise, sample_count, (int)sx, (int)sy, 3);
    } else {
        r0 = randf(seed);
        r1 = randf(seed);
        r2 = randf(seed);
        r3 = randf(seed);
    }
    const f3 pos = mk3(cam.pos[0], cam.pos[1], cam.pos[2]);
    const f3 right = mk3(cam.right[0], cam.right[1], cam.right[2]);
    const f3 up = mk3(cam.up[0], cam.up[1], cam.up[2]);
    const f3 p1 = mk3(cam.p1[0], cam.p1[1], cam.p1[2]);
    O = pos;
    if (cam.lens_size != 0.0f) { // pinhole (uniform branch): pos + 0 * (finite) == pos, so the 9-blade lens sample is only drawn, not evaluated
        const float blade = (float)f2i(r0 * 9.0f);
        r2 = (r2 - blade * (1.0f / 9.0f)) * 9.0f;
        float x1, y1, x2, y2;
        const float piOver4point5 = 3.14159265359f / 4.5f;
        rfw_sincosf(blade * piOver4point5, &y1, &x1);
        rfw_sincosf((blade + 1.0f) * piOver4point5, &y2, &x2);
        if ((r2 + r3) > 1.0f) {
            r2 = 1.0f - r2;
            r3 = 1.0f - r3;
        }
        const float xr = x1 * r2 + x2 * r3;
        const float yr = y1 * r2 + y2 * r3;
        O = pos + cam.lens_size * (right * xr + up * yr);
    }
    const float u = ((float)(int)sx + r0) * (1.0f / (float)(int)cam.width);
    const float v = ((float)(int)sy + r1) * (1.0f / (float)(int)cam.height);
    const f3 pointOnPixel = p1 + u * right + v * up;
    D = normalize(pointOnPixel - O);
}

// ---------------------------------------------------------------- XCD-aware block mapping
// Workgroups are dealt round-robin over the 8 XCDs (launch indices b and b + 8 share an XCD and its 4 MiB L2).  The launch
// index is remapped so that the 64 wavefronts of one 64x64-pixel tile (64 consecutive queue blocks) run on ONE XCD, and the
// tiles are dealt round-robin to the XCDs: an XCD's L2 then holds the BVH working set of its own tiles instead of every L2
// replicating all of them, while the load stays balanced tile by tile.  Speed only: each block is still visited exactly once.
// Launch grids are padded to a multiple of 512 (8 tiles x 64 blocks).
RFW_DI uint32_t xcd_block(const uint32_t b)
{
    return (((b >> 9) << 3) + (b & 7u)) * 64u + ((b & 511u) >> 3);
}

// the same for the streaming kernels, whose wavefronts own runs of 64 x run entries: a 64x64-pixel tile is 64 / run wavefronts
RFW_DI uint32_t xcd_run(const uint32_t b, const uint32_t per_tile)
{
    const uint32_t group = 8u * per_tile;
    return ((b / group) * 8u + (b & 7u)) * per_tile + ((b % group) >> 3);
}

// ---------------------------------------------------------------- ray_gen.comp:39-70
template <bool COUNT> __global__ __launch_bounds__(kTraceBlock, kTraceWaves) void k_primary(const CameraParams cam, const SceneDev sc, const PathDev p)
{
    clear_next_counters(sc);
    __shared__ uint32_t s_stack[kTraceLdsRows * kTraceBlock];
    const uint32_t block = xcd_block(blockIdx.x);
    const uint32_t idx = block * kTraceBlock + threadIdx.x;
    TravCounters tc{0, 0, 0};
    uint32_t px = 0, py = 0;
    const bool valid = idx < p.capacity && slab_to_pixel(cam, idx, px, py);
    if (valid) {
        if (cam.sample_count == 0) p.acc[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const uint32_t path_id = px + py * cam.width;
        uint32_t seed = wang_hash(path_id * 16789u + cam.sample_count * 1791u + 0u * 720898027u);
        f3 O, D;
        generate_eye_ray(cam, O, D, px, py, seed, sc.blue_noise, cam.sample_count);
        float t = 1e26f, hu = 0.0f, hv = 0.0f;
        int32_t hi = -1, ht = -1;
        const SceneView sv = scene_view(sc);
        p.ray_o[0][idx] = make_float4(O.x, O.y, O.z, bitsf(path_id)); // before the trace: the ray need not stay live across it
        p.ray_d[0][idx] = make_float4(D.x, D.y, D.z, 0.0f);
        traverse<false, COUNT>(sv, O, D, 1e-4f, t, hu, hv, hi, ht, s_stack, threadIdx.x, block * kTraceBlock, tc);
        const uint32_t bary = f2u(65535.0f * hu) + (f2u(65535.0f * hv) << 16);
        p.hit[0][idx] = make_uint4((uint32_t)hi, (uint32_t)ht, fbits(t), bary);
    } else if (idx < p.capacity) {
        p.hit[0][idx] = make_uint4(kNoPath, 0u, 0u, 0u); // a slab slot without a pixel (ragged edge tile): k_shade skips it without redoing the index arithmetic
    }
    flush_counters<COUNT>(sc.counters, tc, 0);
}

// The packet flavour (traverse_packet.h): the 64 camera rays of an 8x8-pixel block walk the tree together on one shared stack.  Same rays,
// same triangle tests, same image; no LDS.
template <bool COUNT> __global__ __launch_bounds__(kTraceBlock, kPacketWaves) void k_primary_packet(const CameraParams cam, const SceneDev sc, const PathDev p)
{
    clear_next_counters(sc);
    const uint32_t block = xcd_block(blockIdx.x);
    const uint32_t idx = block * kTraceBlock + threadIdx.x;
    TravCounters tc{0, 0, 0};
    uint32_t px = 0, py = 0;
    const bool valid = idx < p.capacity && slab_to_pixel(cam, idx, px, py);
    f3 O = mk3(0.0f), D = mk3(0.0f);
    if (valid) {
        if (cam.sample_count == 0) p.acc[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        const uint32_t path_id = px + py * cam.width;
        uint32_t seed = wang_hash(path_id * 16789u + cam.sample_count * 1791u + 0u * 720898027u);
        generate_eye_ray(cam, O, D, px, py, seed, sc.blue_noise, cam.sample_count);
        p.ray_o[0][idx] = make_float4(O.x, O.y, O.z, bitsf(path_id));
        p.ray_d[0][idx] = make_float4(D.x, D.y, D.z, 0.0f);
    }
    float t = 1e26f, hu = 0.0f, hv = 0.0f;
    int32_t hi = -1, ht = -1;
    bool occluded;
    const SceneView sv = scene_view(sc);
    traverse_packet<false, COUNT>(sv, sc.tlas_wide, sc.tlas_wide_stride, sc.blas_wide, sc.blas_wide_stride, valid, O, D, 1e-4f, t, hu, hv, hi, ht, occluded, tc);
    if (valid) {
        const uint32_t bary = f2u(65535.0f * hu) + (f2u(65535.0f * hv) << 16);
        p.hit[0][idx] = make_uint4((uint32_t)hi, (uint32_t)ht, fbits(t), bary);
    } else if (idx < p.capacity) {
        p.hit[0][idx] = make_uint4(kNoPath, 0u, 0u, 0u);
    }
    flush_counters<COUNT>(sc.counters, tc, 0);
}

// the same for a batch of independent frames (rfw_hip_render_batch): one launch covers the tiles of every frame of the batch
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock, kTraceWaves) void k_primary_batch(const CameraParams cam, const BatchViews views, const SceneDev sc, const PathDev p)
{
    clear_next_counters(sc);
    __shared__ uint32_t s_stack[kTraceLdsRows * kTraceBlock];
    const uint32_t block = xcd_block(blockIdx.x);
    const uint32_t idx = block * kTraceBlock + threadIdx.x;
    const uint32_t f = (block * kTraceBlock) / cam.frame_capacity; // uniform: a frame's range is a multiple of the workgroup size
    TravCounters tc{0, 0, 0};
    uint32_t px = 0, py = 0;
    const bool valid = f < cam.batch && slab_to_pixel(cam, idx - f * cam.frame_capacity, px, py);
    if (valid) {
        const uint32_t sample = cam.batch_sample[f];
        // a batch of new images clears every frame; a batch of SAMPLES of one image (rfw_hip_render_samples) keeps what frame 0 has
        // accumulated so far (sample_count = samples already in the image)
        if (!(f == 0u && cam.sample_count != 0u)) p.acc[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        CameraParams c = cam;
        const FrameView& v = views.v[f];
        for (int k = 0; k < 3; k++) { c.pos[k] = v.pos[k]; c.right[k] = v.right[k]; c.up[k] = v.up[k]; c.p1[k] = v.p1[k]; }
        c.lens_size = v.lens_size;
        const uint32_t path_id = px + py * cam.width;
        uint32_t seed = wang_hash(path_id * 16789u + sample * 1791u + 0u * 720898027u);
        f3 O, D;
        generate_eye_ray(c, O, D, px, py, seed, sc.blue_noise, sample);
        float t = 1e26f, hu = 0.0f, hv = 0.0f;
        int32_t hi = -1, ht = -1;
        const SceneView sv = scene_view(sc);
        p.ray_o[0][idx] = make_float4(O.x, O.y, O.z, bitsf(path_id | (f << 24)));
        p.ray_d[0][idx] = make_float4(D.x, D.y, D.z, 0.0f);
        traverse<false, COUNT>(sv, O, D, 1e-4f, t, hu, hv, hi, ht, s_stack, threadIdx.x, block * kTraceBlock, tc);
        const uint32_t bary = f2u(65535.0f * hu) + (f2u(65535.0f * hv) << 16);
        p.hit[0][idx] = make_uint4((uint32_t)hi, (uint32_t)ht, fbits(t), bary);
    } else if (f < cam.batch) {
        p.hit[0][idx] = make_uint4(kNoPath, 0u, 0u, 0u);
    }
    flush_counters<COUNT>(sc.counters, tc, 0);
}

// ... and the packet flavour for a batch of frames: a wavefront's 64 paths belong to ONE frame (a frame's range is a multiple of the workgroup
// size), so they are one 8x8-pixel block of one view as in k_primary_packet
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock, kPacketWaves) void k_primary_batch_packet(const CameraParams cam, const BatchViews views, const SceneDev sc, const PathDev p)
{
    clear_next_counters(sc);
    const uint32_t block = xcd_block(blockIdx.x);
    const uint32_t idx = block * kTraceBlock + threadIdx.x;
    const uint32_t f = (block * kTraceBlock) / cam.frame_capacity;
    TravCounters tc{0, 0, 0};
    uint32_t px = 0, py = 0;
    const bool valid = f < cam.batch && slab_to_pixel(cam, idx - f * cam.frame_capacity, px, py);
    f3 O = mk3(0.0f), D = mk3(0.0f);
    if (valid) {
        const uint32_t sample = cam.batch_sample[f];
        if (!(f == 0u && cam.sample_count != 0u)) p.acc[idx] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        CameraParams c = cam;
        const FrameView& v = views.v[f];
        for (int k = 0; k < 3; k++) { c.pos[k] = v.pos[k]; c.right[k] = v.right[k]; c.up[k] = v.up[k]; c.p1[k] = v.p1[k]; }
        c.lens_size = v.lens_size;
        const uint32_t path_id = px + py * cam.width;
        uint32_t seed = wang_hash(path_id * 16789u + sample * 1791u + 0u * 720898027u);
        generate_eye_ray(c, O, D, px, py, seed, sc.blue_noise, sample);
        p.ray_o[0][idx] = make_float4(O.x, O.y, O.z, bitsf(path_id | (f << 24)));
        p.ray_d[0][idx] = make_float4(D.x, D.y, D.z, 0.0f);
    }
    float t = 1e26f, hu = 0.0f, hv = 0.0f;
    int32_t hi = -1, ht = -1;
    bool occluded;
    const SceneView sv = scene_view(sc);
    traverse_packet<false, COUNT>(sv, sc.tlas_wide, sc.tlas_wide_stride, sc.blas_wide, sc.blas_wide_stride, valid, O, D, 1e-4f, t, hu, hv, hi, ht, occluded, tc);
    if (valid) {
        const uint32_t bary = f2u(65535.0f * hu) + (f2u(65535.0f * hv) << 16);
        p.hit[0][idx] = make_uint4((uint32_t)hi, (uint32_t)ht, fbits(t), bary);
    } else if (f < cam.batch) {
        p.hit[0][idx] = make_uint4(kNoPath, 0u, 0u, 0u);
    }
    flush_counters<COUNT>(sc.counters, tc, 0);
}

// ---------------------------------------------------------------- extension rays in spatial order (option "sort_extension_rays")
// Extension rays leave a surface in BSDF-sampled directions: a wavefront of 64 consecutive queue entries (neighbouring pixels) shares
// little of its traversal (node-test lane utilisation 0.2).  Key = 9-bit-per-axis Morton code of the ray's origin inside the scene's
// bounds (the TLAS root's box) over the direction's octant; sorting (key, queue index) pairs and tracing through the sorted indices
// makes a wavefront's rays start in one cell and leave it roughly the same way.  The queue itself keeps its order: shade reads
// entry i as before, so nothing downstream changes and the image is bit-identical.
RFW_DI uint32_t spread3_9(uint32_t v) // 9 bits -> every third bit
{
    v &= 0x1ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}
__global__ __launch_bounds__(256) void k_extension_keys(const SceneDev sc, const PathDev p, const uint32_t bounce, uint32_t* __restrict__ keys, uint32_t* __restrict__ vals)
{
    const uint32_t idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= p.capacity) return;
    const uint32_t count = sc.counters->ext[bounce - 1];
    vals[idx] = idx;
    if (idx >= count) { keys[idx] = 0xffffffffu; return; } // padding sorts behind every ray
    const Node4Q root = sc.tlas_nodes[0];
    const float4 o = p.ray_o[bounce & 1u][idx], d = p.ray_d[bounce & 1u][idx];
    // cell of the origin in a 512^3 grid over the root's (quantisation) box: origin + [0, 255] * scale per axis
    const float fx = (o.x - root.ox) / (255.0f * root.sx), fy = (o.y - root.oy) / (255.0f * root.sy), fz = (o.z - root.oz) / (255.0f * root.sz);
    const uint32_t cx = (uint32_t)gl_clamp(fx * 512.0f, 0.0f, 511.0f), cy = (uint32_t)gl_clamp(fy * 512.0f, 0.0f, 511.0f), cz = (uint32_t)gl_clamp(fz * 512.0f, 0.0f, 511.0f);
    const uint32_t morton = spread3_9(cx) | (spread3_9(cy) << 1) | (spread3_9(cz) << 2); // 27 bits
    const uint32_t octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    keys[idx] = ((morton >> 6) << 9) | (octant << 6) | (morton & 63u); // coarse cell (64^3), then octant, then the fine cell inside it: 30 bits
}
void launch_extension_keys(hipStream_t s, const SceneDev& sc, const PathDev& p, uint32_t bounce, uint32_t* keys, uint32_t* vals)
{
    if (p.capacity) hipLaunchKernelGGL(k_extension_keys, dim3((p.capacity + 255u) / 256u), dim3(256), 0, s, sc, p, bounce, keys, vals);
}

// ---------------------------------------------------------------- ray_extend.comp:245-268
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock, kTraceWaves) void k_extend(const CameraParams cam, const SceneDev sc, const PathDev p, const uint32_t bounce,
                                                                         const uint32_t* __restrict__ order)
{
    __shared__ uint32_t s_stack[kTraceLdsRows * kTraceBlock];
    const uint32_t block = xcd_block(blockIdx.x);
    const uint32_t idx = block * kTraceBlock + threadIdx.x;
    const uint32_t count = sc.counters->ext[bounce - 1];
    if (block * kTraceBlock >= count) return;
    TravCounters tc{0, 0, 0};
    const uint32_t half = bounce & 1u;
    if (idx < count) {
        const uint32_t j = order ? order[idx] : idx; // queue entry this lane traces (sorted order, or the queue's own)
        const float4 o4 = p.ray_o[half][j], d4 = p.ray_d[half][j];
        const f3 O = mk3(o4.x, o4.y, o4.z), D = mk3(d4.x, d4.y, d4.z);
        float t = 1e26f, hu = 0.0f, hv = 0.0f;
        int32_t hi = -1, ht = -1;
        const SceneView sv = scene_view(sc);
        traverse<false, COUNT>(sv, O, D, 1e-4f, t, hu, hv, hi, ht, s_stack, threadIdx.x, block * kTraceBlock, tc);
        const uint32_t bary = f2u(65535.0f * hu) + (f2u(65535.0f * hv) << 16);
        p.hit[half][j] = make_uint4((uint32_t)hi, (uint32_t)ht, fbits(t), bary);
    }
    flush_counters<COUNT>(sc.counters, tc, 1);
}

// The streaming flavour (traverse.h, traverse_stream): a wavefront owns cam.stream_run x 64 consecutive entries of the extension queue
struct ExtendStream {
    const PathDev& p;
    const uint32_t* order;
    uint32_t half;
    uint32_t next, end; // wave-uniform: the run's entries not handed out yet
    uint32_t j;         // the lane's entry
    RFW_DI bool more() const { return next < end; }
    static constexpr float kTMin = 1e-4f;
    RFW_DI bool fetch(const uint64_t idle, f3& O, f3& D, float& t)
    {
        const uint32_t i = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (i >= end) return false;
        j = order ? order[i] : i;
        const float4 o4 = p.ray_o[half][j], d4 = p.ray_d[half][j];
        O = mk3(o4.x, o4.y, o4.z);
        D = mk3(d4.x, d4.y, d4.z);
        t = 1e26f;
        return true;
    }
    RFW_DI void advance(const uint64_t idle) { next += (uint32_t)__popcll(idle); }
    RFW_DI void commit(bool, const float t, const float hu, const float hv, const int32_t hi, const int32_t ht)
    {
        const uint32_t bary = f2u(65535.0f * hu) + (f2u(65535.0f * hv) << 16);
        p.hit[half][j] = make_uint4((uint32_t)hi, (uint32_t)ht, fbits(t), bary);
    }
};
template <bool COUNT>
__global__ __launch_bounds__(kTraceBlock, kStreamWaves) void k_extend_stream(const CameraParams cam, const SceneDev sc, const PathDev p, const uint32_t bounce,
                                                                                const uint32_t* __restrict__ order)
{
    __shared__ uint32_t s_stack[kTraceLdsRows * kTraceBlock];
    const uint32_t run = cam.stream_run * kTraceBlock;
    const uint32_t block = xcd_run(blockIdx.x, 64u / cam.stream_run);
    const uint32_t count = sc.counters->ext[bounce - 1];
    if (block * run >= count) return;
    TravCounters tc{0, 0, 0};
    ExtendStream st{p, order, bounce & 1u, block * run, min(count, (block + 1u) * run), 0u};
    const SceneView sv = scene_view(sc);
    traverse_stream<false, COUNT, false>(sv, st, cam.stream_refill & 0xffu, cam.stream_refill >> 8, s_stack, threadIdx.x, blockIdx.x * kTraceBlock, tc);
    flush_counters<COUNT>(sc.counters, tc, 1);
}

// ---------------------------------------------------------------- ray_shadow.comp:245-268
// Buckets are walked from the LAST light index down (directional lights come last in the reference's light order, shade.comp:471-527): a
// measured choice — k_shadow alone 0.364 -> 0.336 ms when it was made, 0.382 -> 0.329 ms with the far-to-near order of the directional
// light's rays below; with frames in flight the order of the buckets does not matter.
RFW_DI bool bucket_far_first(const CameraParams& cam, const uint32_t bucket);
template <bool COUNT, bool BATCH = false>
__global__ __launch_bounds__(kTraceBlock, kTraceWavesAny) void k_shadow(const CameraParams cam, const SceneDev sc, const PathDev p, const uint32_t bounce)
{
    __shared__ uint32_t s_stack[kTraceLdsRowsAny * kTraceBlock];
    // the queue is bucketed by light (shade pushes directional lights into the last region, positional lights into region light % 7): walk the buckets, each padded to whole wavefronts, so
    // the 64 rays of a wavefront start on neighbouring pixels AND aim at the same light
    uint32_t block = xcd_block(blockIdx.x);
    const uint32_t spill_base = block * kTraceBlock; // (of the launch index's block: unique per wavefront, whatever bucket it lands in)
    uint32_t bucket = 0, count = 0;
    {
        bool found = false;
        for (int kk = 0; kk < kShadowBuckets; kk++) {
            const int k = kShadowBuckets - 1 - kk;
            if ((cam.flags & kFlagPacketShadowFar) && bucket_far_first(cam, (uint32_t)k)) continue; // (those went as packets: launch_shadow)
            const uint32_t c = sc.counters->shadow[bounce][k];
            const uint32_t nb = (c + kTraceBlock - 1) / kTraceBlock;
            if (!found) {
                if (block < nb) { found = true; bucket = (uint32_t)k; count = c; }
                else block -= nb;
            }
        }
        if (!found) return;
    }
    const uint32_t local = block * kTraceBlock + threadIdx.x;
    const uint32_t idx = bucket * p.capacity + local;
    TravCounters tc{0, 0, 0};
    if (local < count) {
        const float4 o4 = p.sh_o[idx], d4 = p.sh_d[idx];
        const f3 O = mk3(o4.x, o4.y, o4.z), D = mk3(d4.x, d4.y, d4.z);
        float t = d4.w - 0.0001f, hu, hv;
        if (t > 3.0e38f) t = 3.0e38f; // a light at infinite distance (see k_query_closest)
        int32_t hi = -1, ht = -1;
        const SceneView sv = scene_view(sc);
        // Rays towards a directional light leave the scene: what blocks the sky is most often the LAST thing on their way (roofs, upper
        // floors), so their occluder search starts at the far end (measured on the bench scene's real shadow queue: 21.6 -> 12.1 nodes per
        // ray; rays towards the area lights get up to 15 % longer that way and keep the near-to-far order).  shade files every directional
        // light's rays under the last bucket, so the bucket tells the kind of light.
        const bool far_first = bucket == (uint32_t)kShadowBuckets - 1u ? !(cam.flags & kFlagNearFirstDirectional) : (cam.flags & kFlagFarFirstPositional) != 0u; // option "shadow_order" overrides the default per light kind
        const bool occluded = far_first ? traverse<true, COUNT, true>(sv, O, D, 0.001f, t, hu, hv, hi, ht, s_stack, threadIdx.x, spill_base, tc)
                                        : traverse<true, COUNT, false>(sv, O, D, 0.001f, t, hu, hv, hi, ht, s_stack, threadIdx.x, spill_base, tc);
        if (!occluded) {
            // (the entry's index is formed again from the wave-uniform part and the lane: kept across the traversal it is a 64-bit register
            // pair that went to scratch memory at 8 waves per SIMD)
            uint32_t lane = threadIdx.x;
            asm volatile("" : "+v"(lane));
            const float4 e = p.sh_e[bucket * p.capacity + block * kTraceBlock + lane];
            const uint32_t slot = fbits(e.w); // the path's accumulator slot rides in the queue entry (k_shade knows it without arithmetic)
            // single writer per pixel per pass (one shadow ray per path per bounce), as ray_shadow.comp:268
            float4 a = p.acc[slot];
            a.x += e.x; a.y += e.y; a.z += e.z; a.w += 0.0f;
            p.acc[slot] = a;
        }
    }
    flush_counters<COUNT>(sc.counters, tc, 2);
}

// The packet flavour for the camera paths' shadow rays (bounce 0): the 64 rays of a wavefront start on neighbouring pixels and aim at one
// light.  One instantiation per visiting order, each launch walking the buckets of its order (as the streaming flavour below).
RFW_DI bool bucket_far_first(const CameraParams& cam, const uint32_t bucket);
template <bool COUNT, bool FAR>
__global__ __launch_bounds__(kTraceBlock, kPacketWaves) void k_shadow_packet(const CameraParams cam, const SceneDev sc, const PathDev p, const uint32_t bounce)
{
    uint32_t block = xcd_block(blockIdx.x);
    uint32_t bucket = 0, count = 0;
    {
        bool found = false;
        for (int kk = 0; kk < kShadowBuckets; kk++) {
            const int k = kShadowBuckets - 1 - kk;
            if (bucket_far_first(cam, (uint32_t)k) != FAR) continue;
            const uint32_t c = sc.counters->shadow[bounce][k];
            const uint32_t nb = (c + kTraceBlock - 1) / kTraceBlock;
            if (!found) {
                if (block < nb) { found = true; bucket = (uint32_t)k; count = c; }
                else block -= nb;
            }
        }
        if (!found) return;
    }
    const uint32_t local = block * kTraceBlock + threadIdx.x;
    const uint32_t idx = bucket * p.capacity + local;
    TravCounters tc{0, 0, 0};
    const bool valid = local < count;
    f3 O = mk3(0.0f), D = mk3(0.0f);
    float t = 1.0f, hu, hv;
    if (valid) {
        const float4 o4 = p.sh_o[idx], d4 = p.sh_d[idx];
        O = mk3(o4.x, o4.y, o4.z);
        D = mk3(d4.x, d4.y, d4.z);
        t = d4.w - 0.0001f;
        if (t > 3.0e38f) t = 3.0e38f;
    }
    int32_t hi = -1, ht = -1;
    bool occluded;
    const SceneView sv = scene_view(sc);
    traverse_packet<true, COUNT, FAR>(sv, sc.tlas_wide, sc.tlas_wide_stride, sc.blas_wide, sc.blas_wide_stride, valid, O, D, 0.001f, t, hu, hv, hi, ht, occluded, tc);
    if (valid && !occluded) {
        const float4 e = p.sh_e[idx];
        const uint32_t slot = fbits(e.w);
        float4 a = p.acc[slot]; // single writer per pixel per pass, as ray_shadow.comp:268
        a.x += e.x; a.y += e.y; a.z += e.z; a.w += 0.0f;
        p.acc[slot] = a;
    }
    flush_counters<COUNT>(sc.counters, tc, 2);
}

struct ShadowStream {
    const PathDev& p;
    uint32_t base;      // the bucket's first queue entry
    uint32_t next, end; // wave-uniform: the run's entries (bucket-local) not handed out yet
    uint32_t idx;       // the lane's entry
    RFW_DI bool more() const { return next < end; }
    static constexpr float kTMin = 0.001f;
    RFW_DI bool fetch(const uint64_t idle, f3& O, f3& D, float& t)
    {
        const uint32_t i = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
        if (i >= end) return false;
        idx = base + i;
        const float4 o4 = p.sh_o[idx], d4 = p.sh_d[idx];
        O = mk3(o4.x, o4.y, o4.z);
        D = mk3(d4.x, d4.y, d4.z);
        t = d4.w - 0.0001f;
        if (t > 3.0e38f) t = 3.0e38f;
        return true;
    }
    RFW_DI void advance(const uint64_t idle) { next += (uint32_t)__popcll(idle); }
    RFW_DI void commit(const bool occluded, float, float, float, int32_t, int32_t)
    {
        if (occluded) return;
        const float4 e = p.sh_e[idx];
        const uint32_t slot = fbits(e.w);
        float4 a = p.acc[slot]; // single writer per pixel per pass, whichever lane traces the ray
        a.x += e.x; a.y += e.y; a.z += e.z; a.w += 0.0f;
        p.acc[slot] = a;
    }
};
// One instantiation per visiting order (FAR: hit children by decreasing exit distance): a kernel that holds both traversals spills ~50 scalar
// registers into vector lanes.  Each launch walks the buckets of its own order only.
RFW_DI bool bucket_far_first(const CameraParams& cam, const uint32_t bucket)
{
    return bucket == (uint32_t)kShadowBuckets - 1u ? !(cam.flags & kFlagNearFirstDirectional) : (cam.flags & kFlagFarFirstPositional) != 0u;
}
template <bool COUNT, bool FAR>
__global__ __launch_bounds__(kTraceBlock, kStreamWavesAny) void k_shadow_stream(const CameraParams cam, const SceneDev sc, const PathDev p, const uint32_t bounce)
{
    __shared__ uint32_t s_stack[kTraceLdsRowsAny * kTraceBlock];
    const uint32_t run = cam.stream_run * kTraceBlock;
    uint32_t block = xcd_run(blockIdx.x, 64u / cam.stream_run);
    uint32_t bucket = 0, count = 0;
    {
        bool found = false;
        for (int kk = 0; kk < kShadowBuckets; kk++) {
            const int k = kShadowBuckets - 1 - kk;
            if (bucket_far_first(cam, (uint32_t)k) != FAR) continue;
            const uint32_t c = sc.counters->shadow[bounce][k];
            const uint32_t nb = (c + run - 1u) / run;
            if (!found) {
                if (block < nb) { found = true; bucket = (uint32_t)k; count = c; }
                else block -= nb;
            }
        }
        if (!found) return;
    }
    TravCounters tc{0, 0, 0};
    ShadowStream st{p, bucket * p.capacity, block * run, min(count, (block + 1u) * run), 0u};
    const SceneView sv = scene_view(sc);
    traverse_stream<true, COUNT, FAR>(sv, st, cam.stream_refill & 0xffu, cam.stream_refill >> 8, s_stack, threadIdx.x, blockIdx.x * kTraceBlock, tc);
    flush_counters<COUNT>(sc.counters, tc, 2);
}

// ---------------------------------------------------------------- shade.comp:70-266
// Workgroup size of k_shade: the wavefronts of a workgroup share ONE atomic per queue (a returning atomic on one address retires at ~88 per us
// chip-wide), so larger is cheaper — but with several frames in flight a workgroup of 8 wavefronts waits for 8 free wavefront slots and
// its LDS on one CU while the trace kernels' single-wavefront workgroups take every slot the moment it frees (round 5: a launch that takes
// 0.11 ms alone was resident for 1-2.7 ms).  256 where frames overlap (+1.5 % headline, +4 % path traced), 512 where one call fills the chip
// by itself (a batch, samples per call, one frame at a time: there 256 only doubles the atomics — path traced batches -9 %): launch_shade
// the octant-major filing of the extension rays below lets the SECOND wavefront prefix-sum the 8 x (kShadeBlock / 64) counts, one per lane
template <bool BATCH, int kShadeBlock>
__global__ __launch_bounds__(kShadeBlock) void k_shade(const CameraParams cam, const SceneDev sc, const PathDev p, const uint32_t bounce)
{
    static_assert(kShadeBlock >= 128 && 8 * (kShadeBlock / 64) <= 64, "k_shade: the extension-ray filing needs a second wavefront and at most 64 (octant, wavefront) counts");
    const uint32_t idx = blockIdx.x * kShadeBlock + threadIdx.x;
    const uint32_t count = bounce == 0 ? p.capacity : sc.counters->ext[bounce - 1];
    if (blockIdx.x * kShadeBlock >= count) return;
    const uint32_t half = bounce & 1u, next_half = half ^ 1u;
    const uint32_t path_length = bounce;

    bool live = idx < count;

    bool push_ext = false, push_shadow = false;
    int light_bucket = 0;
    f3 ext_o = mk3(0.0f), ext_d = mk3(0.0f), ext_thr = mk3(0.0f);
    float ext_pdf = 0.0f;
    uint32_t ext_normal = 0;
    f3 sh_o = mk3(0.0f), sh_d = mk3(0.0f), sh_e = mk3(0.0f);
    float sh_dist = 0.0f;
    uint32_t PATH_ID = 0, PATH_WORD = 0, sh_slot = 0;

    uint4 S = make_uint4(kNoPath, 0u, 0u, 0u);
    if (live) {
        S = p.hit[half][idx];
        live = S.x != kNoPath; // bounce 0: a slab slot of a ragged edge tile that holds no pixel (marked by k_primary)
    }
    if (live) {
        const float4 O4 = p.ray_o[half][idx], D4 = p.ray_d[half][idx];
        const f3 O = mk3(O4.x, O4.y, O4.z), D = mk3(D4.x, D4.y, D4.z);
        f3 throughput = mk3(1.0f);
        float bsdfPdf = 1.0f;
        if (path_length != 0) {
            const float4 T4 = p.thr[half][idx];
            throughput = mk3(T4.x, T4.y, T4.z);
            bsdfPdf = T4.w;
        }
        const uint32_t path_word = fbits(O4.w); // batch: frame index in the top byte
        PATH_ID = BATCH ? (path_word & 0xffffffu) : path_word;
        PATH_WORD = path_word;
        // bounce 0: path idx of the (tall) virtual frame IS its accumulator slot; later bounces recompute it from the pixel
        uint32_t owner, slot = idx;
        if (bounce != 0) {
            const uint32_t py_ = div_magic(PATH_ID, cam.width, cam.width_magic);
            slot = pixel_to_slab(cam, PATH_ID - py_ * cam.width, py_, owner) + (BATCH ? (path_word >> 24) * cam.frame_capacity : 0u);
        }
        const int32_t INST_ID = (int32_t)S.x;
        const uint32_t TRI_ID = S.y;
        const float T_VAL = bitsf(S.z);

        if (INST_ID < 0) { // shade.comp:90-96: lat-long skybox lookup at LOD = path length (constant sky colour when none is set)
            f3 sky = mk3(cam.sky[0], cam.sky[1], cam.sky[2]);
            if (sc.skybox.mips) {
                const float su = 0.5f * (1.0f + rfw_atan2f(D.x, -D.z) * (1.0f / 3.14159265359f));
                const float sv = 1.0f - rfw_acosf(D.y) * (1.0f / 3.14159265359f);
                const f4 c = texture_sample(sc.tex_data, sc.skybox, su, sv, (float)(int)path_length);
                sky = mk3(c.x, c.y, c.z);
            }
            f3 contribution = throughput * sky * (1.0f / bsdfPdf);
            CLAMPINTENSITY(contribution, cam.clamp_value);
            float4 a = p.acc[slot];
            a.x += contribution.x; a.y += contribution.y; a.z += contribution.z; a.w += 0.0f;
            p.acc[slot] = a;
        } else {
            // 176-B RTTriangle as 11 dwordx4 loads (only here, once per shaded hit)
            const float4* tp = reinterpret_cast<const float4*>(sc.triangles + TRI_ID);
            const float4 q3 = tp[3], q4 = tp[4], q5 = tp[5], q6 = tp[6], T0 = tp[7], T1 = tp[8], T2 = tp[9];
            const uint4 q10 = *reinterpret_cast<const uint4*>(tp + 10);
            const int32_t mat_id = (int32_t)q10.y;
            const float tri_area = bitsf(q10.w);
            ShadingData sd = extractParameters(sc.materials + mat_id);

            // shade.comp:102: pathId / (w * h) + sample count, and a path id is a pixel index here; a frame of a batch carries its own sample index
            const uint32_t sampleId = BATCH ? cam.batch_sample[path_word >> 24] : cam.sample_count;
            uint32_t seed = wang_hash(PATH_ID * 16789u + sampleId * 1791u + path_length * 720898027u);

            const float u = (float)(S.w & 65535u) * (1.0f / 65535.0f);
            const float v = (float)(S.w >> 16) * (1.0f / 65535.0f);
            const float w = 1.0f - u - v;

            f3 gN = mk3(q3.x, q3.y, q3.z);
            f3 N = w * mk3(q4.x, q4.y, q4.z) + u * mk3(q5.x, q5.y, q5.z) + v * mk3(q6.x, q6.y, q6.z);
            f3 T = w * mk3(T0.x, T0.y, T0.z) + u * mk3(T1.x, T1.y, T1.z) + v * mk3(T2.x, T2.y, T2.z);
            const float Tw = w * T0.w + u * T1.w + v * T2.w;

            const float4* np = reinterpret_cast<const float4*>(sc.instance_normals + INST_ID);
            const float4 n0 = np[0], n1 = np[1], n2 = np[2];
            gN = normalize(xform_rows(n0, n1, n2, gN, 0.0f));
            N = normalize(xform_rows(n0, n1, n2, N, 0.0f));
            T = normalize(xform_rows(n0, n1, n2, T, 0.0f));
            const f3 B = cross(N, T) * Tw;
            const f3 P = O + T_VAL * D;

            const bool any_map = (sd.flags & 63u) != 0u;
            // shade.comp:128-160 hit a light (a material with an emissive map is shaded as a surface instead)
            if ((sd.color.x > 1.0f || sd.color.y > 1.0f || sd.color.z > 1.0f) && !(sd.flags & RFW_MAT_HAS_EMISSIVE_MAP)) {
                f3 contribution = mk3(0.0f);
                const float DdotNL = -dot(D, N);
                bool add = true;
                if (DdotNL > 0.0f) {
                    if (path_length == 0) {
                        contribution = throughput * sd.color * (1.0f / bsdfPdf);
                    } else {
                        const float lightPdf = CalculateLightPDF(D, T_VAL, tri_area, N);
                        const int lc = (int)(cam.area_light_count + cam.point_light_count + cam.spot_light_count + cam.directional_light_count);
                        const float pickProb = 1.0f / (float)lc;
                        if ((bsdfPdf + lightPdf * pickProb) <= 0.0f) add = false;
                        else contribution = throughput * sd.color * (1.0f / (bsdfPdf + lightPdf * pickProb));
                    }
                    CLAMPINTENSITY(contribution, cam.clamp_value);
                }
                if (add) {
                    float4 a = p.acc[slot];
                    a.x += contribution.x; a.y += contribution.y; a.z += contribution.z; a.w += 0.0f;
                    p.acc[slot] = a;
                }
            } else {
                if (any_map) { // shade.comp:162-175
                    const float lambda = __builtin_sqrtf(bitsf(q10.z)) + rfw_log2f(cam.spread_angle * (1.0f / gl_abs(dot(D, N))));
                    const float4 q0 = tp[0], q1 = tp[1], q2 = tp[2];
                    const float tu = w * q0.w + u * q1.w + v * q2.w;   // u0, u1, u2 ride in the .w of the vertex slots
                    const float tv = w * q3.w + u * q4.w + v * q5.w;   // v0, v1, v2 in the .w of normal / n0 / n1
                    if ((sd.flags & RFW_MAT_HAS_DIFFUSE_MAP) && sd.diffuse_map >= 0 && (uint32_t)sd.diffuse_map < sc.n_textures) {
                        const f4 c = fetchTexelTrilinear(sc.tex_data, sc.tex_desc[sd.diffuse_map], lambda, tu, tv);
                        sd.color = sd.color * mk3(c.x, c.y, c.z);
                    }
                    if ((sd.flags & RFW_MAT_HAS_NORMAL_MAP) && sd.normal_map >= 0 && (uint32_t)sd.normal_map < sc.n_textures) {
                        const f4 c = texture_sample(sc.tex_data, sc.tex_desc[sd.normal_map], tu, tv, (float)f2i(lambda));
                        const f3 m = (mk3(c.x, c.y, c.z) - mk3(0.5f)) * 2.0f;
                        N = normalize((T * m.x + B * m.y) + N * m.z); // mat3(T, B, N) * m
                    }
                }
                prepare_tint(sd);
                const bool backFacing = dot(D, gN) >= 0.0f;
                if (backFacing) {
                    N = N * -1.0f;
                    gN = gN * -1.0f;
                }
                throughput = throughput * (1.0f / bsdfPdf);
                float newBsdfPdf = 0.0f;
                f3 R = mk3(0.0f);
                float r1, r2;
                const bool blue = sc.blue_noise != nullptr && sampleId < 256u; // shade.comp:189-195; the xorshift seed is not advanced on this branch
                const int by = (int)div_magic(PATH_ID, cam.width, cam.width_magic), bx = (int)(PATH_ID - (uint32_t)by * cam.width);
                if (blue) {
                    r1 = blueNoiseSampler(sc.blue_noise, sampleId, bx, by, (int)(4u + 4u * path_length));
                    r2 = blueNoiseSampler(sc.blue_noise, sampleId, bx, by, (int)(5u + 4u * path_length));
                } else {
                    r1 = randf(seed);
                    r2 = randf(seed);
                }
                const float F_gN = Fr_of(sd, dot(gN, D * -1.0f)); // the Fresnel term both pdfs of this hit use (sampled direction, light direction: same normal, same wo)
                const f3 bsdf = SampleBSDF(sd, N, gN, T, B, D * -1.0f, T_VAL, backFacing, r1, r2, R, newBsdfPdf, F_gN);
                throughput = throughput * bsdf * gl_abs(dot(N, R));
                throughput = gl_max(throughput, mk3(0.0f));
                if (!(newBsdfPdf <= 1e-4f || gl_isnan(newBsdfPdf))) {
                    const int lc = (int)(cam.area_light_count + cam.point_light_count + cam.spot_light_count + cam.directional_light_count);
                    if (!(cam.flags & kFlagNoNee) && lc > 0) {
                        float r3; // r4 (dimension 7 + 4 * path length, or the next xorshift draw) is drawn by the reference and never used
                        if (blue) r3 = blueNoiseSampler(sc.blue_noise, sampleId, bx, by, (int)(6u + 4u * path_length)); // shade.comp:216-221
                        else r3 = randf(seed);
                        LightView lv;
                        lv.area = sc.area_lights; lv.point = sc.point_lights; lv.spot = sc.spot_lights; lv.directional = sc.directional_lights;
                        lv.n_area = (int)cam.area_light_count; lv.n_point = (int)cam.point_light_count;
                        lv.n_spot = (int)cam.spot_light_count; lv.n_directional = (int)cam.directional_light_count;
                        f3 lightColor = mk3(0.0f);
                        float pickProb = 0.0f, lightPdf = 0.0f;
                        int picked = 0;
                        f3 L = RandomPointOnLight(lv, r3, P, N, pickProb, lightPdf, lightColor, picked) - P;
                        const float dist = length(L);
                        L = L * (1.0f / dist);
                        const float NdotL = dot(L, N);
                        if (NdotL > 0.0f && lightPdf > 0.0f) {
                            float shadowPdf = 0.0f;
                            const f3 sampledBSDF = EvaluateBSDF(sd, gN, D * -1.0f, L, shadowPdf, F_gN);
                            if (shadowPdf > 0.0f) {
                                f3 contribution = throughput * sampledBSDF * lightColor * (NdotL / (lightPdf * pickProb));
                                if (!(gl_isnan(contribution.x) || gl_isnan(contribution.y) || gl_isnan(contribution.z))) {
                                    CLAMPINTENSITY(contribution, cam.clamp_value);
                                    sh_o = safe_origin(P, L, gN);
                                    sh_d = L;
                                    sh_dist = dist - 1e-4f;
                                    sh_e = contribution;
                                    sh_slot = slot;
                                    push_shadow = true;
                                    // directional lights share the LAST bucket (their rays are traced far to near, k_shadow), the positional lights
                                    // are dealt over the other seven: the bucket tells the kind of light however many lights there are
                                    light_bucket = picked >= lc - (int)cam.directional_light_count ? kShadowBuckets - 1 : (int)((uint32_t)picked % (uint32_t)(kShadowBuckets - 1));
                                }
                            }
                        }
                    }
                    // the reference pushes the extension unconditionally (shade.comp:260-265) and its host loop simply stops
                    // after the last bounce; nothing reads that last queue, so it is not written here
                    if (bounce + 1 < cam.max_path_length) {
                        ext_o = safe_origin(P, R, gN);
                        ext_d = R;
                        ext_normal = PackNormal(N);
                        ext_thr = throughput;
                        ext_pdf = newBsdfPdf;
                        push_ext = true;
                    }
                }
            }
        }
    }

    // ---- queue compaction: ballot + mbcnt prefix inside each wavefront, wave totals combined through LDS, ONE atomic per
    // workgroup and queue (the reference issues one global atomic per thread, shade.comp:250,261; one per wavefront still
    // serialised 32 k returning atomics on a single address)
    constexpr int kQ = kShadowBuckets + 1; // queues: one shadow region per light bucket + the extension queue
    __shared__ uint32_t s_cnt[kQ][kShadeBlock / 64];
    __shared__ uint32_t s_base[kQ];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t my_rank = 0;
    for (int q = 0; q < kShadowBuckets; q++) {
        const bool mine = push_shadow && light_bucket == q;
        const unsigned long long m = __ballot(mine);
        if (lane == 0) s_cnt[q][wave] = (uint32_t)__popcll(m);
        if (mine) my_rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
    }
    // The extension rays of a workgroup go into its range of the queue OCTANT BY OCTANT of their direction (round 4): the trace kernels read
    // the copy of the tree made for a ray's octant, so a wavefront whose 64 consecutive entries share one or two octants touches one or two
    // copies of every node instead of up to eight (measured: k_extend 1.63 -> 1.54 ms per frame alone; +0.4 ... 2.5 % frame rate path traced).  The queue order
    // never shows in the image.
    __shared__ uint32_t s_ext[8][kShadeBlock / 64]; // [octant][wavefront]: count, then exclusive offset inside the workgroup's range
    const bool any_ext = bounce + 1 < cam.max_path_length; // (uniform)
    const uint32_t ext_oct = (fbits(ext_d.x) >> 31) | ((fbits(ext_d.y) >> 31) << 1) | ((fbits(ext_d.z) >> 31) << 2);
    uint32_t r_ex = 0;
    if (any_ext) {
        uint32_t total_wave = 0;
        for (uint32_t o8 = 0; o8 < 8u; o8++) {
            const bool mine = push_ext && ext_oct == o8;
            const unsigned long long m = __ballot(mine);
            if (lane == 0) s_ext[o8][wave] = (uint32_t)__popcll(m);
            if (mine) r_ex = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            total_wave += (uint32_t)__popcll(m);
        }
        if (lane == 0) s_cnt[kShadowBuckets][wave] = total_wave;
    } else if (lane == 0) s_cnt[kShadowBuckets][wave] = 0u;
    __syncthreads();
    if (threadIdx.x < kQ) {
        uint32_t tot = 0;
        for (int k = 0; k < kShadeBlock / 64; k++) tot += s_cnt[threadIdx.x][k];
        uint32_t base = 0;
        if (tot) base = atomicAdd(threadIdx.x < kShadowBuckets ? &sc.counters->shadow[bounce][threadIdx.x] : &sc.counters->ext[bounce], tot);
        s_base[threadIdx.x] = base;
    }
    uint32_t ext_before = 0;
    constexpr uint32_t kExtCounts = 8u * (kShadeBlock / 64);
    if (any_ext && threadIdx.x >= 64u && threadIdx.x < 64u + kExtCounts) { // the second wavefront: (octant, wavefront) t's offset = everything filed before it, octant-major
        const uint32_t t8 = threadIdx.x - 64u;
        const uint32_t* flat = &s_ext[0][0];
        for (uint32_t k = 0; k < t8; k++) ext_before += flat[k];
    }
    __syncthreads();
    if (any_ext && threadIdx.x >= 64u && threadIdx.x < 64u + kExtCounts) (&s_ext[0][0])[threadIdx.x - 64u] = ext_before;
    __syncthreads();
    if (push_shadow) {
        uint32_t off = s_base[light_bucket];
        for (uint32_t k = 0; k < wave; k++) off += s_cnt[light_bucket][k];
        const size_t j = (size_t)light_bucket * p.capacity + off + my_rank;
        p.sh_o[j] = make_float4(sh_o.x, sh_o.y, sh_o.z, bitsf(PATH_WORD));
        p.sh_d[j] = make_float4(sh_d.x, sh_d.y, sh_d.z, sh_dist);
        p.sh_e[j] = make_float4(sh_e.x, sh_e.y, sh_e.z, bitsf(sh_slot));
    }
    if (push_ext) {
        const uint32_t j = s_base[kShadowBuckets] + s_ext[ext_oct][wave] + r_ex;
        p.ray_o[next_half][j] = make_float4(ext_o.x, ext_o.y, ext_o.z, bitsf(PATH_WORD));
        p.ray_d[next_half][j] = make_float4(ext_d.x, ext_d.y, ext_d.z, bitsf(ext_normal));
        p.thr[next_half][j] = make_float4(ext_thr.x, ext_thr.y, ext_thr.z, ext_pdf);
    }
}

// ---------------------------------------------------------------- blit.comp:15-23 (+ slab -> frame de-tiling): k_assemble below
// what a rank contributes to the all-gather: the RGB of its slab (alpha is never written: 12 B per pixel on the links instead of 16)
__global__ __launch_bounds__(256) void k_pack_rgb(const float4* __restrict__ acc_slab, float* __restrict__ out, const uint64_t n)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = acc_slab[i];
    out[3 * i] = a.x;
    out[3 * i + 1] = a.y;
    out[3 * i + 2] = a.z;
}
// What a rank contributes to the all-gather when only the FINISHED frame has to travel (option "gather_format"; the accumulators stay on
// the rank that owns the tiles, where the next sample is added): blit.comp's sqrt(acc / samples) as three halves (6 B per pixel) ...
__global__ __launch_bounds__(256) void k_pack_f16(const float4* __restrict__ acc_slab, _Float16* __restrict__ out, const uint64_t n, const uint32_t samples)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const float4 a = acc_slab[i];
    const float s = (float)(int)samples;
    out[3 * i] = (_Float16)__builtin_sqrtf(a.x * 1.0f / s);
    out[3 * i + 1] = (_Float16)__builtin_sqrtf(a.y * 1.0f / s);
    out[3 * i + 2] = (_Float16)__builtin_sqrtf(a.z * 1.0f / s);
}
// ... or as the swap-chain image itself, B, G, R, A bytes (4 B per pixel): exactly what k_present makes of the de-tiled frame
struct SrgbSteps { float t[256]; };
__device__ inline uint32_t srgb_encode(const float* s_t, float x)
{
    uint32_t lo = 0; // number of steps <= x (a NaN encodes as 0, like a clamped attachment write)
    for (uint32_t bit = 128; bit != 0; bit >>= 1)
        if (x >= s_t[lo + bit - 1]) lo += bit;
    return lo;
}
__global__ __launch_bounds__(256) void k_pack_bgra8(const float4* __restrict__ acc_slab, uint32_t* __restrict__ out, const uint64_t n, const uint32_t samples,
                                                   const SrgbSteps steps)
{
    __shared__ float s_t[256];
    s_t[threadIdx.x] = steps.t[threadIdx.x];
    __syncthreads();
    const float s = (float)(int)samples;
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const float4 a = acc_slab[i];
        const float r = __builtin_sqrtf(a.x * 1.0f / s), g = __builtin_sqrtf(a.y * 1.0f / s), b = __builtin_sqrtf(a.z * 1.0f / s);
        out[i] = srgb_encode(s_t, b) | (srgb_encode(s_t, g) << 8) | (srgb_encode(s_t, r) << 16) | 0xff000000u;
    }
}
// all-gathered finished slabs [world][frame][slab_elems] -> the frame, de-tiled (halves -> the float frame; bytes -> the presented frame)
__global__ void k_assemble_f16(const CameraParams cam, const _Float16* __restrict__ gathered, const uint64_t slab_elems, float4* __restrict__ frame)
{
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= cam.width || py >= cam.height) return;
    uint32_t owner;
    const uint32_t slot = pixel_to_slab(cam, px, py, owner);
    const uint32_t f = blockIdx.z;
    const _Float16* g = gathered + 3 * (((uint64_t)owner * cam.batch + f) * slab_elems + slot);
    frame[(size_t)f * cam.width * cam.height + px + py * cam.width] = make_float4((float)g[0], (float)g[1], (float)g[2], 0.0f);
}
__global__ void k_assemble_bgra8(const CameraParams cam, const uint32_t* __restrict__ gathered, const uint64_t slab_elems, uint32_t* __restrict__ presented)
{
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= cam.width || py >= cam.height) return;
    uint32_t owner;
    const uint32_t slot = pixel_to_slab(cam, px, py, owner);
    const uint32_t f = blockIdx.z;
    presented[(size_t)f * cam.width * cam.height + px + py * cam.width] = gathered[((uint64_t)owner * cam.batch + f) * slab_elems + slot];
}
// rfw_hip_render_samples: the k sample slabs of one image -> slab 0, added in sample order (the same left-to-right sum per pixel whatever
// the launch geometry: deterministic; it differs from k sequential render() calls only in where the partial sums are rounded)
__global__ __launch_bounds__(256) void k_sum_batch(float4* __restrict__ acc, const uint64_t n, const uint32_t count)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 a = acc[i];
    for (uint32_t f = 1; f < count; f++) {
        const float4 b = acc[(uint64_t)f * n + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    acc[i] = a;
}
// Presentation (gpu-rt/src/lib.rs:373,560-585 + shaders/quad.frag): the RGBA32F output is drawn onto a Bgra8UnormSrgb swap chain, i.e.
// clamped, sRGB-encoded and quantised to 8 bits per channel by the attachment write.  Encoding by comparison against the 255 linear
// values at which the encoded byte steps (binary search, 8 compares per channel): exact and identical on every machine.
__global__ __launch_bounds__(256) void k_present(const float4* __restrict__ frame, uint32_t* __restrict__ bgra, const uint64_t n, const SrgbSteps steps)
{
    __shared__ float s_t[256];
    s_t[threadIdx.x] = steps.t[threadIdx.x];
    __syncthreads();
    auto enc = [&](float x) -> uint32_t { return srgb_encode(s_t, x); };
    for (uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256u) {
        const float4 c = frame[i];
        bgra[i] = enc(c.z) | (enc(c.y) << 8) | (enc(c.x) << 16) | 0xff000000u; // B, G, R, A in memory order; the surface is opaque
    }
}
// all-gathered slabs [world][slab_elems] -> full frame.  RGB: the gathered buffer holds 3 floats per element (k_pack_rgb), else the
// instance's own float4 slab (world == 1)
// ACC = false: the finalised frame (blit.comp: sqrt(acc / samples)) — what every frame needs; ACC = true: the linear accumulator itself,
// produced only when somebody asks for it (rfw_hip_read_accumulator): a third of the de-tiling traffic of every frame otherwise
template <bool RGB, bool ACC>
__global__ void k_assemble(const CameraParams cam, const void* __restrict__ gathered_v, const uint64_t slab_elems, float4* __restrict__ frame,
                           const uint32_t samples)
{
    const uint32_t px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y * blockDim.y + threadIdx.y;
    if (px >= cam.width || py >= cam.height) return;
    uint32_t owner;
    const uint32_t slot = pixel_to_slab(cam, px, py, owner);
    const uint32_t f = blockIdx.z; // frame of a batch; gathered = [rank][frame][slot]
    const uint64_t e = ((uint64_t)owner * cam.batch + f) * slab_elems + slot;
    float4 a;
    if (RGB) {
        const float* g = static_cast<const float*>(gathered_v) + 3 * e;
        a = make_float4(g[0], g[1], g[2], 0.0f);
    } else {
        a = static_cast<const float4*>(gathered_v)[e];
    }
    frame += (size_t)f * cam.width * cam.height;
    if (ACC) {
        frame[px + py * cam.width] = a;
    } else {
        const float n = (float)(int)samples;
        frame[px + py * cam.width] = make_float4(__builtin_sqrtf(a.x * 1.0f / n), __builtin_sqrtf(a.y * 1.0f / n), __builtin_sqrtf(a.z * 1.0f / n),
                                                 __builtin_sqrtf(a.w * 1.0f / n));
    }
}

// ---------------------------------------------------------------- ray queries (TIntersector::intersect / occludes)
template <bool DEPTH>
__global__ __launch_bounds__(kTraceBlock) void k_query_closest(const SceneDev sc, const float* __restrict__ origins, const float* __restrict__ directions,
                                                                const float t_min, const float t_max, const uint64_t n, rfw_hip_hit* __restrict__ hits,
                                                                uint32_t* __restrict__ depth)
{
    __shared__ uint32_t s_stack[kTraceLdsRows * kTraceBlock];
    const uint64_t idx = (uint64_t)blockIdx.x * kTraceBlock + threadIdx.x;
    if (idx >= n) return;
    const f3 O = mk3(origins[3 * idx], origins[3 * idx + 1], origins[3 * idx + 2]);
    const f3 D = mk3(directions[3 * idx], directions[3 * idx + 1], directions[3 * idx + 2]);
    // The search interval ends at a FINITE distance.  With t = +inf a degenerate ray (zero direction components: 1 / d = inf) gives child
    // boxes an entry distance of +inf that still passes `entry <= t`; such children tie with the +inf keys of the empty slots in the
    // ordering network, an empty slot's reference (kInvalidRef) can be taken for a child, and it decodes as a leaf far outside the
    // triangle array — a memory fault reproduced through rfw_hip_occludes on the 1 M-triangle scene.  A miss still reports the caller's t_max.
    float t = t_max > 3.0e38f ? 3.0e38f : t_max, hu = 0.0f, hv = 0.0f; // below FLT_MAX, the key the ordering network gives to children that are not hit
    int32_t hi = -1, ht = -1;
    TravCounters tc{0, 0, 0};
    const SceneView sv = scene_view(sc);
    traverse<false, DEPTH>(sv, O, D, t_min, t, hu, hv, hi, ht, s_stack, threadIdx.x, (uint32_t)((idx - threadIdx.x) % sc.spill_stride), tc);
    if (hi < 0) t = t_max;
    rfw_hip_hit h;
    if (hi >= 0) { // storage order -> the boundary's triangle numbering (identical after a full build)
        const MeshRecord r = sc.meshes[sc.instances[hi].mesh];
        ht = (int32_t)((uint32_t)ht - r.tri_base + r.tri_logical);
    }
    h.inst = hi; h.tri = ht; h.t = t; h.u = hu; h.v = hv;
    hits[idx] = h;
    if (DEPTH) depth[idx] = tc.nodes; // 4-wide nodes this ray visited, TLAS and BLAS
}
template <bool DEPTH>
__global__ __launch_bounds__(kTraceBlock) void k_query_any(const SceneDev sc, const float* __restrict__ origins, const float* __restrict__ directions,
                                                            const float t_min, const float* __restrict__ t_max, const uint64_t n,
                                                            uint8_t* __restrict__ occluded, uint32_t* __restrict__ depth)
{
    __shared__ uint32_t s_stack[kTraceLdsRowsAny * kTraceBlock];
    const uint64_t idx = (uint64_t)blockIdx.x * kTraceBlock + threadIdx.x;
    if (idx >= n) return;
    const f3 O = mk3(origins[3 * idx], origins[3 * idx + 1], origins[3 * idx + 2]);
    const f3 D = mk3(directions[3 * idx], directions[3 * idx + 1], directions[3 * idx + 2]);
    float t = t_max[idx], hu, hv;
    if (t > 3.0e38f) t = 3.0e38f; // see k_query_closest: the interval ends at a finite distance (a NaN t_max stays NaN: nothing is hit)
    int32_t hi = -1, ht = -1;
    TravCounters tc{0, 0, 0};
    const SceneView sv = scene_view(sc);
    const bool occ = traverse<true, DEPTH>(sv, O, D, t_min, t, hu, hv, hi, ht, s_stack, threadIdx.x, (uint32_t)((idx - threadIdx.x) % sc.spill_stride), tc);
    occluded[idx] = occ ? 1 : 0;
    if (DEPTH) depth[idx] = tc.nodes; // 4-wide nodes visited until the first occluder / the end of the traversal
}

// Node regions are sized for the worst case (one node per primitive); a tree uses a quarter of its region or less, and the packet kernels'
// copies are 1 KB per node: the slots behind a tree's last node are skipped.  `live` (nullable) points at the tree's node count on the device
// (the builders write it; no read-back); the *_regions flavour covers several trees in one launch and finds a slot's tree by bisection.
__global__ void k_quantize_nodes(const Node4* __restrict__ in, Node4Q* __restrict__ out, uint32_t n, const uint32_t* __restrict__ live)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (!live || i < *live)) out[i] = quantize_node(in[i]);
}
// one thread per (node, octant copy): the 1 KB of copies per node is the larger part of the traffic
__global__ void k_expand_nodes(const Node4Q* __restrict__ in, PacketNode* __restrict__ wide, Node4Q* __restrict__ octq, const uint32_t wide_stride, const uint32_t n,
                               const uint32_t* __restrict__ live)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = g >> 3, oct = g & 7u;
    if (i < n && (!live || i < *live)) {
        const Node4Q q = in[i];
        if (wide) wide[(size_t)oct * wide_stride + i] = make_packet_node(q, oct); // (absent where no packet kernel can run: follow_copies)
        octq[(size_t)oct * wide_stride + i] = make_octant_node(q, oct);
    }
}
RFW_DI bool slot_is_live(const MeshRecord* __restrict__ recs, const uint32_t* __restrict__ counts, const uint32_t n_recs, const uint32_t slot)
{
    uint32_t lo = 0, hi = n_recs; // the record with the largest node_base <= slot (records are laid out in order)
    while (hi - lo > 1u) {
        const uint32_t mid = (lo + hi) >> 1;
        if (recs[mid].node_base <= slot) lo = mid; else hi = mid;
    }
    return slot - recs[lo].node_base < counts[lo];
}
__global__ void k_quantize_regions(const Node4* __restrict__ in, Node4Q* __restrict__ out, const uint32_t n, const MeshRecord* __restrict__ recs,
                                   const uint32_t* __restrict__ counts, const uint32_t n_recs)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && slot_is_live(recs, counts, n_recs, i)) out[i] = quantize_node(in[i]);
}
__global__ void k_expand_regions(const Node4Q* __restrict__ in, PacketNode* __restrict__ wide, Node4Q* __restrict__ octq, const uint32_t wide_stride, const uint32_t n,
                                 const MeshRecord* __restrict__ recs, const uint32_t* __restrict__ counts, const uint32_t n_recs)
{
    const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t i = g >> 3, oct = g & 7u;
    if (i < n && slot_is_live(recs, counts, n_recs, i)) {
        const Node4Q q = in[i];
        if (wide) wide[(size_t)oct * wide_stride + i] = make_packet_node(q, oct); // (absent where no packet kernel can run: follow_copies)
        octq[(size_t)oct * wide_stride + i] = make_octant_node(q, oct);
    }
}

// What follows the fused TLAS build (lbvh.hip, k_tlas_fused) in ONE launch: the first `expand_blocks` workgroups quantise the 4-wide nodes and
// write their eight per-octant copies (k_quantize_nodes + k_expand_nodes: one thread per (node, octant); the quantisation is a few operations,
// done by all eight), the other workgroups make the instance descriptors (k_prepare_instances).  With the one copy of the staging block in
// front of them a frame's whole instance update is three API calls where it was 27.
__global__ __launch_bounds__(256) void k_tlas_finish(const Node4* __restrict__ raw, Node4Q* __restrict__ out, PacketNode* __restrict__ wide, Node4Q* __restrict__ octq,
                                                     const uint32_t wide_stride, const uint32_t n_nodes_max, const uint32_t* __restrict__ live, const uint32_t expand_blocks,
                                                     const rfw_mat4* __restrict__ matrices, const uint32_t* __restrict__ mesh_of_instance, const MeshRecord* __restrict__ meshes,
                                                     const uint32_t n_instances, InstanceXform* __restrict__ xf, InstanceNormal* __restrict__ nm)
{
    if (blockIdx.x < expand_blocks) {
        const uint32_t g = blockIdx.x * 256u + threadIdx.x;
        const uint32_t i = g >> 3, oct = g & 7u;
        if (i < n_nodes_max && i < *live) {
            const Node4Q q = quantize_node(raw[i]);
            if (oct == 0u) out[i] = q;
            if (wide) wide[(size_t)oct * wide_stride + i] = make_packet_node(q, oct);
            if (octq) octq[(size_t)oct * wide_stride + i] = make_octant_node(q, oct);
        }
    } else {
        const uint32_t i = (blockIdx.x - expand_blocks) * 256u + threadIdx.x;
        if (i < n_instances) prepare_instance(i, matrices, mesh_of_instance, meshes, xf, nm);
    }
}

// ---------------------------------------------------------------- launch wrappers
static inline uint32_t ceil_div(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
static SrgbSteps make_steps(const float* steps255)
{
    SrgbSteps st;
    for (int k = 0; k < 255; k++) st.t[k] = steps255[k];
    st.t[255] = __builtin_inff(); // never reached: lo + bit - 1 <= 254
    return st;
}

void launch_prepare_instances(hipStream_t s, const rfw_mat4* matrices, const uint32_t* mesh_of_instance, const MeshRecord* meshes, uint32_t n,
                              InstanceXform* xf, InstanceNormal* nm)
{
    if (n == 0) return;
    hipLaunchKernelGGL(k_prepare_instances, dim3(ceil_div(n, 64)), dim3(64), 0, s, matrices, mesh_of_instance, meshes, n, xf, nm);
}
void launch_tlas_finish(hipStream_t s, const Node4* raw, Node4Q* out, const OctantCopies& oc, uint32_t n_nodes_max, const uint32_t* live, const rfw_mat4* matrices,
                        const uint32_t* mesh_of_instance, const MeshRecord* meshes, uint32_t n_instances, InstanceXform* xf, InstanceNormal* nm)
{
    const uint32_t expand_blocks = ceil_div((uint64_t)n_nodes_max * 8u, 256), prepare_blocks = ceil_div(n_instances, 256);
    if (expand_blocks + prepare_blocks == 0u) return;
    hipLaunchKernelGGL(k_tlas_finish, dim3(expand_blocks + prepare_blocks), dim3(256), 0, s, raw, out, oc.wide, oc.quant, oc.stride, n_nodes_max, live, expand_blocks, matrices,
                       mesh_of_instance, meshes, n_instances, xf, nm);
}
void launch_primary(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, bool count)
{
    const dim3 grid((ceil_div(p.capacity, kTraceBlock) + 511u) & ~511u), block(kTraceBlock);
    if ((cam.flags & kFlagPacketPrimary) && sc.tlas_wide && sc.blas_wide) {
        if (count) hipLaunchKernelGGL(k_primary_packet<true>, grid, block, 0, s, cam, sc, p);
        else hipLaunchKernelGGL(k_primary_packet<false>, grid, block, 0, s, cam, sc, p);
        return;
    }
    if (count) hipLaunchKernelGGL(k_primary<true>, grid, block, 0, s, cam, sc, p);
    else hipLaunchKernelGGL(k_primary<false>, grid, block, 0, s, cam, sc, p);
}
void launch_primary_batch(hipStream_t s, const CameraParams& cam, const BatchViews& views, const SceneDev& sc, const PathDev& p, bool count)
{
    const dim3 grid((ceil_div(p.capacity, kTraceBlock) + 511u) & ~511u), block(kTraceBlock);
    if ((cam.flags & kFlagPacketPrimary) && sc.tlas_wide && sc.blas_wide) {
        if (count) hipLaunchKernelGGL(k_primary_batch_packet<true>, grid, block, 0, s, cam, views, sc, p);
        else hipLaunchKernelGGL(k_primary_batch_packet<false>, grid, block, 0, s, cam, views, sc, p);
        return;
    }
    if (count) hipLaunchKernelGGL(k_primary_batch<true>, grid, block, 0, s, cam, views, sc, p);
    else hipLaunchKernelGGL(k_primary_batch<false>, grid, block, 0, s, cam, views, sc, p);
}
void launch_extend(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, uint32_t bounce, bool count, const uint32_t* order)
{
    const dim3 block(kTraceBlock);
    if (cam.stream_run) {
        const dim3 grid((ceil_div(p.capacity, kTraceBlock * cam.stream_run) + 511u) & ~511u);
        if (count) hipLaunchKernelGGL(k_extend_stream<true>, grid, block, 0, s, cam, sc, p, bounce, order);
        else hipLaunchKernelGGL(k_extend_stream<false>, grid, block, 0, s, cam, sc, p, bounce, order);
        return;
    }
    const dim3 grid((ceil_div(p.capacity, kTraceBlock) + 511u) & ~511u);
    if (count) hipLaunchKernelGGL(k_extend<true>, grid, block, 0, s, cam, sc, p, bounce, order);
    else hipLaunchKernelGGL(k_extend<false>, grid, block, 0, s, cam, sc, p, bounce, order);
}
void launch_shade(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, uint32_t bounce)
{
    const bool small = (cam.flags & kFlagShadeSmallGroups) != 0u; // (do_render: several frame slots and one frame per call, or option "shade_group")
    if (cam.batch > 1) {
        if (small) hipLaunchKernelGGL((k_shade<true, 256>), dim3(ceil_div(p.capacity, 256)), dim3(256), 0, s, cam, sc, p, bounce);
        else hipLaunchKernelGGL((k_shade<true, 512>), dim3(ceil_div(p.capacity, 512)), dim3(512), 0, s, cam, sc, p, bounce);
    } else {
        if (small) hipLaunchKernelGGL((k_shade<false, 256>), dim3(ceil_div(p.capacity, 256)), dim3(256), 0, s, cam, sc, p, bounce);
        else hipLaunchKernelGGL((k_shade<false, 512>), dim3(ceil_div(p.capacity, 512)), dim3(512), 0, s, cam, sc, p, bounce);
    }
}
void launch_shadow(hipStream_t s, const CameraParams& cam_in, const SceneDev& sc, const PathDev& p, uint32_t bounce, bool count)
{
    CameraParams cam = cam_in;
    // (k_shadow leaves the far-to-near buckets to the packet launch below only when that launch happens)
    if (!(bounce == 0 && sc.tlas_wide && sc.blas_wide && cam.batch <= 1) || (cam.flags & kFlagPacketShadow)) cam.flags &= ~kFlagPacketShadowFar;
    if (bounce == 0 && (cam.flags & kFlagPacketShadow) && sc.tlas_wide && sc.blas_wide && cam.batch <= 1) {
        // worst case: every path pushed a shadow ray, every bucket padded to whole wavefronts; one launch per visiting order
        const uint32_t blocks_ = (p.capacity + kTraceBlock - 1) / kTraceBlock + kShadowBuckets;
        const dim3 grid_(((blocks_ + 511u) / 512u) * 512u), block_(kTraceBlock);
        if (count) {
            hipLaunchKernelGGL((k_shadow_packet<true, true>), grid_, block_, 0, s, cam, sc, p, bounce);
            hipLaunchKernelGGL((k_shadow_packet<true, false>), grid_, block_, 0, s, cam, sc, p, bounce);
        } else {
            hipLaunchKernelGGL((k_shadow_packet<false, true>), grid_, block_, 0, s, cam, sc, p, bounce);
            hipLaunchKernelGGL((k_shadow_packet<false, false>), grid_, block_, 0, s, cam, sc, p, bounce);
        }
        return;
    }
    if (cam.flags & kFlagPacketShadowFar) {
        // packets for the buckets traced far to near only (the directional lights: parallel rays from neighbouring pixels); k_shadow below skips them
        const uint32_t blocks_ = (p.capacity + kTraceBlock - 1) / kTraceBlock + kShadowBuckets;
        const dim3 grid_(((blocks_ + 511u) / 512u) * 512u), block_(kTraceBlock);
        if (count) hipLaunchKernelGGL((k_shadow_packet<true, true>), grid_, block_, 0, s, cam, sc, p, bounce);
        else hipLaunchKernelGGL((k_shadow_packet<false, true>), grid_, block_, 0, s, cam, sc, p, bounce);
    }
    const dim3 block(kTraceBlock);
    // streaming pays where a wavefront's rays differ in length and direction: the shadow rays of the bounces.  The camera paths' own shadow rays
    // (bounce 0) start on neighbouring pixels towards one light and stay one ray per lane (measured: -16 % when they stream too with the nested
    // loops of rounds 3-4; break-even, 8012-8080 against 8059-8115 Mrays/s, with the flat ones of round 5)
    if (cam.stream_run && bounce >= 1u) { // (a batch needs nothing special here: the queue entry carries the accumulator slot)
        const dim3 grid((ceil_div(p.capacity, kTraceBlock * cam.stream_run) + kShadowBuckets + 511u) & ~511u);
        // the two orders' launches follow each other on the stream (no rays of the other kind: the blocks return at once)
        if (count) {
            hipLaunchKernelGGL((k_shadow_stream<true, true>), grid, block, 0, s, cam, sc, p, bounce);
            hipLaunchKernelGGL((k_shadow_stream<true, false>), grid, block, 0, s, cam, sc, p, bounce);
        } else {
            hipLaunchKernelGGL((k_shadow_stream<false, true>), grid, block, 0, s, cam, sc, p, bounce);
            hipLaunchKernelGGL((k_shadow_stream<false, false>), grid, block, 0, s, cam, sc, p, bounce);
        }
        return;
    }
    const dim3 grid((ceil_div(p.capacity, kTraceBlock) + kShadowBuckets + 511u) & ~511u);
    if (cam.batch > 1) {
        if (count) hipLaunchKernelGGL((k_shadow<true, true>), grid, block, 0, s, cam, sc, p, bounce);
        else hipLaunchKernelGGL((k_shadow<false, true>), grid, block, 0, s, cam, sc, p, bounce);
    } else if (count) hipLaunchKernelGGL((k_shadow<true, false>), grid, block, 0, s, cam, sc, p, bounce);
    else hipLaunchKernelGGL((k_shadow<false, false>), grid, block, 0, s, cam, sc, p, bounce);
}
// bandwidth probe: 16 B per lane per access; every workgroup owns chunks of 4 x 256 elements, 4 loads in flight per lane before the first
// store, non-temporal on both sides (a copy streams: nothing is read again).  tools/probes/mem_probe.hip measured the variants on MI355X:
// grid-stride 1-deep 4.6-5.4 TB/s, 4-deep 5.0-5.6, 4-deep non-temporal with 16 384 workgroups 5.9 (hipMemcpy device-to-device: 4.7)
typedef float copy_v4f __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_f4(const float4* __restrict__ src4, float4* __restrict__ dst4, const uint64_t n)
{
    const copy_v4f* __restrict__ src = reinterpret_cast<const copy_v4f*>(src4);
    copy_v4f* __restrict__ dst = reinterpret_cast<copy_v4f*>(dst4);
    constexpr uint64_t kChunk = 4u * 256u;
    for (uint64_t base = (uint64_t)blockIdx.x * kChunk; base < n; base += (uint64_t)gridDim.x * kChunk) {
        copy_v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = base + (uint64_t)u * 256u + threadIdx.x;
            if (i < n) v[u] = __builtin_nontemporal_load(src + i);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint64_t i = base + (uint64_t)u * 256u + threadIdx.x;
            if (i < n) __builtin_nontemporal_store(v[u], dst + i);
        }
    }
}
// Issue-rate probe (rfw_hip_issue_probe): streams of vector instructions, 8 wavefronts per SIMD on every CU.
//   MIX 0 = v_fma_f32 alone (the instruction the guide's FP32 peak is quoted on): 32 per trip;
//   MIX 1 = one child of the PER-LANE node test of traverse_node.inc, instruction for instruction, twice per trip: 6 byte -> float
//           conversions, 3 packed FMAs, max3, min3, min, two compares = 14, i.e. 28 per trip (round 4 counted 32: ADVICE r04);
//   MIX 2 = one node step of the PACKET kernel (traverse_packet.h, slab4 + pushes): per child 6 v_fma_f32 with the plane as a scalar
//           operand, v_max3, v_min3, v_min, v_max and ONE v_cmp into a scalar register pair = 11, four children = 44 vector instructions per
//           trip, with the step's 27 scalar instructions issued beside them (they take issue slots too; only the vector ones are counted).
// Conversions, min / max, compares and packed FMAs issue at half the FMA rate or less (tools/probes/valu_peak.cpp lists them one by one).
// kIssueProbeVectorPerTrip is what the host multiplies by; tests/test_isa_budget.py counts the v_* instructions of each loop body in the ISA.
template <int MIX> __global__ __launch_bounds__(256) void k_issue_probe(float* out, const uint32_t trips, const float seed)
{
    typedef float v2 __attribute__((ext_vector_type(2)));
    float v[8];
    for (int i = 0; i < 8; i++) v[i] = seed + (float)(threadIdx.x + i);
    const float a = seed * 1.0001f, b = seed * 0.5f;
    const uint32_t u = fbits(seed) | 0x01020304u;
    v2 p[3] = {{v[0], v[1]}, {v[2], v[3]}, {v[4], v[5]}};
    const v2 pa = {a, a}, pb = {b, b};
    uint32_t sc0 = trips, sc1 = trips + 1u, sc2 = trips + 2u; // (MIX 2) scalar registers the scalar companions count in
    for (uint32_t it = 0; it < trips; it++) {
        if (MIX == 0) {
#pragma unroll
            for (int r = 0; r < 4; r++)
#pragma unroll
                for (int k = 0; k < 8; k++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
        } else if (MIX == 1) {
#pragma unroll
            for (int r = 0; r < 2; r++) { // 2 x 14 instructions per trip
                asm volatile("v_cvt_f32_ubyte0 %0, %6\n\tv_cvt_f32_ubyte1 %1, %6\n\tv_cvt_f32_ubyte2 %2, %6\n\tv_cvt_f32_ubyte3 %3, %6\n\tv_cvt_f32_ubyte0 %4, %6\n\tv_cvt_f32_ubyte1 %5, %6"
                             : "=v"(p[0].x), "=v"(p[0].y), "=v"(p[1].x), "=v"(p[1].y), "=v"(p[2].x), "=v"(p[2].y) : "v"(u));
                asm volatile("v_pk_fma_f32 %0, %0, %3, %4\n\tv_pk_fma_f32 %1, %1, %3, %4\n\tv_pk_fma_f32 %2, %2, %3, %4" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]) : "v"(pa), "v"(pb));
                asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %5, %6, %7" : "=v"(v[6]), "=v"(v[7]) : "v"(p[0].x), "v"(p[1].x), "v"(p[2].x), "v"(p[0].y), "v"(p[1].y), "v"(p[2].y));
                asm volatile("v_min_f32 %0, %0, %1" : "+v"(v[7]) : "v"(a));
                asm volatile("v_cmp_ge_f32 vcc, %0, %1\n\tv_cmp_ge_f32 vcc, %0, %2" : : "v"(v[7]), "v"(v[6]), "v"(b) : "vcc");
            }
        } else {
            // inv / b of the lane's ray in v[0..5] and a, b; the planes of the node in scalar registers (here: one, the probe measures issue, not values)
            const float plane = __builtin_amdgcn_readfirstlane(seed);
#pragma unroll
            for (int c = 0; c < 4; c++) { // 4 children x 11 vector instructions
                uint64_t m;
                asm volatile("v_fma_f32 %0, %6, %7, %8\n\tv_fma_f32 %1, %6, %8, %7\n\tv_fma_f32 %2, %6, %7, %7\n\tv_fma_f32 %3, %6, %8, %8\n\tv_fma_f32 %4, %6, %7, %8\n\tv_fma_f32 %5, %6, %8, %7"
                             : "=v"(v[0]), "=v"(v[1]), "=v"(v[2]), "=v"(v[3]), "=v"(v[4]), "=v"(v[5]) : "s"(plane), "v"(a), "v"(b));
                asm volatile("v_max3_f32 %0, %2, %3, %4\n\tv_min3_f32 %1, %5, %6, %7" : "=v"(v[6]), "=v"(v[7]) : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]));
                asm volatile("v_min_f32 %0, %0, %2\n\tv_max_f32 %1, %1, 0" : "+v"(v[7]), "+v"(v[6]) : "v"(a));
                asm volatile("v_cmp_ge_f32 %0, %1, %2" : "=s"(m) : "v"(v[7]), "v"(v[6]));
                // the scalar companions of a child: stack push (compare + add with carry), mask bookkeeping: 6 per child, 3 more per step below = 27
                asm volatile("s_cmp_lg_u64 %3, 0\n\ts_addc_u32 %0, %0, 0\n\ts_add_u32 %1, %1, 1\n\ts_and_b32 %2, %2, %1\n\ts_add_u32 %2, %2, %0\n\ts_lshl_b32 %1, %1, 1"
                             : "+s"(sc0), "+s"(sc1), "+s"(sc2) : "s"(m) : "scc");
            }
            asm volatile("s_add_u32 %0, %0, %1\n\ts_xor_b32 %1, %1, %2\n\ts_add_u32 %2, %2, 1" : "+s"(sc0), "+s"(sc1), "+s"(sc2) : : "scc");
        }
    }
    float r = p[0].x + p[1].y + p[2].x + (float)(sc0 ^ sc1 ^ sc2);
    for (int i = 0; i < 8; i++) r += v[i];
    out[blockIdx.x * 256u + threadIdx.x] = r;
}
// vector instructions per trip of the loop above (what rfw_hip_issue_probe multiplies by)
uint32_t issue_probe_vector_per_trip(int mix) { return mix == 0 ? 32u : (mix == 1 ? 28u : 44u); }
// (blocks of 256 threads, 8 per CU: every SIMD holds 8 wavefronts; `out` holds cus * 8 * 256 floats)
void launch_issue_probe(hipStream_t s, int mix, uint32_t cus, uint32_t trips, float* out)
{
    if (mix == 0) hipLaunchKernelGGL(k_issue_probe<0>, dim3(cus * 8u), dim3(256), 0, s, out, trips, 1.5f);
    else if (mix == 1) hipLaunchKernelGGL(k_issue_probe<1>, dim3(cus * 8u), dim3(256), 0, s, out, trips, 1.5f);
    else hipLaunchKernelGGL(k_issue_probe<2>, dim3(cus * 8u), dim3(256), 0, s, out, trips, 1.5f);
}
void launch_copy_f4(hipStream_t s, const float4* src, float4* dst, uint64_t n)
{
    if (n) hipLaunchKernelGGL(k_copy_f4, dim3(16384), dim3(256), 0, s, src, dst, n);
}
void launch_expand_nodes(hipStream_t s, const Node4Q* in, const OctantCopies& oc, uint32_t first, uint32_t n, const uint32_t* live)
{
    if (!n || !oc.quant) return;
    PacketNode* const wide = oc.wide ? oc.wide + first : nullptr; // (the packet form of the copies may be absent: offset only what exists — ADVICE r05)
    hipLaunchKernelGGL(k_expand_nodes, dim3(ceil_div((uint64_t)n * 8u, 256)), dim3(256), 0, s, in, wide, oc.quant + first, oc.stride, n, live);
}
void launch_quantize_nodes(hipStream_t s, const Node4* in, Node4Q* out, const OctantCopies& oc, uint32_t first, uint32_t n, const uint32_t* live)
{
    if (n) hipLaunchKernelGGL(k_quantize_nodes, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, out, n, live);
    launch_expand_nodes(s, out, oc, first, n, live);
}
void launch_quantize_regions(hipStream_t s, const Node4* in, Node4Q* out, const OctantCopies& oc, uint32_t n, const MeshRecord* recs, const uint32_t* counts, uint32_t n_recs)
{
    if (!n || !n_recs) return;
    hipLaunchKernelGGL(k_quantize_regions, dim3(ceil_div(n, 256)), dim3(256), 0, s, in, out, n, recs, counts, n_recs);
    if (oc.quant) hipLaunchKernelGGL(k_expand_regions, dim3(ceil_div((uint64_t)n * 8u, 256)), dim3(256), 0, s, out, oc.wide, oc.quant, oc.stride, n, recs, counts, n_recs);
}
void launch_assemble(hipStream_t s, const CameraParams& cam, const void* gathered, bool rgb, bool accumulator, uint64_t slab_elems, float4* frame,
                     uint32_t samples)
{
    const dim3 block(16, 4), grid(ceil_div(cam.width, 16), ceil_div(cam.height, 4), cam.batch > 1 ? cam.batch : 1u);
    if (rgb && accumulator) hipLaunchKernelGGL((k_assemble<true, true>), grid, block, 0, s, cam, gathered, slab_elems, frame, samples);
    else if (rgb) hipLaunchKernelGGL((k_assemble<true, false>), grid, block, 0, s, cam, gathered, slab_elems, frame, samples);
    else if (accumulator) hipLaunchKernelGGL((k_assemble<false, true>), grid, block, 0, s, cam, gathered, slab_elems, frame, samples);
    else hipLaunchKernelGGL((k_assemble<false, false>), grid, block, 0, s, cam, gathered, slab_elems, frame, samples);
}
// `narrow`: the destination is host memory written over the PCIe link: a few workgroups saturate the link, and more would only hold
// wave slots the trace kernels of the other frames in flight want
void launch_present(hipStream_t s, const float4* frame, uint32_t* bgra, uint64_t n, const float* steps255, bool narrow)
{
    const SrgbSteps st = make_steps(steps255);
    if (n) hipLaunchKernelGGL(k_present, dim3(narrow ? 64u : (unsigned)std::min<uint64_t>(ceil_div(n, 256), 8192)), dim3(256), 0, s, frame, bgra, n, st);
}
void launch_sum_batch(hipStream_t s, float4* acc_slabs, uint64_t slab_elems, uint32_t count)
{
    if (slab_elems && count > 1) hipLaunchKernelGGL(k_sum_batch, dim3((unsigned)ceil_div(slab_elems, 256)), dim3(256), 0, s, acc_slabs, slab_elems, count);
}
void launch_pack_finished(hipStream_t s, const float4* acc_slab, void* out, uint64_t n, uint32_t samples, uint32_t format, const float* steps255)
{
    if (!n) return;
    if (format == 1) hipLaunchKernelGGL(k_pack_f16, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, acc_slab, (_Float16*)out, n, samples);
    else hipLaunchKernelGGL(k_pack_bgra8, dim3((unsigned)std::min<uint64_t>(ceil_div(n, 256), 8192)), dim3(256), 0, s, acc_slab, (uint32_t*)out, n, samples, make_steps(steps255));
}
void launch_assemble_finished(hipStream_t s, const CameraParams& cam, const void* gathered, uint64_t slab_elems, uint32_t format, float4* frame, uint32_t* presented)
{
    const dim3 block(16, 4), grid(ceil_div(cam.width, 16), ceil_div(cam.height, 4), cam.batch > 1 ? cam.batch : 1u);
    if (format == 1) hipLaunchKernelGGL(k_assemble_f16, grid, block, 0, s, cam, (const _Float16*)gathered, slab_elems, frame);
    else hipLaunchKernelGGL(k_assemble_bgra8, grid, block, 0, s, cam, (const uint32_t*)gathered, slab_elems, presented);
}
// ---------------------------------------------------------------- peer-to-peer exchange (rfw_hip_p2p_*)
// The data travels as ordinary stores of the pack kernels into a peer's buffer; these two order it.  A kernel boundary on the sender's
// stream makes the packed tiles visible before the flag (the end of a kernel releases at system scope), the flag itself lives in uncached
// memory and is written / read with system-scope atomics, and the de-tiling kernel starts after the wait kernel has ended.
__global__ __launch_bounds__(64) void k_p2p_wait(const uint32_t* flags, const uint32_t first, const uint32_t count, const uint32_t want, const uint64_t limit_ticks,
                                                 uint32_t* timeout_flag)
{
    const uint32_t lane = threadIdx.x;
    if (lane >= count) return;
    const uint32_t* f = flags + first + lane;
    const uint64_t t0 = wall_clock64();
    for (;;) {
        const uint32_t v = __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((int32_t)(v - want) >= 0) return;
        if (wall_clock64() - t0 > limit_ticks) {
            __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}
__global__ __launch_bounds__(64) void k_p2p_signal(const P2PTargets targets, const uint32_t count, const uint32_t value)
{
    if (threadIdx.x < count) __hip_atomic_store(targets.p[threadIdx.x], value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
void launch_p2p_wait(hipStream_t s, const uint32_t* flags, uint32_t first, uint32_t count, uint32_t want, uint64_t limit_ticks, uint32_t* timeout_flag)
{
    if (count) hipLaunchKernelGGL(k_p2p_wait, dim3(1), dim3(64), 0, s, flags, first, count, want, limit_ticks, timeout_flag);
}
void launch_p2p_signal(hipStream_t s, const P2PTargets& targets, uint32_t count, uint32_t value)
{
    if (count) hipLaunchKernelGGL(k_p2p_signal, dim3(1), dim3(64), 0, s, targets, count, value);
}
void launch_pack_rgb(hipStream_t s, const float4* acc_slab, float* out, uint64_t n)
{
    if (n) hipLaunchKernelGGL(k_pack_rgb, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, s, acc_slab, out, n);
}

// ---------------------------------------------------------------- shading functions one by one (rfw_hip_debug_eval_shading)
// The device functions k_shade is made of, evaluated on caller-supplied inputs so that tests can hold each of them against an independent
// float64 formulation (tests/test_shading_kat.py).  Layout per case: 48 input floats, 12 output floats (include/rfw_hip.h).
__global__ __launch_bounds__(64) void k_eval_shading(const SceneDev sc, const CameraParams cam, const int op, const uint32_t n, const float* __restrict__ in,
                                                     float* __restrict__ out)
{
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const float* q = in + 48u * i;
    float* r = out + 12u * i;
    for (int k = 0; k < 12; k++) r[k] = 0.0f;
    ShadingData sd = extractParameters(reinterpret_cast<const rfw_device_material*>(q));
    prepare_tint(sd);
    const f3 N = mk3(q[24], q[25], q[26]), wo = mk3(q[27], q[28], q[29]), wi = mk3(q[30], q[31], q[32]);
    const f3 T = mk3(q[33], q[34], q[35]), B = mk3(q[36], q[37], q[38]);
    if (op == 0) {
        const f3 f = BSDFEval(sd, N, wo, wi, q[39], q[40] != 0.0f);
        r[0] = f.x; r[1] = f.y; r[2] = f.z;
    } else if (op == 1) {
        r[0] = BSDFPdf(sd, N, wo, wi);
    } else if (op == 2) {
        f3 w = mk3(0.0f);
        float pdf = 0.0f;
        int type = BSDF_TYPE_REFLECTED;
        BSDFSample(sd, T, B, N, wo, w, pdf, type, q[41], q[42]);
        r[0] = w.x; r[1] = w.y; r[2] = w.z; r[3] = pdf; r[4] = (float)type;
    } else if (op == 3) {
        r[0] = CalculateLightPDF(wo, q[39], q[43], N);
    } else if (op == 4) {
        LightView lv;
        lv.area = sc.area_lights; lv.point = sc.point_lights; lv.spot = sc.spot_lights; lv.directional = sc.directional_lights;
        lv.n_area = (int)cam.area_light_count; lv.n_point = (int)cam.point_light_count;
        lv.n_spot = (int)cam.spot_light_count; lv.n_directional = (int)cam.directional_light_count;
        float pick = 0.0f, lpdf = 0.0f;
        f3 col = mk3(0.0f);
        int picked = 0;
        const f3 P = RandomPointOnLight(lv, q[41], wo, N, pick, lpdf, col, picked);
        r[0] = P.x; r[1] = P.y; r[2] = P.z; r[3] = pick; r[4] = lpdf; r[5] = col.x; r[6] = col.y; r[7] = col.z; r[8] = (float)picked;
    } else if (op == 5) {
        const f3 b = RandomBarycentrics(q[41]);
        r[0] = b.x; r[1] = b.y; r[2] = b.z;
    }
}
void launch_eval_shading(hipStream_t s, const SceneDev& sc, const CameraParams& cam, int op, uint32_t n, const float* in, float* out)
{
    if (n) hipLaunchKernelGGL(k_eval_shading, dim3(ceil_div(n, 64)), dim3(64), 0, s, sc, cam, op, n, in, out);
}
void launch_query_closest(hipStream_t s, const SceneDev& sc, const float* origins, const float* directions, float t_min, float t_max, uint64_t n,
                          rfw_hip_hit* hits, uint32_t* depth)
{
    if (n == 0) return;
    const dim3 grid(ceil_div(n, kTraceBlock)), block(kTraceBlock);
    if (depth) hipLaunchKernelGGL(k_query_closest<true>, grid, block, 0, s, sc, origins, directions, t_min, t_max, n, hits, depth);
    else hipLaunchKernelGGL(k_query_closest<false>, grid, block, 0, s, sc, origins, directions, t_min, t_max, n, hits, depth);
}
void launch_query_any(hipStream_t s, const SceneDev& sc, const float* origins, const float* directions, float t_min, const float* t_max, uint64_t n,
                      uint8_t* occluded, uint32_t* depth)
{
    if (n == 0) return;
    if (depth) hipLaunchKernelGGL(k_query_any<true>, dim3(ceil_div(n, kTraceBlock)), dim3(kTraceBlock), 0, s, sc, origins, directions, t_min, t_max, n, occluded, depth);
    else hipLaunchKernelGGL(k_query_any<false>, dim3(ceil_div(n, kTraceBlock)), dim3(kTraceBlock), 0, s, sc, origins, directions, t_min, t_max, n, occluded, depth);
}

} // namespace rfwhip
