// device_types.h — layout of everything that lives in HBM (shared by host code and kernels).
//
// All buffers are structure-of-arrays of 16-byte elements so that a wavefront reading element
// `lane` of an array issues one coalesced dwordx4 load (1 KiB per wave-instruction).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

#include "../../include/rfw_pod.h"

namespace rfwhip {

// 4-wide BVH node, 128 B = one cache line, 128-B aligned.  Child boxes SoA across the 4 children.
// child[i]: kInvalidRef = empty slot; bit31 set = leaf: (count-1) << 27 | first  (first: index into the
// leaf-ordered triangle packets of the mesh, or into the TLAS instance-index list); else interior node index
// (relative to the owning BVH's node base).
// (alignas(16) on the node and box types: struct copies then move as 16-byte accesses — with the natural alignment of 4 a 128-B node store
// is 32 dword stores, each touching 64 lines per wavefront)
struct alignas(16) Node4 {
    float lox[4], hix[4], loy[4], hiy[4], loz[4], hiz[4];
    uint32_t child[4];
    uint32_t pad[4];
};
static_assert(sizeof(Node4) == 128, "Node4 must be one 128-B line");
constexpr uint32_t kInvalidRef = 0xffffffffu;
constexpr uint32_t kLeafBit = 0x80000000u;
constexpr uint32_t kNoPath = 0xfffffffeu; // hit.inst of a slab slot that holds no pixel (a miss is -1)
constexpr uint32_t kLeafFirstMask = 0x07ffffffu;
constexpr int kMaxLeafTris = 8;
inline __host__ __device__ uint32_t make_leaf(uint32_t first, uint32_t count) { return kLeafBit | ((count - 1u) << 27) | first; }

// The node the traversal kernels actually read: the same 4 child boxes quantised to 8 bits per plane relative to the
// node's own origin with power-of-two scales (floor for lo, ceil for hi => conservative), 64 B = 4 dwordx4 per lane
// instead of 7.  The trace kernels are bound by the L1/TA data-return path (16 cycles per dwordx4 wave-instruction,
// profiles/), so bytes per node visit are what matters.  plane = origin + q * scale, scale = a power of two per axis, stored as the
// float itself (the kernels are instruction-bound: an exponent byte would cost a shift and a mask per axis and visit to unpack).
struct alignas(16) Node4Q {
    float ox, oy, oz;
    float sx;           // scale of axis x
    uint32_t qlo[3];    // byte i of qlo[a] = quantised lower plane of child i on axis a
    uint32_t qhi[3];
    float sy, sz;       // scales of axes y, z
    uint32_t child[4];  // as Node4::child
};
static_assert(sizeof(Node4Q) == 64, "Node4Q must be half a 128-B line");

inline __host__ __device__ uint32_t rfw_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline __host__ __device__ float rfw_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

inline __host__ __device__ Node4Q quantize_node(const Node4& n)
{
    Node4Q q;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    const float* nlo[3] = {n.lox, n.loy, n.loz};
    const float* nhi[3] = {n.hix, n.hiy, n.hiz};
    for (int i = 0; i < 4; i++) {
        if (n.child[i] == 0xffffffffu) continue;
        for (int a = 0; a < 3; a++) {
            lo[a] = nlo[a][i] < lo[a] ? nlo[a][i] : lo[a];
            hi[a] = nhi[a][i] > hi[a] ? nhi[a][i] : hi[a];
        }
    }
    float scale[3];
    for (int a = 0; a < 3; a++) {
        if (!(hi[a] >= lo[a])) { lo[a] = 0.0f; hi[a] = 0.0f; } // node without children
        // smallest power of two s with 255 * s >= extent (biased exponent clamped to the normal range)
        const float want = (hi[a] - lo[a]) * (1.0f / 255.0f);
        uint32_t e = (rfw_f2bits(want) >> 23) & 0xffu;
        if ((rfw_f2bits(want) & 0x007fffffu) != 0u) e += 1u;
        if (e < 1u) e = 1u;
        if (e > 254u) e = 254u;
        scale[a] = rfw_bits2f(e << 23);
        if (255.0f * scale[a] < (hi[a] - lo[a]) && e < 254u) { e += 1u; scale[a] = rfw_bits2f(e << 23); }
    }
    q.ox = lo[0]; q.oy = lo[1]; q.oz = lo[2];
    q.sx = scale[0]; q.sy = scale[1]; q.sz = scale[2];
    for (int a = 0; a < 3; a++) { q.qlo[a] = 0u; q.qhi[a] = 0u; }
    for (int i = 0; i < 4; i++) {
        q.child[i] = n.child[i];
        for (int a = 0; a < 3; a++) {
            uint32_t ql = 255u, qh = 0u; // empty slot: inverted
            if (n.child[i] != 0xffffffffu) {
                const float inv = 1.0f / scale[a];
                float fl = floorf((nlo[a][i] - lo[a]) * inv), fh = ceilf((nhi[a][i] - lo[a]) * inv);
                fl = fl < 0.0f ? 0.0f : (fl > 255.0f ? 255.0f : fl);
                fh = fh < 0.0f ? 0.0f : (fh > 255.0f ? 255.0f : fh);
                ql = (uint32_t)fl;
                qh = (uint32_t)fh;
                while (ql > 0u && lo[a] + (float)ql * scale[a] > nlo[a][i]) ql--;   // decoded planes must enclose the original box
                while (qh < 255u && lo[a] + (float)qh * scale[a] < nhi[a][i]) qh++;
            }
            q.qlo[a] |= ql << (8 * i);
            q.qhi[a] |= qh << (8 * i);
        }
    }
    return q;
}

// A BVH4 node as the PACKET kernels read it (traverse_packet.h): ONE OF EIGHT copies, the one for a ray octant (bit a: direction component
// a negative) — the near and far plane of every axis already picked by the octant's signs, the children sorted front to back along the
// octant's diagonal (the order every packet of that octant visits them in: no per-node sorting at run time).  Copy `oct` of node i lives at
// [oct * stride + i] of the array.  Made from the quantised node: the boxes are the ones the 64-B node encodes, as floats (plane = origin +
// q * scale, pushed outwards by an ulp for the rounding of the add): conservative like those, and no result depends on the boxes.
// An empty slot gets the box (+inf, -inf) — every ray's entry distance is +inf, so it needs no test of its own — and keeps kInvalidRef.
struct alignas(16) PacketNode {
    float nx[4], ny[4], nz[4], fx[4];
    float fy[4], fz[4];
    uint32_t child[4], pad[4];
};
static_assert(sizeof(PacketNode) == 128, "PacketNode");
constexpr uint64_t kPacketNodeCopies = 8;

// The decoded child boxes of a quantised node (floats; empty slots (+inf, -inf)) and the order of its four slots for ray octant `oct`:
// front to back along the octant's diagonal — by where a plane swept along (+-1, +-1, +-1) meets each box first — stable, empty slots last.
inline __host__ __device__ void octant_order(const Node4Q& q, const uint32_t oct, float (&lo)[3][4], float (&hi)[3][4], int (&ord)[4])
{
    const float o[3] = {q.ox, q.oy, q.oz}, sc[3] = {q.sx, q.sy, q.sz};
    float key[4];
    for (int i = 0; i < 4; i++) {
        key[i] = 0.0f;
        for (int a = 0; a < 3; a++) {
            if (q.child[i] == 0xffffffffu) {
                lo[a][i] = INFINITY;
                hi[a][i] = -INFINITY;
                continue;
            }
            const float l = o[a] + (float)((q.qlo[a] >> (8 * i)) & 0xffu) * sc[a], h = o[a] + (float)((q.qhi[a] >> (8 * i)) & 0xffu) * sc[a];
            lo[a][i] = l - 1.1920929e-7f * (l < 0.0f ? -l : l); // (the rounding of the add: the float box must enclose the exact decoded box)
            hi[a][i] = h + 1.1920929e-7f * (h < 0.0f ? -h : h);
            key[i] += ((oct >> a) & 1u) ? -hi[a][i] : lo[a][i];
        }
        if (q.child[i] == 0xffffffffu) key[i] = INFINITY;
    }
    for (int i = 0; i < 4; i++) ord[i] = i;
    for (int i = 1; i < 4; i++)
        for (int j = i; j > 0 && key[ord[j]] < key[ord[j - 1]]; j--) { const int t_ = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t_; }
}
inline __host__ __device__ PacketNode make_packet_node(const Node4Q& q, const uint32_t oct)
{
    float lo[3][4], hi[3][4];
    int ord[4];
    octant_order(q, oct, lo, hi, ord);
    PacketNode n;
    for (int k = 0; k < 4; k++) {
        const int i = ord[k];
        n.nx[k] = (oct & 1u) ? hi[0][i] : lo[0][i]; n.fx[k] = (oct & 1u) ? lo[0][i] : hi[0][i];
        n.ny[k] = (oct & 2u) ? hi[1][i] : lo[1][i]; n.fy[k] = (oct & 2u) ? lo[1][i] : hi[1][i];
        n.nz[k] = (oct & 4u) ? hi[2][i] : lo[2][i]; n.fz[k] = (oct & 4u) ? lo[2][i] : hi[2][i];
        n.child[k] = q.child[i];
        n.pad[k] = 0u;
    }
    return n;
}
// The same copy for the ONE-RAY-PER-LANE kernels (traverse.h), still quantised (64 B: a lane fetches its own node): in copy `oct`, qlo[a]
// holds the planes a ray of that octant ENTERS through (the upper ones where its direction component is negative), qhi[a] the ones it
// leaves through, and the children are in the octant's front-to-back order — so a visit needs neither the six per-plane selects nor the
// sorting network: hit children go on the lane's stack in the stored order.  Origin and scales are the node's own.
inline __host__ __device__ Node4Q make_octant_node(const Node4Q& q, const uint32_t oct)
{
    float lo[3][4], hi[3][4];
    int ord[4];
    octant_order(q, oct, lo, hi, ord);
    Node4Q n = q;
    for (int a = 0; a < 3; a++) { n.qlo[a] = 0u; n.qhi[a] = 0u; }
    for (int k = 0; k < 4; k++) {
        const int i = ord[k];
        n.child[k] = q.child[i];
        for (int a = 0; a < 3; a++) {
            const uint32_t l = (q.qlo[a] >> (8 * i)) & 0xffu, h = (q.qhi[a] >> (8 * i)) & 0xffu;
            const bool neg = ((oct >> a) & 1u) != 0u;
            n.qlo[a] |= (neg ? h : l) << (8 * k);
            n.qhi[a] |= (neg ? l : h) << (8 * k);
        }
    }
    return n;
}

// Triangle packet for traversal, 48 B, stored in BLAS leaf order (no index indirection in the leaf loop):
//   p0 = (v0.xyz, bits(global triangle id)), p1 = (edge1.xyz, 1/dot(gn,gn)), p2 = (edge2.xyz, 0)
// edge1 = v1 - v0 and edge2 = v2 - v0 are the very subtractions intersection.glsl:7-8 performs per test, done once.
// The reference's leaf loop gathers the whole 176-B RTTriangle (ray_gen.comp:230-233).
struct TriPacket {
    float v0x, v0y, v0z; uint32_t tri_id;
    float e1x, e1y, e1z; float inv_gn2;
    float e2x, e2y, e2z; float pad;
};
static_assert(sizeof(TriPacket) == 48, "TriPacket");

// Per-instance record used by traversal (64 B): rows of the inverse matrix + where the mesh's BLAS lives.
struct alignas(16) InstanceXform {
    float inv_r0[4], inv_r1[4], inv_r2[4]; // row i = (m[i], m[4+i], m[8+i], m[12+i]) of the column-major inverse
    uint32_t node_base;                    // first node of the mesh's BLAS in blas_nodes
    uint32_t tri_base;                     // first packet of the mesh in tri_packets
    uint32_t flags;                        // bit0: valid; bit 1 (kInstanceIdentity): the inverse is exactly the identity
    uint32_t mesh;
};
static_assert(sizeof(InstanceXform) == 64, "InstanceXform");
constexpr uint32_t kInstanceIdentity = 2u;

// Per-instance record used by shade (48 B): rows of the normal matrix transpose(inverse(M)).
struct alignas(16) InstanceNormal {
    float n_r0[4], n_r1[4], n_r2[4];
};

// A texture as handed over by set_textures / set_skybox (crates/rfw-backend/src/structs.rs:69-121): texels of all mip levels
// back to back in one device array of 4-byte texels.
struct TexDesc {
    uint32_t offset;  // first texel of level 0 in tex_data
    uint32_t w, h, mips;
    uint32_t format;  // RFW_FORMAT_BGRA8 / RFW_FORMAT_RGBA8
    uint32_t pad[3];
};
static_assert(sizeof(TexDesc) == 32, "TexDesc");

struct MeshRecord {
    uint32_t node_base, node_count;
    uint32_t tri_base, tri_count; // tri_base = first triangle / packet of the mesh in the mega-buffers = offset of the ids the packets carry
    // The triangle id the boundary reports (rfw_hip_hit.tri): offset of the mesh in the concatenation of all meshes in mesh-id order, then
    // the skinned copies (gpu-rt's numbering, ray_gen.comp:231).  After a full build it equals tri_base; after an incremental synchronize
    // (one mesh rebuilt in place or appended behind the others) the storage order differs and the query kernels translate.
    uint32_t tri_logical;
    uint32_t pad[3];
};
static_assert(sizeof(MeshRecord) == 32, "MeshRecord");

// Camera block handed to the kernels (CameraData of backends/gpu-rt/src/lib.rs:147-175 without the queue counters).
struct CameraParams {
    float pos[3]; float lens_size;
    float right[3]; float spread_angle;
    float up[3]; float clamp_value;
    float p1[3]; uint32_t tile_shift;   // log2(tile_size), or 0xffffffff when the tile size is not a power of two
    uint32_t width, height, sample_count, width_magic; // (x / width == __umulhi(x, width_magic) for every pixel index; 0: divide.  api_frame.cpp index_magic)
    uint32_t point_light_count, area_light_count, spot_light_count, directional_light_count;
    // shard description: tiles of tile_size^2 pixels dealt round-robin to `world` ranks
    uint32_t tile_size, tiles_x, tiles_y, rank;
    uint32_t world, local_tiles, flags, max_path_length;
    float sky[3]; uint32_t tiles_x_magic; // tile / tiles_x, as width_magic
    // a batch of independent frames traced as ONE tall virtual frame (rfw_hip_render_batch): frame f owns paths
    // [f * frame_capacity, (f + 1) * frame_capacity); a path carries f in the top byte of its path-id word
    // streaming trace kernels (traverse.h, traverse_stream): a wavefront owns stream_run x 64 consecutive queue entries and hands a new ray to its idle
    // lanes whenever stream_refill of them are idle; stream_run = 0: one ray per lane, the wavefront lasts as long as its longest ray
    uint32_t batch, frame_capacity, stream_run, stream_refill;
    // sample index of every frame of a batch (seeds the RNG / indexes the blue-noise sequence): 0 for rfw_hip_render_batch's new images,
    // first_sample + f for rfw_hip_render_samples
    uint32_t batch_sample[16];
};

constexpr int kMaxBatch = 16;
constexpr uint32_t kBlueNoiseWords = 5u * 65536u; // gpu_rt::blue_noise::create_blue_noise_buffer(): Sobol bytes, scrambling tile, ranking tile
struct FrameView { // the view-dependent part of CameraParams, one per frame of a batch
    float pos[3]; float lens_size;
    float right[3]; float pad0;
    float up[3]; float pad1;
    float p1[3]; float pad2;
};
struct BatchViews {
    FrameView v[kMaxBatch];
};

// Device-side queue counters; one slot per bounce so nothing has to be reset or read back between bounces
// (the reference reads extensionId/shadowId back to the host every bounce: gpu-rt/src/lib.rs:2052-2069).
// shadow rays are queued per bucket: every directional light's rays in the LAST bucket (they are traced far to near), the positional lights
// dealt over the other seven (picked light % 7) — a wavefront's rays aim at one light (or one kind of light).  (16 buckets — light x 4 classes
// of the surface orientation — were measured in round 3: EXPERIMENTS.md)
constexpr int kShadowBuckets = 8;
struct QueueCounters {
    uint32_t ext[8];
    uint32_t shadow[8][kShadowBuckets]; // [bounce][bucket]
    unsigned long long trav[3][3]; // [kind: 0 primary, 1 extend, 2 shadow][0 nodes visited, 1 triangles tested, 2 instances entered]
    unsigned long long overflow, pad;
    unsigned long long wave_max_nodes[3]; // COUNT mode: sum over wavefronts of the largest per-lane node count (lane utilisation = nodes / (64 * this))
    unsigned long long wave_exec[3][2];   // COUNT mode: wavefront-level executions of the node test / of the triangle test
    unsigned long long max_nodes[3];      // COUNT mode: the largest per-ray node count
    unsigned long long pad2;
    unsigned long long wave_uniform[3];   // COUNT mode: node-test executions in which every active lane visited the SAME node
    unsigned long long pad3;
};

enum : uint32_t { kFlagNoNee = 1u, kFlagCount = 2u, kFlagNearFirstDirectional = 4u, kFlagFarFirstPositional = 8u,
                  kFlagPacketPrimary = 16u, kFlagPacketShadow = 32u, // option "packet_trace": which rays walk the tree as wavefront packets (traverse_packet.h)
                  kFlagPacketShadowFar = 64u, kFlagShadeSmallGroups = 128u }; // (kFlagShadeSmallGroups: k_shade in workgroups of 256 instead of 512, option "shade_group") // ... bit 2: only the camera paths' shadow rays of the buckets traced far to near (the directional lights' bucket: parallel rays)
// (kFlagNearFirstDirectional, kFlagFarFirstPositional: option "shadow_order")


} // namespace rfwhip
