// kernels.h — launch wrappers of the HIP kernels (defined in kernels.hip), callable from the C-ABI host code.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/rfw_hip.h"
#include "device_types.h"

namespace rfwhip {

struct SceneDev {
    const Node4Q* tlas_nodes;
    const uint32_t* tlas_prims;
    const InstanceXform* instances;
    const InstanceNormal* instance_normals;
    const MeshRecord* meshes;         // per mesh record: where it lives, and the triangle-id offset the boundary reports
    const Node4Q* blas_nodes;
    // the same trees as the packet kernels read them (traverse_packet.h: a node is fetched once per wavefront through the scalar cache):
    // eight copies, one per ray octant, child boxes as floats, near / far planes picked and children sorted for the octant (PacketNode);
    // nullptr = not available
    const PacketNode* tlas_wide;
    const PacketNode* blas_wide;
    uint32_t tlas_wide_stride, blas_wide_stride; // nodes per octant copy
    // ... and as the one-ray-per-lane kernels read them (traverse.h): the same eight copies, still quantised (make_octant_node), same strides
    const Node4Q* tlas_oct;
    const Node4Q* blas_oct;
    const TriPacket* tri_packets;
    const rfw_rt_triangle* triangles; // shading attributes, global triangle id order
    const rfw_device_material* materials;
    const rfw_area_light* area_lights;
    const rfw_point_light* point_lights;
    const rfw_spot_light* spot_lights;
    const rfw_directional_light* directional_lights;
    const uint32_t* tex_data;   // all textures' texels (set_textures), then the skybox's
    const TexDesc* tex_desc;
    uint32_t n_textures;
    TexDesc skybox;             // mips == 0: no skybox image, a miss adds the constant sky colour
    const uint8_t* blue_noise;  // kBlueNoiseWords table entries (each a byte), or nullptr: every sample draws from xorshift
    uint32_t* spill;
    uint32_t spill_stride;
    uint32_t spill_rows;        // <= kStackSpill
    uint32_t* overflow_flag;    // pinned host word, device-visible
    QueueCounters* counters;
    // the counter block of the NEXT frame on this instance's stream (the other half of a ring of two): k_primary clears it, so that no frame starts
    // with a clear of its own — a 512-byte fill is a dispatch like any other, and with frames in flight it queued behind the trace kernels'
    // wavefronts for up to 0.76 ms before its frame's first kernel could start (kernel timeline, round 5)
    QueueCounters* counters_next;
};

// Wavefront state, structure-of-arrays of 16-B elements (the reference's 64-B AoS PathState, structs.glsl:4-9, split):
//   ray_o  = (origin.xyz, bits(path id))        ray_d = (direction.xyz, bits(packed previous normal))
//   thr    = (throughput.rgb, postponed bsdf pdf)  hit = (inst, tri, bits(t), bary 16:16)
// two halves each (ping-pong on path_length % 2, shade.comp:77-81); shadow queue = PotentialContribution (structs.glsl:172-176)
struct PathDev {
    float4* ray_o[2];
    float4* ray_d[2];
    float4* thr[2];
    uint4* hit[2];
    float4* sh_o; // (origin.xyz, bits(pixel))
    float4* sh_d; // (direction.xyz, distance)
    float4* sh_e; // (contribution.rgb, 0)
    float4* acc;  // accumulator in slab order (this rank's tiles)
    uint32_t capacity; // local paths = local_tiles * tile_size^2
};

void launch_prepare_instances(hipStream_t s, const rfw_mat4* matrices, const uint32_t* mesh_of_instance, const MeshRecord* meshes, uint32_t n,
                              InstanceXform* xf, InstanceNormal* nm);
void launch_primary(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, bool count);
void launch_primary_batch(hipStream_t s, const CameraParams& cam, const BatchViews& views, const SceneDev& sc, const PathDev& p, bool count);
void launch_extend(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, uint32_t bounce, bool count, const uint32_t* order = nullptr);
void launch_extension_keys(hipStream_t s, const SceneDev& sc, const PathDev& p, uint32_t bounce, uint32_t* keys, uint32_t* vals);
void launch_shade(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, uint32_t bounce);
void launch_shadow(hipStream_t s, const CameraParams& cam, const SceneDev& sc, const PathDev& p, uint32_t bounce, bool count);
void launch_copy_f4(hipStream_t s, const float4* src, float4* dst, uint64_t n); // bandwidth probe
void launch_issue_probe(hipStream_t s, int mix, uint32_t cus, uint32_t trips, float* out); // issue-rate probe (mix 0, 1, 2: kernels.hip)
uint32_t issue_probe_vector_per_trip(int mix);                                                // vector instructions per trip and wavefront of that mix
// The eight per-octant copies of a node array (PacketNode for the packet kernels, Node4Q for the one-ray-per-lane kernels): copy `oct` of
// node i at [oct * stride + i] of both
struct OctantCopies {
    PacketNode* wide = nullptr;
    Node4Q* quant = nullptr;
    uint32_t stride = 0;
};
// out[i] = quantised in[i] and its octant copies at oc[first + i], i < n.  `live` (nullable): the tree's node count ON THE DEVICE — slots behind
// it are skipped (regions are sized for the worst case, one node per primitive)
void launch_quantize_nodes(hipStream_t s, const Node4* in, Node4Q* out, const OctantCopies& oc, uint32_t first, uint32_t n, const uint32_t* live = nullptr);
// behind tlas_build_fused (lbvh.h): quantised nodes + their per-octant copies (the first *live of n_nodes_max) and the instance descriptors, one launch
void launch_tlas_finish(hipStream_t s, const Node4* raw, Node4Q* out, const OctantCopies& oc, uint32_t n_nodes_max, const uint32_t* live, const rfw_mat4* matrices,
                        const uint32_t* mesh_of_instance, const MeshRecord* meshes, uint32_t n_instances, InstanceXform* xf, InstanceNormal* nm);
void launch_expand_nodes(hipStream_t s, const Node4Q* in, const OctantCopies& oc, uint32_t first, uint32_t n, const uint32_t* live = nullptr); // the copies of already quantised nodes
// the same for slots [0, n) holding SEVERAL trees: record k's tree lives at recs[k].node_base and has counts[k] nodes (both on the device)
void launch_quantize_regions(hipStream_t s, const Node4* in, Node4Q* out, const OctantCopies& oc, uint32_t n, const MeshRecord* recs, const uint32_t* counts, uint32_t n_recs);
void launch_assemble(hipStream_t s, const CameraParams& cam, const void* gathered, bool rgb, bool accumulator, uint64_t slab_elems, float4* frame,
                     uint32_t samples);
void launch_sum_batch(hipStream_t s, float4* acc_slabs, uint64_t slab_elems, uint32_t count); // slab 0 += slabs 1 .. count - 1, in order
void launch_pack_rgb(hipStream_t s, const float4* acc_slab, float* out, uint64_t n);
// the FINISHED frame of a slab for the all-gather (format 1: three halves per pixel; 2: presented B, G, R, A bytes) and its de-tiling
void launch_pack_finished(hipStream_t s, const float4* acc_slab, void* out, uint64_t n, uint32_t samples, uint32_t format, const float* steps255);
// peer-to-peer exchange: flag words in uncached memory.  wait: lanes first .. first + count - 1 of ONE wavefront poll flags[lane] until it has
// reached `want` (wrap-safe), at most `limit_ticks` of the 100 MHz wall clock, then *timeout_flag = 1.  signal: *targets.p[i] = value.
struct P2PTargets { uint32_t* p[16]; };
void launch_p2p_wait(hipStream_t s, const uint32_t* flags, uint32_t first, uint32_t count, uint32_t want, uint64_t limit_ticks, uint32_t* timeout_flag);
void launch_p2p_signal(hipStream_t s, const P2PTargets& targets, uint32_t count, uint32_t value);
void launch_assemble_finished(hipStream_t s, const CameraParams& cam, const void* gathered, uint64_t slab_elems, uint32_t format, float4* frame, uint32_t* presented);
void launch_present(hipStream_t s, const float4* frame, uint32_t* bgra, uint64_t n, const float* steps255, bool narrow);
void launch_eval_shading(hipStream_t s, const SceneDev& sc, const CameraParams& cam, int op, uint32_t n, const float* in, float* out);
void launch_query_closest(hipStream_t s, const SceneDev& sc, const float* origins, const float* directions, float t_min, float t_max, uint64_t n,
                          rfw_hip_hit* hits, uint32_t* depth = nullptr /* optional: nodes visited per ray */);
void launch_query_any(hipStream_t s, const SceneDev& sc, const float* origins, const float* directions, float t_min, const float* t_max, uint64_t n,
                      uint8_t* occluded, uint32_t* depth = nullptr /* optional: nodes visited per ray */);

} // namespace rfwhip
