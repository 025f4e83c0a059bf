// lbvh.h — device-side BVH construction: Morton-ordered LBVH (Karras 2012) -> bottom-up fit -> 4-wide collapse,
// emitting the same Node4 layout the traversal kernels read.  Used for the TLAS every synchronize() (the
// reference rebuilds it on the CPU and re-uploads it, backends/gpu-rt/src/lib.rs:1576-1581,1617-1632) and,
// with builder = DEVICE_LBVH, for the per-mesh BLAS (the reference: rayon over rtbvh builds, :1345-1383).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_types.h"

namespace rfwhip {

// primitive bounds on the device: lo.xyz / hi.xyz (w unused)
struct alignas(16) DevBox {
    float lo[4], hi[4];
};

struct LbvhWorkspace {
    void* base = nullptr;
    size_t bytes = 0;
};

// bytes of scratch lbvh_build needs for n primitives (allocate once for the largest n)
size_t lbvh_workspace_bytes(uint32_t n);

// Builds a BVH4 over boxes[0..n) entirely on `stream`.  nodes_out needs room for max(n, 1) nodes; prim_order_out for n
// entries (leaf refs index into it; one primitive per leaf).  No host synchronisation.  Returns hipSuccess or the first error.
hipError_t lbvh_build(hipStream_t stream, const DevBox* boxes, uint32_t n, void* workspace, size_t workspace_bytes, Node4* nodes_out,
                      uint32_t* prim_order_out, uint32_t* node_count_out /* device, optional */);

// The TLAS of up to kTlasFusedMax instances in ONE launch (one workgroup: lbvh.hip, k_tlas_fused): instance boxes -> Morton sort -> hierarchy
// -> fit -> 4-wide nodes, and tlas_prims[k] = valid_gids[leaf order k].  The same tree, node for node, as launch_instance_boxes + lbvh_build +
// launch_gather_u32.  inst_boxes: room for n; workspace: lbvh_workspace_bytes(n).  hipErrorInvalidValue when n is outside [2, kTlasFusedMax].
constexpr uint32_t kTlasFusedMax = 16384;
hipError_t tlas_build_fused(hipStream_t s, const rfw_mat4* matrices, const uint32_t* mesh_of_instance, const DevBox* mesh_local_boxes, const uint32_t* valid_gids,
                            uint32_t n, void* workspace, size_t workspace_bytes, DevBox* inst_boxes, Node4* nodes_out, uint32_t* tlas_prims, uint32_t* node_count_out);

// Stress test of lbvh_build's fence-free bottom-up fit: `iterations` times n jittered boxes -> tree -> exact structural check, all on the
// stream (no host round trip).  nodes / order / seen: room for n; node_count: 1 word; result: 2 words, zeroed by the caller
// ([0] += mismatches, [1] += child boxes checked).
hipError_t lbvh_stress(hipStream_t s, uint32_t n, uint32_t iterations, uint32_t seed, void* workspace, size_t workspace_bytes, DevBox* boxes, Node4* nodes,
                       uint32_t* order, uint32_t* node_count, uint32_t* seen, unsigned long long* result);

// world-space boxes of instances: box k = local_aabb(mesh of gid[k]) through matrix gid[k], padded
void launch_instance_boxes(hipStream_t s, const rfw_mat4* matrices, const uint32_t* mesh_of_instance, const DevBox* mesh_local_boxes,
                           const uint32_t* valid_gids, uint32_t n_valid, DevBox* out);
// tlas_prims[k] = valid_gids[order[k]]
void launch_gather_u32(hipStream_t s, const uint32_t* src, const uint32_t* order, uint32_t n, uint32_t* dst);
// triangle boxes (padded) from the boundary's RTTriangle array
void launch_triangle_boxes(hipStream_t s, const rfw_rt_triangle* tris, uint32_t n, DevBox* out);
// the same from the 48-B heads of the records (vertex0 u0 | vertex1 u1 | vertex2 u2): all a builder needs of a triangle, and what a full
// build uploads FIRST so that the trees are built while the other 128 B per triangle are still on the bus
struct TriHead { float v[12]; };
static_assert(sizeof(TriHead) == 48, "TriHead = the first three float4 of rfw_rt_triangle");
void launch_triangle_boxes(hipStream_t s, const TriHead* heads, uint32_t n, DevBox* out);
// Spatial splits: boxes[piece.index] = the piece's box, padded like k_triangle_boxes pads (`pieces` = n x {index, lo[3], hi[3], pad}, 32 B each)
void launch_patch_boxes(hipStream_t s, const void* pieces, uint32_t n, DevBox* boxes);
// ... and the packets of duplicate records take the id of the triangle they duplicate: a packet whose mesh-local id is >= n_orig reads the
// original's mesh-local index from the bits of its own record's 16th float (the v0 texture coordinate: set_3d_mesh put it there)
void launch_resolve_duplicates(hipStream_t s, TriPacket* packets, uint32_t n, uint32_t id_offset, uint32_t n_orig, const rfw_rt_triangle* tris);
// leaf-ordered traversal packets: packet k = triangle order[k]; tri_id = order[k] + id_offset
void launch_make_packets(hipStream_t s, const rfw_rt_triangle* tris, const uint32_t* order, uint32_t n, uint32_t id_offset, TriPacket* out);

// SkinnedTriangles3D::apply (crates/rfw-backend/src/structs.rs:820-877) on the device: triangle i of `src` blended with the joint
// matrices selected by the joint data of its own three vertices (skin[3i..3i+2]); writes the deformed triangle to dst[i]
void launch_skin_triangles(hipStream_t s, const rfw_rt_triangle* src, const rfw_joint_data* skin, const rfw_mat4* joints, uint32_t n_joints,
                           uint32_t n_tris, rfw_rt_triangle* dst);
// bounds of n triangles' vertices -> out (one DevBox); scratch = 6 uint32
void launch_mesh_bounds(hipStream_t s, const rfw_rt_triangle* tris, uint32_t n, uint32_t* scratch, DevBox* out);

// (key, value) radix sort of n 32-bit pairs on `s` (hipCUB); bits [0, end_bit) of the keys take part
size_t sort_pairs_workspace_bytes(uint32_t n);
hipError_t sort_pairs_u32(hipStream_t s, void* workspace, size_t workspace_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                          uint32_t* vals_out, uint32_t n, int end_bit);

} // namespace rfwhip
