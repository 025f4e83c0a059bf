// sah_build.h — binned-SAH BVH construction on the device (sah_build.hip): the quality of the host builder (bvh_host.cpp) at
// device speed, for static meshes and for meshes rebuilt every frame alike.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "device_types.h"
#include "lbvh.h" // DevBox

namespace rfwhip {

// bytes of scratch sah_build needs for n primitives
size_t sah_workspace_bytes(uint32_t n);

// Builds a BVH4 over boxes[0..n) on `stream`.  nodes_out needs room for max(n, 1) nodes, prim_order_out for n entries (leaf refs
// index into it).  Synchronises the stream ONCE (a 16-byte read-back after the levels of the upper tree), so it is a blocking call.
// max_leaf <= kMaxLeafTris; trav_cost as in build_bvh4_host.
hipError_t sah_build(hipStream_t stream, const DevBox* boxes, uint32_t n, void* workspace, size_t workspace_bytes, Node4* nodes_out,
                     uint32_t* prim_order_out, uint32_t* node_count_out /* device, optional */, int max_leaf, float trav_cost);

// Many meshes in ONE build (a scene of many small meshes is all launch latency when each is built alone).  boxes[0..n): the meshes' primitive
// boxes one mesh after the other; trees[m] (device, sorted by `first`, ranges disjoint and covering [0, n)): the mesh's range and the first node
// of its region in nodes_base.  Per tree: nodes at nodes_base + node_base (root = 0, child indices relative to the region, leaf ranges
// relative to the tree's first primitive), node_counts_out[m] = its node count.  prim_order_out[0 .. n) = leaf-ordered primitive POSITIONS of
// the whole forest (what one packet-building launch over all meshes wants); launch_forest_relative_order then makes every tree's part
// relative to its `first` (what a per-mesh consumer wants).  largest_tree = max count (host-known).  Blocks like sah_build (one 16-byte read-back).
struct ForestTree { uint32_t first, count, node_base, pad; };
size_t sah_forest_workspace_bytes(uint32_t n, uint32_t n_trees);
hipError_t sah_build_forest(hipStream_t stream, const DevBox* boxes, uint32_t n, const ForestTree* trees, uint32_t n_trees, uint32_t largest_tree, void* workspace,
                            size_t workspace_bytes, Node4* nodes_base, uint32_t* prim_order_out, uint32_t* node_counts_out, int max_leaf, float trav_cost);
void launch_forest_relative_order(hipStream_t stream, uint32_t* order, uint32_t n, const ForestTree* trees, uint32_t n_trees);

// ---- refit: new boxes for a tree whose topology stays (skinned meshes: the triangles move every frame)
// parent_slot[i] = 4 * parent + child slot of node i (0xffffffff for the root), n_internal[i] = interior children of node i
void launch_refit_setup(hipStream_t s, const Node4* nodes, uint32_t n_nodes, uint32_t* parent_slot, uint32_t* n_internal);
// leaf boxes from the (moved) triangles, then bottom-up through the interior nodes; `arrive` is n_nodes words of scratch
void launch_refit(hipStream_t s, Node4* nodes, uint32_t n_nodes, const rfw_rt_triangle* tris, const uint32_t* order, const uint32_t* parent_slot,
                  const uint32_t* n_internal, uint32_t* arrive);

} // namespace rfwhip
