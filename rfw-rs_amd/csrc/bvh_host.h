// bvh_host.h — host-side (multi-threaded) binned-SAH builder producing the device Node4 layout.
// Replaces the reference's rtbvh calls (backends/gpu-rt/src/lib.rs:1345-1383 per-mesh refit_bvh,
// :1576-1581 BinnedSahBuilder + MBVH::construct for the TLAS).
#pragma once
#include <cstdint>
#include <vector>

#include "device_types.h"

namespace rfwhip {

struct PrimBox {
    float lo[3], hi[3];
};

struct HostBvh4 {
    std::vector<Node4> nodes;          // node 0 = root
    std::vector<uint32_t> prim_order;  // leaf-ordered primitive ids; leaf refs index into this
};

// Builds a 4-wide BVH over `boxes` (already padded by the caller).  max_leaf <= kMaxLeafTris.
// trav_cost: cost of one node step in units of one primitive test (SAH termination: split iff cost_split + trav_cost * area < cost_leaf)
void build_bvh4_host(const std::vector<PrimBox>& boxes, int max_leaf, int threads, HostBvh4& out, float trav_cost = 1.0f);

// Structural self-check used by the CPU tests: every primitive in exactly one leaf, child boxes contain their subtree.
uint64_t validate_bvh4(const HostBvh4& bvh, const std::vector<PrimBox>& boxes);

} // namespace rfwhip
