// env_switches.h — every environment switch of the library, in one place.  They are read when an instance is created (rfw_hip_create ->
// read_env_switches) and nowhere else: no call of the hot or the build path asks the environment anything.  What is left are switches tests,
// tools/scale.sh and diagnosis use; the A/B switches of rounds 1-5 went with their experiments (experiments/, EXPERIMENTS.md).
#pragma once
#include <cstdint>

namespace rfwhip {
struct EnvSwitches {
    bool build_trace = false;               // RFW_BUILD_TRACE: stderr lines saying where the host time of a build goes
    bool no_forest = false;                 // RFW_NO_FOREST (tests): the meshes of a full build are built one by one instead of as one forest
    bool lbvh_fenced = false;               // RFW_LBVH_FENCED=1 (tests): the bottom-up fit with device-scope fences instead of the fence-free form
    bool p2p_data_cached = false;           // RFW_P2P_DATA_CACHED (tools/scale.sh): the peer-store receive buffer in ordinary instead of fine-grained memory
    bool p2p_flags_finegrained = false;     // RFW_P2P_FLAGS_FINEGRAINED (tools/scale.sh): the flag words in the fall-back kind of memory
    uint64_t packet_auto_max_triangles = 0; // RFW_PACKET_AUTO_MAX_TRIANGLES (tests): moves kPacketAutoMaxTriangles; 0 = not set
    bool has_spatial_splits = false;        // RFW_SPATIAL_SPLITS: the default of option "spatial_splits"
    float spatial_splits = 0.0f;
    int node_order = 0;                     // RFW_NODE_ORDER (experiment, round 6): static BLAS nodes renumbered on the host after a full build — 1 depth-first, 2 treelets of <= 32 nodes
    int packet_trace = -1;                  // RFW_PACKET_TRACE: the default of option "packet_trace"; -1 = not set
};
const EnvSwitches& env_switches(); // (api_frame.cpp)
void read_env_switches();
} // namespace rfwhip
