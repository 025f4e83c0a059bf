// lbvh_emu.cpp — TEST INFRASTRUCTURE: rfw-rs_amd/csrc/lbvh.hip — the file — compiled as C++ against tests/emu/fake_hip and run under wave_emu.h.
//   lbvh_emu stress <boxes> <iterations> <seed>   the library's own stress test (lbvh_stress: jittered boxes -> lbvh_build -> the device-side structural
//                                                 check of every child box and every primitive) on the CPU; prints "OK checked=<n>"
//   lbvh_emu tlas <instances> <seed>              the TLAS two ways: launch_instance_boxes + lbvh_build + launch_gather_u32 — the chain of
//                                                 seventeen launches — against tlas_build_fused, the one workgroup; byte for byte; prints "OK nodes4=<n>"
#include "lbvh.hip"

#include <cstdio>
#include <cstdlib>
#include <random>

namespace rfwhip {
const EnvSwitches& env_switches()
{
    static EnvSwitches e;
    return e;
}
} // namespace rfwhip
using namespace rfwhip;

int main(int argc, char** argv)
{
    if (argc < 4) return 2;
    const std::string mode = argv[1];
    const uint32_t n = (uint32_t)std::atoi(argv[2]);
    if (mode == "stress") {
        const uint32_t iterations = (uint32_t)std::atoi(argv[3]), seed = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 1u;
        std::vector<char> ws(lbvh_workspace_bytes(n));
        std::vector<DevBox> boxes(n);
        std::vector<Node4> nodes(std::max(n, 1u));
        std::vector<uint32_t> order(n), seen(n);
        uint32_t node_count = 0;
        unsigned long long result[2] = {0, 0};
        const hipError_t e = lbvh_stress(nullptr, n, iterations, seed, ws.data(), ws.size(), boxes.data(), nodes.data(), order.data(), &node_count, seen.data(), result);
        if (e != hipSuccess || result[0] != 0 || result[1] == 0) { std::printf("DIFF error %d, mismatches %llu, checked %llu\n", (int)e, result[0], result[1]); return 1; }
        std::printf("OK checked=%llu\n", result[1]);
        return 0;
    }
    if (mode == "tlas") {
        std::mt19937 rng((uint32_t)std::atoi(argv[3]));
        std::uniform_real_distribution<float> U(0.0f, 1.0f);
        const uint32_t n_mesh = 4, n_slots = n + n / 4 + 1;
        std::vector<DevBox> mesh_local(n_mesh);
        for (auto& b : mesh_local)
            for (int a = 0; a < 3; a++) { const float c = U(rng) - 0.5f, e = 0.1f + U(rng); b.lo[a] = c - e; b.hi[a] = c + e; b.lo[3] = b.hi[3] = 0.0f; }
        std::vector<rfw_mat4> matrices(n_slots);
        std::vector<uint32_t> mesh_of(n_slots), gids(n_slots);
        for (uint32_t i = 0; i < n_slots; i++) {
            float* m = matrices[i].m;
            for (int k = 0; k < 16; k++) m[k] = 0.0f;
            const float s = 0.3f + U(rng), ang = 6.28318f * U(rng);
            m[0] = s * std::cos(ang); m[2] = -s * std::sin(ang); m[5] = s; m[8] = s * std::sin(ang); m[10] = s * std::cos(ang); m[15] = 1.0f;
            m[12] = std::floor(40.0f * U(rng)) * 2.5f; m[13] = 3.0f * U(rng); m[14] = std::floor(40.0f * U(rng)) * 2.5f; // (a lattice: equal keys occur)
            mesh_of[i] = (uint32_t)(rng() % n_mesh);
            gids[i] = i;
        }
        std::shuffle(gids.begin(), gids.end(), rng);
        std::vector<uint32_t> valid(gids.begin(), gids.begin() + n);
        std::vector<char> ws(lbvh_workspace_bytes(n));
        // the chain
        std::vector<DevBox> boxes_a(n), boxes_b(n);
        std::vector<Node4> nodes_a(n), nodes_b(n);
        std::memset(nodes_a.data(), 0, n * sizeof(Node4)); std::memset(nodes_b.data(), 0, n * sizeof(Node4));
        std::vector<uint32_t> order(n), prims_a(n), prims_b(n);
        uint32_t count_a = 0, count_b = 0;
        launch_instance_boxes(nullptr, matrices.data(), mesh_of.data(), mesh_local.data(), valid.data(), n, boxes_a.data());
        hipError_t e = lbvh_build(nullptr, boxes_a.data(), n, ws.data(), ws.size(), nodes_a.data(), order.data(), &count_a);
        launch_gather_u32(nullptr, valid.data(), order.data(), n, prims_a.data());
        if (e != hipSuccess) { std::printf("DIFF lbvh_build returned %d\n", (int)e); return 1; }
        // the one workgroup
        e = tlas_build_fused(nullptr, matrices.data(), mesh_of.data(), mesh_local.data(), valid.data(), n, ws.data(), ws.size(), boxes_b.data(), nodes_b.data(), prims_b.data(), &count_b);
        if (e != hipSuccess) { std::printf("DIFF tlas_build_fused returned %d\n", (int)e); return 1; }
        if (count_a != count_b) { std::printf("DIFF node counts %u %u\n", count_a, count_b); return 1; }
        if (std::memcmp(boxes_a.data(), boxes_b.data(), n * sizeof(DevBox)) != 0) { std::printf("DIFF instance boxes\n"); return 1; }
        if (prims_a != prims_b) { std::printf("DIFF leaf order\n"); return 1; }
        if (std::memcmp(nodes_a.data(), nodes_b.data(), count_a * sizeof(Node4)) != 0) { std::printf("DIFF nodes\n"); return 1; }
        std::printf("OK nodes4=%u\n", count_a);
        return 0;
    }
    return 2;
}
