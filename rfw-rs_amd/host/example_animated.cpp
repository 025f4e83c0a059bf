// example_animated.cpp — the reference's `examples/animated` (examples/animated/src/main.rs) as a headless C++ program on the MI355X
// backend: a scene (a glTF file, or the synthetic atrium), a grid of icosphere instances that bounce every frame (bounce_spheres,
// main.rs:197-219), animated glTF models placed the way the reference places its two CesiumMan graphs (main.rs:79-103; `--actor`, up to
// twice) and the animation timer system (set_animation_timers, main.rs:221-223: Scene::set_animations_time every frame -> joint matrices
// -> skinning and BLAS refit on the device), driven exactly the way rfw drives a backend — synchronize_system, then render_system, every
// frame (rfw/src/system/mod.rs:19-206, rfw/src/lib.rs:411-430) — through the rfw::Backend interface of rfw_host.hpp.
//
// Where the reference presents to a window, this program downloads the presented (Bgra8UnormSrgb) frame of every frame into a ring of
// pinned host buffers without stalling the frames in flight, and writes the last one as a PPM image.
//
//   example_animated [--gltf scene.glb | scene.obj] [--actor animated.gltf]... [--frames N] [--size WxH] [--spheres NXxNZ] [--path-length L] [--out last.ppm]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "rfw_host.hpp"

int main(int argc, char** argv)
{
    std::string gltf, out = "example_animated.ppm";
    std::vector<std::string> actors;
    uint32_t frames = 240, width = 1280, height = 720, nx = 100, nz = 100, path_length = 2;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--gltf") gltf = next();
        else if (a == "--actor") actors.push_back(next());
        else if (a == "--frames") frames = (uint32_t)std::atoi(next());
        else if (a == "--size") std::sscanf(next(), "%ux%u", &width, &height);
        else if (a == "--spheres") std::sscanf(next(), "%ux%u", &nx, &nz);
        else if (a == "--path-length") path_length = (uint32_t)std::atoi(next());
        else if (a == "--out") out = next();
        else { std::fprintf(stderr, "unknown argument %s\n", a.c_str()); return 2; }
    }
    if (!width || !height || !frames) { std::fprintf(stderr, "bad --size / --frames\n"); return 2; }
    try {
        rfw::Scene scene;
        rfw::Camera3D camera;
        if (!gltf.empty()) {
            std::string err;
            const bool obj = gltf.size() > 4 && gltf.compare(gltf.size() - 4, 4, ".obj") == 0; // Scene::load picks the loader by extension
            if (!(obj ? rfw::load_obj(gltf, scene, err) : rfw::load_gltf(gltf, scene, &camera, err))) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
        } else {
            rfw::build_atrium(scene, camera, 262267, 0xC0FFEE);
        }
        for (size_t k = 0; k < actors.size(); k++) { // main.rs:84-103: the first graph scaled by 3, the second moved to x = -3, both turned by 180 degrees about y
            std::string err;
            if (k > 0 && actors[k] == actors[0]) scene.instantiate_graph(scene.graphs.size() - k); // add_3d(&descriptor) again: same meshes, new instances and skins
            else if (!rfw::load_gltf(actors[k], scene, nullptr, err)) { std::fprintf(stderr, "%s\n", err.c_str()); return 1; }
            const double turn[4] = {0.0, 1.0, 0.0, 6.123233995736766e-17}; // quaternion of a half turn about y
            const double t0[3] = {0, 0, 0}, t1[3] = {-3.0 * (double)k, 0, 0}, s3[3] = {3, 3, 3}, s1[3] = {1, 1, 1};
            scene.graphs.back().set_root_transform(scene, k == 0 ? t0 : t1, turn, k == 0 ? s3 : s1);
        }
        scene.set_animations_time(0.0); // main.rs:112
        camera.aspect_ratio = (float)width / (float)height;
        const float spacing = 0.28f;
        rfw::add_sphere_grid(scene, nx, nz, spacing);
        const uint32_t grid_mesh = scene.meshes_3d.rbegin()->first;

        rfw_hip_options opt;
        std::memset(&opt, 0, sizeof(opt));
        opt.struct_size = sizeof(opt);
        opt.device = -1;
        opt.max_path_length = path_length;
        opt.world = 1;
        opt.frames_in_flight = 4; // GPU_MAX_HW_QUEUES >= 4 is the runtime's default
        std::unique_ptr<rfw::HipBackend> renderer(rfw::HipBackend::init(width, height, 1.0, &opt));

        const uint64_t px = (uint64_t)width * height;
        std::vector<uint32_t*> ring(8, nullptr); // presented frames land here, 8 frames of slack
        for (uint32_t*& p : ring)
            if (!(p = (uint32_t*)rfw_hip_host_alloc(px * 4))) throw std::runtime_error("rfw_hip_host_alloc failed");

        rfw::synchronize_system(scene, *renderer); // first synchronize: uploads and builds everything
        const auto t0 = std::chrono::steady_clock::now();
        uint64_t rays = 0;
        for (uint32_t f = 0; f < frames; f++) {
            rfw::animate_sphere_grid(scene, grid_mesh, nx, nz, spacing, (float)f / 60.0f); // bounce_spheres at 60 Hz
            scene.set_animations_time((double)f / 60.0);                                   // set_animation_timers
            rfw::synchronize_system(scene, *renderer);                                    // changed instance lists -> TLAS of this frame
            rfw::render_system(camera, width, height, *renderer);
            uint32_t* dst = ring[f % ring.size()];
            if (f >= ring.size() && rfw_hip_wait_download(renderer->raw(), dst) != RFW_HIP_OK) throw std::runtime_error(rfw_hip_last_error(renderer->raw()));
            if (rfw_hip_download_frame(renderer->raw(), 2, 0, (float*)dst, px) != RFW_HIP_OK) throw std::runtime_error(rfw_hip_last_error(renderer->raw()));
        }
        if (rfw_hip_wait_downloads(renderer->raw()) != RFW_HIP_OK) throw std::runtime_error(rfw_hip_last_error(renderer->raw()));
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        rfw_hip_frame_stats st;
        std::memset(&st, 0, sizeof(st));
        if (rfw_hip_get_frame_stats(renderer->raw(), &st) == RFW_HIP_OK) rays = st.primary_rays + st.extension_rays + st.shadow_rays;
        std::printf("%u frames of %ux%u, %llu triangles, %u animated instances: %.2f ms per frame (%.0f frames/s), ~%.0f Mrays/s, every frame presented to host memory\n",
                    frames, width, height, (unsigned long long)scene.triangle_count(), nx * nz, secs / frames * 1e3, frames / secs,
                    (double)rays * frames / secs / 1e6);

        const uint32_t* last = ring[(frames - 1) % ring.size()];
        if (FILE* fp = std::fopen(out.c_str(), "wb")) {
            std::fprintf(fp, "P6\n%u %u\n255\n", width, height);
            std::vector<uint8_t> row(3 * (size_t)width);
            for (uint32_t y = 0; y < height; y++) {
                for (uint32_t x = 0; x < width; x++) {
                    const uint32_t c = last[(size_t)y * width + x]; // B, G, R, A in memory order
                    row[3 * x] = (uint8_t)(c >> 16); row[3 * x + 1] = (uint8_t)(c >> 8); row[3 * x + 2] = (uint8_t)c;
                }
                std::fwrite(row.data(), 1, row.size(), fp);
            }
            std::fclose(fp);
            std::printf("last frame -> %s\n", out.c_str());
        }
        for (uint32_t* p : ring) rfw_hip_host_free(p);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "example_animated: %s\n", e.what());
        return 1;
    }
    return 0;
}
