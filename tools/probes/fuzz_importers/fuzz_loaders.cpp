// fuzz_loaders.cpp — byte-level mutation run of the host library's importers (glTF / GLB with PNG and JPEG images, skins, animations; OBJ + MTL + TGA;
// the image decoders alone) under the address and undefined-behaviour sanitizers.  Not a test: a campaign tool (round 6: 1.98 M mutations, one
// finding — offsets that wrapped gltf.cpp's bounds arithmetic — fixed).
//   python3 tools/probes/fuzz_importers/make_seeds.py                  (seed documents from the tests' writers -> /tmp/fuzz/seeds)
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=undefined -pthread -Irfw-rs_amd/host -o /tmp/fuzz/fuzz_loaders \
//       tools/probes/fuzz_importers/fuzz_loaders.cpp rfw-rs_amd/host/{rfw_host,gltf,gltf_export,jpeg,obj}.cpp -lz
//   ASAN_OPTIONS=detect_leaks=0:allocator_may_return_null=1 /tmp/fuzz/fuzz_loaders /tmp/fuzz/seeds <mutations per seed set> <seed>
// (struct_fuzz.py beside this file: the same for random edits of the JSON tree — indices, counts, types, node cycles — through the Python
// bindings, optionally on a sanitized build of the library: see its docstring.)
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <fstream>
#include <random>
#include <string>
#include <vector>
extern "C" {
void* rfwhost_scene_create();
void rfwhost_scene_destroy(void*);
int rfwhost_load_gltf(void*, const char*, int);
int rfwhost_load_obj(void*, const char*);
int rfwhost_decode_image(const uint8_t*, uint64_t, uint32_t*, uint32_t*, uint8_t*, uint64_t, const char**);
int rfwhost_animate(void*, float);
int rfwhost_save_glb(void*, const char*);
}
namespace fs = std::filesystem;
static std::vector<uint8_t> slurp(const fs::path& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), {}); }
static void mutate(std::vector<uint8_t>& raw, std::mt19937& rng, bool text)
{
    const int n = 1 + (int)(rng() % 5);
    for (int k = 0; k < n && raw.size() > 2; k++) {
        size_t pos = rng() % raw.size();
        switch (rng() % (text ? 6 : 4)) {
        case 0: raw[pos] = (uint8_t)rng(); break;
        case 1: raw.erase(raw.begin() + pos, raw.begin() + std::min(raw.size(), pos + 1 + rng() % 64)); break;
        case 2: { std::vector<uint8_t> ins(1 + rng() % 16); for (auto& b : ins) b = (uint8_t)rng(); raw.insert(raw.begin() + pos, ins.begin(), ins.end()); break; }
        case 3: raw[pos] ^= (uint8_t)(1u << (rng() % 8)); break;
        case 4: { // a number in the text becomes another number
            while (pos < raw.size() && !(raw[pos] >= '0' && raw[pos] <= '9')) pos++;
            if (pos < raw.size()) { static const char* v[] = {"0", "-1", "4294967295", "65536", "1e30", "999999999999", "2147483648", "7"}; const char* s = v[rng() % 8];
                size_t e = pos; while (e < raw.size() && ((raw[e] >= '0' && raw[e] <= '9') || raw[e] == '.')) e++;
                raw.erase(raw.begin() + pos, raw.begin() + e); raw.insert(raw.begin() + pos, s, s + strlen(s)); }
            break; }
        case 5: { size_t a = rng() % raw.size(), len = 1 + rng() % 40; if (a + len < raw.size()) { std::vector<uint8_t> c(raw.begin() + a, raw.begin() + a + len); raw.insert(raw.begin() + pos, c.begin(), c.end()); } break; }
        }
    }
}
int main(int argc, char** argv)
{
    const fs::path seeds = argv[1];
    const int iters = std::atoi(argv[2]);
    std::mt19937 rng((uint32_t)std::atoi(argv[3]));
    const fs::path work = fs::path("/tmp/fuzz/work_") += argv[3];
    long loaded = 0, rejected = 0;
    for (const auto& dir : fs::directory_iterator(seeds)) {
        if (!dir.is_directory()) continue;
        std::vector<fs::path> files;
        for (const auto& f : fs::recursive_directory_iterator(dir)) if (f.is_regular_file()) files.push_back(f.path());
        for (int it = 0; it < iters; it++) {
            fs::remove_all(work); fs::copy(dir, work, fs::copy_options::recursive);
            // mutate one file of the set (the document most of the time)
            fs::path target;
            for (int tries = 0; tries < 8; tries++) { target = files[rng() % files.size()]; const auto e = target.extension(); if (e == ".gltf" || e == ".glb" || e == ".obj" || e == ".mtl" || e == ".jpg" || rng() % 4 == 0) break; }
            std::vector<uint8_t> raw = slurp(target);
            const auto ext = target.extension().string();
            mutate(raw, rng, ext == ".gltf" || ext == ".obj" || ext == ".mtl");
            const fs::path dst = work / fs::relative(target, dir);
            { std::ofstream o(dst, std::ios::binary | std::ios::trunc); o.write((const char*)raw.data(), (std::streamsize)raw.size()); }
            if (dir.path().filename() == "img") {
                uint32_t w = 0, h = 0; const char* err = nullptr;
                std::vector<uint8_t> out(1 << 22);
                (rfwhost_decode_image(raw.data(), raw.size(), &w, &h, out.data(), out.size(), &err) == 0 ? loaded : rejected)++;
                continue;
            }
            for (const auto& f : fs::directory_iterator(work)) {
                const auto e = f.path().extension();
                if (e != ".gltf" && e != ".glb" && e != ".obj") continue;
                void* s = rfwhost_scene_create();
                const int rc = e == ".obj" ? rfwhost_load_obj(s, f.path().c_str()) : rfwhost_load_gltf(s, f.path().c_str(), 1);
                if (rc == 0) { loaded++; rfwhost_animate(s, 0.37f); if (it % 8 == 0) rfwhost_save_glb(s, (work / "out.glb").c_str()); } else rejected++;
                rfwhost_scene_destroy(s);
            }
        }
    }
    std::printf("loaded %ld rejected %ld\n", loaded, rejected);
    return 0;
}
