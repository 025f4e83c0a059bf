// fuzz_importers.cpp — mutation run over the host importers' decoders (JPEG, PNG, TGA, glTF documents with animations) under ASan + UBSan:
// not part of the product, not a test the suites run.  Seeds: /tmp/fuzz/s0..s7.jpg, s8..s13.png, a.tga, b.tga, animated.gltf (written
// with tests/gltf_util.py and Pillow: baseline / progressive JPEGs with and without restart markers, PNGs of several depths with Adam7).
//   cd rfw-rs_amd/host && g++ -O1 -g -std=c++17 -fsanitize=address,undefined -I. -o /tmp/fuzz/fuzz ../../tools/probes/fuzz_importers.cpp \
//       rfw_host.cpp gltf.cpp gltf_export.cpp jpeg.cpp obj.cpp -lz -pthread && /tmp/fuzz/fuzz 40000
// Round 2: 40 000 iterations = 160 000 decodes, 32 k accepted and 128 k refused, no sanitizer report.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>
#include <vector>
#include "rfw_host.hpp"
static std::vector<uint8_t> slurp(const std::string& p) { std::ifstream f(p, std::ios::binary); return std::vector<uint8_t>(std::istreambuf_iterator<char>(f), {}); }
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 1000;
    std::mt19937 rng(12345);
    std::vector<std::string> imgs;
    for (int i = 0; i < 8; i++) imgs.push_back("/tmp/fuzz/s" + std::to_string(i) + ".jpg");
    for (int i = 8; i < 14; i++) imgs.push_back("/tmp/fuzz/s" + std::to_string(i) + ".png");
    size_t ok = 0, bad = 0;
    for (int it = 0; it < iters; it++) {
        std::vector<uint8_t> raw = slurp(imgs[it % imgs.size()]);
        const int edits = 1 + rng() % 5;
        for (int e = 0; e < edits && raw.size() > 4; e++) {
            const size_t pos = 2 + rng() % (raw.size() - 2);
            switch (rng() % 3) {
            case 0: raw[pos] = (uint8_t)rng(); break;
            case 1: raw.erase(raw.begin() + pos, raw.begin() + std::min(raw.size(), pos + 1 + rng() % 40)); break;
            default: raw.insert(raw.begin() + pos, (size_t)(1 + rng() % 8), (uint8_t)rng()); break;
            }
        }
        uint32_t w, h; std::vector<uint8_t> rgba; std::string err;
        try { (rfw::decode_image(raw.data(), raw.size(), w, h, rgba, err) ? ok : bad)++; } catch (const std::exception& e) { printf("decode_image threw %s\n", e.what()); bad++; }
        for (const char* t : {"/tmp/fuzz/a.tga", "/tmp/fuzz/b.tga"}) {
            std::vector<uint8_t> r2 = slurp(t);
            r2[rng() % 18] = (uint8_t)rng();
            if (rng() & 1) r2.resize(18 + rng() % (r2.size() - 18));
            (rfw::decode_tga(r2.data(), r2.size(), w, h, rgba, err) ? ok : bad)++;
        }
        // the animated document: mutate the JSON text
        std::vector<uint8_t> g = slurp("/tmp/fuzz/animated.gltf");
        for (int e = 0; e < 3; e++) g[rng() % g.size()] = (uint8_t)(32 + rng() % 95);
        { std::ofstream o("/tmp/fuzz/m.gltf", std::ios::binary); o.write((const char*)g.data(), (std::streamsize)g.size()); }
        rfw::Scene sc; rfw::Camera3D cam;
        try {
            if (rfw::load_gltf("/tmp/fuzz/m.gltf", sc, &cam, err)) { sc.set_animations_time(0.37 * it); if (!sc.graphs.empty()) sc.instantiate_graph(0); ok++; } else bad++;
        } catch (const std::exception& e) { bad++; } // the C API (rfwhost_load_gltf) catches the same way
    }
    printf("decoded %zu, refused %zu\n", ok, bad);
    return 0;
}
