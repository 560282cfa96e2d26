// Test driver for png_min.hpp: png_tool IN.png OUT.rgba writes width, height (two little-endian uint32) and the RGBA8 pixels that
// Scene::loadEnvMap would receive; prints the decoder's error text and exits 1 on failure.
#include <cstdio>
#include <fstream>

#include "png_min.hpp"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    std::vector<uint8_t> rgba;
    unsigned w = 0, h = 0;
    const std::string err = png_min::decode_rgba(argv[1], rgba, w, h);
    if (!err.empty()) {
        std::fprintf(stderr, "%s\n", err.c_str());
        return 1;
    }
    std::ofstream out(argv[2], std::ios::binary);
    const uint32_t hdr[2] = {w, h};
    out.write((const char *)hdr, sizeof hdr);
    out.write((const char *)rgba.data(), (std::streamsize)rgba.size());
    return out.good() ? 0 : 1;
}
