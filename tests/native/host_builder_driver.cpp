// Sanitizer driver for the host scene builder (csrc/mcpt_scene.cpp): reads a scene description dumped by tests/test_host_builder_asan.py
// and runs build_host_scene for every builder / instancing combination under -fsanitize=address,undefined (CPU only).
#include <cstdint>
#include <cstdio>
#include <fstream>
#include <vector>

#include "../../final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd/csrc/mcpt_internal.h"

int main(int argc, char **argv) {
    if (argc != 2) return 2;
    std::ifstream in(argv[1], std::ios::binary);
    int32_t hdr[3];
    in.read((char *)hdr, sizeof hdr);
    std::vector<mcpt_triangle> tris(hdr[0]);
    std::vector<mcpt_material> mats(hdr[1]);
    std::vector<mcpt_object> objs(hdr[2]);
    in.read((char *)tris.data(), (std::streamsize)(tris.size() * sizeof(mcpt_triangle)));
    in.read((char *)mats.data(), (std::streamsize)(mats.size() * sizeof(mcpt_material)));
    in.read((char *)objs.data(), (std::streamsize)(objs.size() * sizeof(mcpt_object)));
    if (!in) return 3;
    mcpt_scene_desc d{};
    d.n_triangles = hdr[0];
    d.n_materials = hdr[1];
    d.n_objects = hdr[2];
    d.triangles = tris.data();
    d.materials = mats.data();
    d.objects = objs.data();
    const int builders[3] = {MCPT_BUILD_SAH, MCPT_BUILD_REFERENCE, MCPT_BUILD_GPU_LBVH};
    for (int b : builders)
        for (int inst = 0; inst < 2; ++inst)
            for (int quant = -1; quant < 2; ++quant) {
                mcpt::HostScene hs;
                mcpt::BuildChoice c;
                c.builder = b;
                c.instancing = inst;
                c.quantise = quant;
                const char *err = "";
                const int rc = mcpt::build_host_scene(d, hs, &err, c);
                std::printf("builder %d instancing %d quantise %2d: rc %d, %zu nodes, %zu qnodes, %zu instances, height %d, %zu light tris %s\n", b, inst, quant, rc,
                            hs.nodes.size(), hs.qnodes.size(), hs.instances.size(), hs.height, hs.light_tris.size(), err);
                if (rc != 0) return 4;
            }
    return 0;
}
