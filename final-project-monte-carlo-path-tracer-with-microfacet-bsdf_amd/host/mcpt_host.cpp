// Host side above the C ABI: MeshTriangle/OBJ loading, Scene and Renderer of the reference restated over
// include/mcpt.h.  Reference behaviour mirrored: src/Triangle.hpp:83-135 (+ src/OBJ_Loader.hpp:363-521,633-729 for
// the vertex stream), src/Scene.hpp:39-57,104-119, src/Scene.cpp:14-21, src/Renderer.cpp:21-110.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_map>

#include "mcpt_host.hpp"
#include "png_min.hpp"

// ------------------------------------------------------------------------------------------------ OBJ
static int obj_index(const std::string &tok, size_t count) {  // OBJ_Loader.hpp:336-345: 1-based, negative = from the end
    const int idx = std::stoi(tok);
    return idx < 0 ? (int)count + idx : idx - 1;
}

bool load_obj_vertex_stream(const std::string &path, std::vector<Vector3f> &positions, std::vector<Vector2f> &texcoords) {
    positions.clear();
    texcoords.clear();
    if (path.size() < 4 || path.substr(path.size() - 4) != ".obj") return false;  // OBJ_Loader.hpp:365-366
    std::ifstream file(path);
    if (!file.is_open()) return false;
    std::vector<Vector3f> P;
    std::vector<Vector2f> T;
    bool listening = false;
    std::string line;
    while (std::getline(file, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        std::istringstream ss(line);
        std::string head;
        if (!(ss >> head)) continue;
        if (head == "o" || head == "g" || line[0] == 'g') {  // a second mesh is never read by the reference (Triangle.hpp:91)
            if (!listening) listening = true;
            else if (!positions.empty()) break;
        }
        if (head == "v") {
            std::string a, b, c;
            ss >> a >> b >> c;
            P.emplace_back(std::stof(a), std::stof(b), std::stof(c));
        } else if (head == "vt") {
            std::string a, b;
            ss >> a >> b;
            T.emplace_back(std::stof(a), std::stof(b));
        } else if (head == "f") {
            std::string vert;
            while (ss >> vert) {  // v, v/vt, v//vn, v/vt/vn
                const size_t s1 = vert.find('/');
                const std::string vi = vert.substr(0, s1);
                std::string ti;
                if (s1 != std::string::npos) {
                    const size_t s2 = vert.find('/', s1 + 1);
                    ti = vert.substr(s1 + 1, s2 == std::string::npos ? std::string::npos : s2 - s1 - 1);
                }
                positions.push_back(P.at(obj_index(vi, P.size())));
                texcoords.push_back(ti.empty() ? Vector2f(0, 0) : T.at(obj_index(ti, T.size())));
            }
        }
    }
    return !positions.empty();
}

MeshTriangle::MeshTriangle(const std::string &filename, Material *mt, const Vector3f &translation, float zoom) : m(mt) {
    std::vector<Vector3f> P;
    std::vector<Vector2f> T;
    loaded = load_obj_vertex_stream(filename, P, T);  // the reference ignores LoadFile's result (Triangle.hpp:87)
    for (size_t i = 0; i + 2 < P.size(); i += 3) {
        Triangle t;
        t.v0 = zoom * P[i] + translation;
        t.v1 = zoom * P[i + 1] + translation;
        t.v2 = zoom * P[i + 2] + translation;
        if (mt->textured) {
            t.t0 = T[i];
            t.t1 = T[i + 1];
            t.t2 = T[i + 2];
        }
        triangles.push_back(t);
        area += t.area();
    }
}

// ------------------------------------------------------------------------------------------------ Scene
Scene::~Scene() {
    mcpt_scene_destroy(gpu);
    mcpt_group_destroy(group);
}

void Scene::loadEnvMap(const std::string &path) {
    std::vector<uint8_t> rgba;
    const std::string err = png_min::decode_rgba(path, rgba, envWidth, envHeight);
    if (!err.empty()) {
        std::cerr << "Error loading env map (" << path << "): " << err << std::endl;
        return;
    }
    useEnvMap = true;
    envPixels.resize((size_t)envWidth * envHeight * 3);
    for (size_t i = 0; i < (size_t)envWidth * envHeight; ++i)
        for (int c = 0; c < 3; ++c) envPixels[3 * i + c] = rgba[4 * i + c] / 255.0f;
}

void Scene::flatten(std::vector<mcpt_triangle> &tris, std::vector<mcpt_material> &mats, std::vector<mcpt_object> &objs) const {
    std::unordered_map<const Material *, int> ids;
    auto material = [&](const Material *m) {
        auto it = ids.find(m);
        if (it != ids.end()) return it->second;
        mcpt_material r{};
        r.type = (int)m->m_type;
        r.textured = m->textured ? 1 : 0;
        r.roughness = m->roughness;
        r.iorA = m->iorA;
        r.iorB = m->iorB;
        for (int k = 0; k < 3; ++k) {
            r.base_reflectance[k] = m->base_reflectance[k];
            r.emission[k] = m->m_emission[k];
        }
        mats.push_back(r);
        return ids[m] = (int)mats.size() - 1;
    };
    for (const Object *o : objects) {
        mcpt_object r{};
        if (const auto *mesh = dynamic_cast<const MeshTriangle *>(o)) {
            r.kind = MCPT_OBJ_MESH;
            r.material = material(mesh->m);
            r.first_tri = (int)tris.size();
            r.n_tri = (int)mesh->triangles.size();
            for (const Triangle &t : mesh->triangles) {
                mcpt_triangle q{};
                for (int k = 0; k < 3; ++k) {
                    q.v0[k] = t.v0[k];
                    q.v1[k] = t.v1[k];
                    q.v2[k] = t.v2[k];
                }
                for (int k = 0; k < 2; ++k) {
                    q.t0[k] = t.t0[k];
                    q.t1[k] = t.t1[k];
                    q.t2[k] = t.t2[k];
                }
                tris.push_back(q);
            }
        } else if (const auto *s = dynamic_cast<const Sphere *>(o)) {
            r.kind = MCPT_OBJ_SPHERE;
            r.material = material(s->m);
            for (int k = 0; k < 3; ++k) r.center[k] = s->center[k];
            r.radius = s->radius;
        }
        objs.push_back(r);
    }
}

void Scene::buildBVH() {
    printf(" - Generating BVH...\n\n");
    std::vector<mcpt_triangle> tris;
    std::vector<mcpt_material> mats;
    std::vector<mcpt_object> objs;
    flatten(tris, mats, objs);
    mcpt_scene_desc d{};
    d.n_objects = (int)objs.size();
    d.objects = objs.data();
    d.n_triangles = (int)tris.size();
    d.triangles = tris.data();
    d.n_materials = (int)mats.size();
    d.materials = mats.data();
    for (int k = 0; k < 3; ++k) d.background[k] = backgroundColor[k];
    if (useEnvMap) {
        d.env_w = (int)envWidth;
        d.env_h = (int)envHeight;
        d.env_pixels = envPixels.data();
    }
    mcpt_scene_destroy(gpu);
    gpu = nullptr;
    mcpt_group_destroy(group);
    group = nullptr;
    if (devices.size() > 1) {  // one replica per listed device; the tree is built once (mcpt_group_create)
        if (mcpt_group_create(&d, (int)devices.size(), devices.data(), &group) != MCPT_OK) {
            std::cerr << "mcpt: " << mcpt_group_last_error() << std::endl;
            group = nullptr;
            return;
        }
        mcpt_group_info gi;
        if (mcpt_group_get_info(group, &gi) == MCPT_OK)
            std::cout << "[mcpt] scene set-up on " << gi.n_devices << " GPU replicas: " << gi.setup_ms << " ms (tree built once: " << gi.build_ms
                      << " ms; slowest upload " << gi.upload_ms_max << " ms; device start-up " << gi.init_ms_max << " ms beside the build)" << std::endl;
        return;
    }
    if (mcpt_scene_create(&d, devices.empty() ? -1 : devices[0], &gpu) != MCPT_OK) std::cerr << "mcpt: " << mcpt_last_error() << std::endl;
    mcpt_scene_info si;
    if (gpu && mcpt_scene_get_info(gpu, &si) == MCPT_OK)
        std::cout << "[mcpt] scene set-up: build " << si.build_ms << " ms, upload " << si.upload_ms << " ms, device start-up " << si.init_ms
                  << " ms beside the build" << std::endl;
}

mcpt_params Scene::params(int spp) const {
    mcpt_params p{};
    p.spp = spp;
    p.rr_rate = rrRate;
    p.n_dir_sample = n_dir_sample;
    p.enable_shadow = enable_shadow ? 1 : 0;
    p.seed = 1;
    p.tile_size = 32;
    p.nranks = 1;
    return p;
}

mcpt_camera Scene::cameraDesc() const {
    mcpt_camera c{};
    c.width = camera.width;
    c.height = camera.height;
    c.fov = camera.fov;
    c.use_dof = camera.useDOF ? 1 : 0;
    c.focal_distance = camera.focal_distance;
    c.aperture_radius = camera.aperture_radius;
    const Matrix3f R = camera.getOrientation();
    for (int i = 0; i < 3; ++i) {
        c.position[i] = camera.position[i];
        for (int j = 0; j < 3; ++j) c.orientation[3 * i + j] = R(i, j);
    }
    return c;
}

Intersection Scene::intersect(const Ray &ray) const {
    Intersection r;
    if (!gpu) return r;
    int32_t prim = -1;
    double t = r.distance;
    if (mcpt_intersect(gpu, 1, ray.origin.data(), ray.direction.data(), &t, &prim) == MCPT_OK) {
        r.happened = prim >= 0;
        r.distance = t;
        r.primitive = prim;
    }
    return r;
}

float Scene::castRay(const Ray &ray, int depth, const WaveLenType &wavelen) const {
    if (!gpu || depth != 0) return 0.f;
    const mcpt_params p = params(1);
    const uint32_t zero = 0;
    const int32_t ch = (int32_t)wavelen;
    float out = 0.f;
    if (mcpt_cast_rays(gpu, &p, 1, ray.origin.data(), ray.direction.data(), &zero, &zero, &ch, &out) != MCPT_OK)
        std::cerr << "mcpt: " << mcpt_last_error() << std::endl;
    return out;
}

// ------------------------------------------------------------------------------------------------ Renderer
namespace {

struct CheckpointHeader {
    char magic[8];
    int32_t width, height, spp_total, spp_done;
    uint32_t seed;
    int32_t n_dir;
    float rr_rate;
    uint32_t scene_hash;
    int32_t enable_shadow, max_depth;  // (MCPTCKP2: a resume after includeShadow, the depth limit or the environment map changed would
    uint32_t env_hash;                 //  sum samples of two different integrands into one frame)
    uint32_t reserved;
};

uint32_t fnv1a(const void *data, size_t n, uint32_t h) {
    const unsigned char *p = (const unsigned char *)data;
    for (size_t i = 0; i < n; ++i) h = (h ^ p[i]) * 16777619u;
    return h;
}

uint32_t scene_hash(const Scene &scene) {
    std::vector<mcpt_triangle> tris;
    std::vector<mcpt_material> mats;
    std::vector<mcpt_object> objs;
    scene.flatten(tris, mats, objs);
    const mcpt_camera c = scene.cameraDesc();
    uint32_t h = 2166136261u;
    h = fnv1a(tris.data(), tris.size() * sizeof(mcpt_triangle), h);
    h = fnv1a(mats.data(), mats.size() * sizeof(mcpt_material), h);
    h = fnv1a(objs.data(), objs.size() * sizeof(mcpt_object), h);
    h = fnv1a(&c, sizeof c, h);
    h = fnv1a(scene.backgroundColor.data(), 3 * sizeof(float), h);
    return h;
}

bool load_checkpoint(const std::string &file, const CheckpointHeader &want, std::vector<float> &fb, int &spp_done) {
    std::ifstream in(file, std::ios::binary);
    if (!in) return false;
    CheckpointHeader h{};
    in.read((char *)&h, sizeof h);
    if (!in || std::memcmp(h.magic, want.magic, 8) != 0 || h.width != want.width || h.height != want.height || h.spp_total != want.spp_total ||
        h.seed != want.seed || h.n_dir != want.n_dir || h.rr_rate != want.rr_rate || h.scene_hash != want.scene_hash || h.enable_shadow != want.enable_shadow ||
        h.max_depth != want.max_depth || h.env_hash != want.env_hash || h.spp_done <= 0 ||
        h.spp_done > h.spp_total)
        return false;
    std::vector<float> tmp(fb.size());
    in.read((char *)tmp.data(), (std::streamsize)(tmp.size() * sizeof(float)));
    if (!in) return false;
    fb.swap(tmp);
    spp_done = h.spp_done;
    return true;
}

bool save_checkpoint(const std::string &file, CheckpointHeader h, const std::vector<float> &fb, int spp_done) {
    h.spp_done = spp_done;
    const std::string tmp = file + ".tmp";
    {
        std::ofstream out(tmp, std::ios::binary | std::ios::trunc);
        out.write((const char *)&h, sizeof h);
        out.write((const char *)fb.data(), (std::streamsize)(fb.size() * sizeof(float)));
        if (!out) return false;
    }
    return std::rename(tmp.c_str(), file.c_str()) == 0;  // atomic replacement: a crash never leaves half a file under the name
}

}  // namespace

void Renderer::Render(const Scene &scene) {
    const Camera &camera = scene.camera;
    std::vector<float> framebuffer((size_t)camera.width * camera.height * 3, 0.f);
    std::cout << "SPP: " << spp << "\n";
    if (!scene.handle()) {
        std::cerr << "mcpt: the scene was not built (call Scene::buildBVH first): " << mcpt_last_error() << std::endl;
        return;
    }
    const mcpt_camera c = scene.cameraDesc();
    mcpt_params p = scene.params(spp);
    mcpt_stats st{}, total{};
    auto render = [&](const mcpt_params &q) {  // Renderer.cpp:36-90
        const int rc = scene.groupHandle() ? mcpt_group_render(scene.groupHandle(), &c, &q, framebuffer.data(), &st)
                                           : mcpt_render(scene.handle(), &c, &q, framebuffer.data(), &st);
        if (rc != MCPT_OK) std::cerr << "mcpt: " << (scene.groupHandle() ? mcpt_group_last_error() : mcpt_last_error()) << std::endl;
        total.samples += st.samples;
        total.iterations += st.iterations;
        total.ms_total += st.ms_total;
        return rc;
    };
    if (checkpoint_path.empty()) {
        const int rc = render(p);
        if (rc != MCPT_OK && rc != MCPT_ERR_OVERFLOW) return;
    } else {
        CheckpointHeader h{};
        std::memcpy(h.magic, "MCPTCKP2", 8);
        h.width = camera.width;
        h.height = camera.height;
        h.spp_total = spp;
        h.seed = p.seed;
        h.n_dir = p.n_dir_sample;
        h.rr_rate = p.rr_rate;
        h.scene_hash = scene_hash(scene);
        h.enable_shadow = p.enable_shadow;
        h.max_depth = p.max_depth;
        h.env_hash = 0u;
        if (scene.useEnvMap) {
            const unsigned dims[2] = {scene.envWidth, scene.envHeight};
            h.env_hash = fnv1a(scene.envPixels.data(), scene.envPixels.size() * sizeof(float), fnv1a(dims, sizeof dims, 2166136261u));
        }
        int done = 0;
        if (load_checkpoint(checkpoint_path, h, framebuffer, done))
            std::cout << "[mcpt] resuming from " << checkpoint_path << " at " << done << " of " << spp << " spp" << std::endl;
        const int every = checkpoint_every > 0 ? checkpoint_every : spp;
        while (done < spp) {
            mcpt_params q = p;
            q.spp = std::min(every, spp - done);
            q.spp_total = spp;
            q.sample_offset = done;
            q.accumulate = done > 0 ? 1 : 0;
            const int rc = render(q);
            if (rc != MCPT_OK && rc != MCPT_ERR_OVERFLOW) return;
            done += q.spp;
            if (!save_checkpoint(checkpoint_path, h, framebuffer, done)) std::cerr << "mcpt: cannot write checkpoint " << checkpoint_path << std::endl;
            if (stop_after > 0 && done >= stop_after && done < spp) {
                std::cout << "[mcpt] stopped after " << done << " spp (checkpoint kept)" << std::endl;
                return;
            }
        }
    }
    st = total;
    std::cout << "[mcpt] " << st.samples / 1e6 << " Msamples in " << st.ms_total << " ms = "
              << (st.ms_total > 0 ? st.samples / st.ms_total / 1e3 : 0.0) << " Msamples/s, " << st.iterations
              << " wavefront iterations" << (scene.groupHandle() ? " on " + std::to_string(mcpt_group_size(scene.groupHandle())) + " GPU replicas" : std::string()) << std::endl;

    std::cout << "Writing image to " << path << std::endl;
    std::vector<unsigned char> raw((size_t)4 * camera.width * camera.height);
    // Renderer.cpp:95-103 on the GPU (mcpt_tonemap); the loop below is the reference's own, kept as the path taken if that call fails
    if (mcpt_tonemap(scene.handle(), framebuffer.data(), (int64_t)camera.width * camera.height, raw.data()) == MCPT_OK) {
        const std::string err = png_min::encode_rgba(path, raw, camera.width, camera.height);
        if (!err.empty()) std::cerr << "Error when writing image : " << err << std::endl;
        return;
    }
    std::cerr << "mcpt: " << mcpt_last_error() << std::endl;
    const float inv_gamma = 0.45f;
    auto clamp255 = [](float v) {  // clamp(0, 255, v) with std::min/std::max semantics: NaN -> 255 (global.hpp:16-18)
        const float a = (v < 255.f) ? v : 255.f;
        return (0.f < a) ? a : 0.f;
    };
    for (size_t i = 0; i < (size_t)camera.width * camera.height; i++) {  // Renderer.cpp:95-103
        for (int k = 0; k < 3; ++k) raw[4 * i + k] = (unsigned char)clamp255(255 * std::pow(framebuffer[3 * i + k], inv_gamma));
        raw[4 * i + 3] = 255;
    }
    const std::string err = png_min::encode_rgba(path, raw, camera.width, camera.height);
    if (!err.empty()) std::cerr << "Error when writing image : " << err << std::endl;
}
