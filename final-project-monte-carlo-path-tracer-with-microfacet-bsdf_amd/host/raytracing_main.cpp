// RayTracing: the reference's executable (src/main.cpp) over the GPU library.  Same behaviour: material presets,
// the DEMO Cornell box when built with -DDEMO, otherwise the chess scene driven by ./conf.json, output PNG,
// wall-clock print.  Quirks kept on purpose: scene.directLightSample, scene.model_quality,
// renderer.path and renderer.parrallelism are ignored; addDiamond only has to be present; lightBrightness must be a
// JSON float.  Additions: optional command-line overrides, because DEMO hard-codes 384x384 / spp 2048:
//   --width N --height N --spp N --output FILE --conf FILE --models DIR
//   --fixed       honour the keys the shipped main ignores: scene.directLightSample (Scene::setDirectLightSample, Scene.hpp:114),
//                 scene.model_quality (low_* / high_* models; main.cpp:24-26 composes the paths before reading the key) and
//                 scene.addDiamond:false (main.cpp:197-199 tests presence only)
//   --gpus N      render on GPUs 0..N-1 (tile partition + RCCL merge inside the library); --devices 0,0 lists them explicitly
//   --checkpoint FILE [--checkpoint-every N]   render in chunks of N spp (default 64), keep the running frame in FILE and resume from it
//   --dump FILE   write the flattened scene (what mcpt_scene_create receives) and exit without touching the GPU
#include <chrono>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <unordered_map>

#include "mcpt_host.hpp"
#include "json_min.hpp"

[[maybe_unused]] static bool is_v3(const Json &d) {
    if (!d.is_array() || d.size() != 3) return false;
    for (size_t i = 0; i < 3; ++i)
        if (!d[i].is_number()) return false;
    return true;
}
[[maybe_unused]] static Vector3f v3(const Json &d) { return Vector3f(d[0].as_float(), d[1].as_float(), d[2].as_float()); }

static Vector3f light_emission(float scale) {  // main.cpp:100-104,303-308
    return scale * (8.0f * Vector3f(0.747f + 0.058f, 0.747f + 0.258f, 0.747f) + 15.6f * Vector3f(0.740f + 0.287f, 0.740f + 0.160f, 0.740f) +
                    18.4f * Vector3f(0.737f + 0.642f, 0.737f + 0.159f, 0.737f));
}

int main(int argc, char **argv) {
    Camera camera;
    Scene scene(camera);
    Renderer r;

    std::string models = "../models", conf_path = "conf.json", out_override, dump_path;
    int w_override = 0, h_override = 0, spp_override = 0;
    bool fixed = false;
    for (int i = 1; i < argc; ++i)
        if (std::string(argv[i]) == "--fixed") {  // the only flag without a value: take it out of the (flag, value) list
            fixed = true;
            for (int k = i; k + 1 < argc; ++k) argv[k] = argv[k + 1];
            --argc;
            --i;
        }
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--width") w_override = std::atoi(argv[i + 1]);
        else if (a == "--height") h_override = std::atoi(argv[i + 1]);
        else if (a == "--spp") spp_override = std::atoi(argv[i + 1]);
        else if (a == "--output") out_override = argv[i + 1];
        else if (a == "--conf") conf_path = argv[i + 1];
        else if (a == "--models") models = argv[i + 1];
        else if (a == "--dump") dump_path = argv[i + 1];
        else if (a == "--checkpoint") r.checkpoint_path = argv[i + 1];
        else if (a == "--checkpoint-every") r.checkpoint_every = std::atoi(argv[i + 1]);
        else if (a == "--stop-after") r.stop_after = std::atoi(argv[i + 1]);
        else if (a == "--gpus") {
            std::vector<int> dev;
            for (int k = 0; k < std::atoi(argv[i + 1]); ++k) dev.push_back(k);
            scene.setDevices(dev);
        } else if (a == "--devices") {
            std::vector<int> dev;
            std::stringstream ss(argv[i + 1]);
            std::string tok;
            while (std::getline(ss, tok, ',')) dev.push_back(std::atoi(tok.c_str()));
            scene.setDevices(dev);
        }
    }
    bool use_diamond = false;
    std::string model_quality = "low";  // the conf value is read after the paths are composed (main.cpp:24-26,200-202)
#ifndef DEMO
    if (fixed) {  // --fixed: read the key first, as its note in conf.json promises
        try {
            std::ifstream in(conf_path);
            std::stringstream buf;
            buf << in.rdbuf();
            const Json pre = Json::parse(buf.str());
            const Json &q = pre["scene"]["model_quality"];
            if (q.is_string() && (q.as_string() == "low" || q.as_string() == "high")) model_quality = q.as_string();
        } catch (const std::exception &) {
        }
    }
#endif
    const std::string king_model = models + "/" + model_quality + "_king.obj";
    const std::string soldier_model = models + "/" + model_quality + "_soldier.obj";

    int w = 384, h = 384;
    Vector3f camPos(278, 273, -800), camTarget(278, 273, 0), camUp(0, 1, 0);

    // material presets, main.cpp:34-97
    std::unordered_map<std::string, Material *> materials;
    auto preset = [&](const char *name, MaterialType t, float rough, Vector3f refl, float iorA = -1, float iorB = -1) {
        Material *m = new Material(t, Vector3f(0, 0, 0));
        m->roughness = rough;
        m->base_reflectance = refl;
        if (iorA >= 0) m->iorA = iorA;
        if (iorB >= 0) m->iorB = iorB;
        return materials[name] = m;
    };
    // A material name that is not a preset: the reference indexes its map with it (main.cpp:232,259-262,283) and dereferences the null
    // pointer the map hands back.  Here the name is reported and the default of an unset key, rough_plastic, is used.
    [[maybe_unused]] auto named = [&](const std::string &name) -> Material * {
        auto it = materials.find(name);
        if (it != materials.end() && it->second) return it->second;
        std::cerr << "Unknown material \"" << name << "\": using rough_plastic" << std::endl;
        return materials["rough_plastic"];
    };
    preset("rough_red_conductor", ROUGH_CONDUCTOR, 0.1f, {1.0f, 0.0f, 0.0f});
    preset("rough_white_conductor", ROUGH_CONDUCTOR, 0.4f, {0.725f, 0.71f, 0.68f});
    preset("green_mirror", ROUGH_CONDUCTOR, 0.01f, {0.14f, 1.0f, 0.14f});
    preset("gold_conductor", SMOOTH_CONDUCTOR, 0.0001f, {1.0f, 0.85f, 0.57f});
    preset("silver_mirror", SMOOTH_CONDUCTOR, 0.001f, {0.972f, 0.960f, 0.915f});
    preset("smooth_glass", SMOOTH_DIELECTRIC, 0.01f, {0, 0, 0}, 1.7f, 0.04f);
    preset("smooth_glass_gem", SMOOTH_DIELECTRIC, 0.001f, {0, 0, 0}, 1.3f, 0.2f);
    preset("clear_rough_plastic", ROUGH_DIELECTRIC, 0.02f, {0, 0, 0}, 1.5f, 0.01f);
    preset("rough_plastic", ROUGH_DIELECTRIC, 0.4f, {0, 0, 0}, 1.5f, 0.01f);

#ifdef DEMO
    // main.cpp:100-129
    Material *light = new Material(ROUGH_CONDUCTOR, light_emission(3.9f));
    const std::string cb = models + "/cornellbox/";
    scene.Add(new MeshTriangle(cb + "floor.obj", materials["rough_white_conductor"]));
    scene.Add(new MeshTriangle(cb + "shortbox.obj", materials["green_mirror"]));
    scene.Add(new MeshTriangle(cb + "tallbox.obj", materials["rough_plastic"]));
    scene.Add(new MeshTriangle(cb + "left.obj", materials["rough_red_conductor"]));
    scene.Add(new MeshTriangle(cb + "right.obj", materials["gold_conductor"]));
    scene.Add(new MeshTriangle(cb + "light.obj", light));
    scene.Add(new Sphere({400, 90, 3}, 80, materials["smooth_glass"]));
    scene.Add(new Sphere({250, 260, 230}, 60, materials["clear_rough_plastic"]));
    scene.Add(new Sphere({120, 390, 400}, 50, materials["silver_mirror"]));
    camera.useDOF = false;
    camera.focal_distance = 900;
    camera.aperture_radius = 40;
    (void)conf_path;
    (void)fixed;
    (void)use_diamond;
    (void)king_model;
    (void)soldier_model;
#else
    // main.cpp:137-294
    Vector3f kingPosition(0.0f, 0.0f, 0.0f), lightPosition(0, 200, 0);
    Material *kingMaterial = materials["rough_plastic"], *floorMaterial = materials["rough_plastic"];
    float brightness_scale = 1.0f;
    try {
        std::ifstream in(conf_path);
        std::stringstream buf;
        buf << in.rdbuf();
        const Json data = Json::parse(buf.str());
        const Json &cc = data["camera"];
        if (!cc.is_null()) {
            if (cc["width"].is_number()) w = cc["width"].as_int();
            if (cc["height"].is_number()) h = cc["height"].as_int();
            if (cc["fov"].is_number()) camera.fov = cc["fov"].as_float();
            if (is_v3(cc["position"])) camPos = v3(cc["position"]);
            if (is_v3(cc["target"])) camTarget = v3(cc["target"]);
            if (is_v3(cc["up"])) camUp = v3(cc["up"]);
            if (cc["useDOF"].is_boolean()) camera.useDOF = cc["useDOF"].as_bool();
            if (camera.useDOF && cc["focusDistance"].is_number()) camera.focal_distance = cc["focusDistance"].as_float();
            if (camera.useDOF && cc["apertureRadius"].is_number()) camera.aperture_radius = cc["apertureRadius"].as_float();
        }
        const Json &cr = data["renderer"];
        if (!cr.is_null()) {
            if (cr["spp"].is_number()) r.setSpp(cr["spp"].as_int());
            if (cr["output"].is_string()) r.path = cr["output"].as_string();  // "path" is not read (main.cpp:191)
        }
        const Json &cs = data["scene"];
        if (!cs.is_null()) {
            if (cs["addDiamond"].is_boolean()) use_diamond = fixed ? cs["addDiamond"].as_bool() : true;  // main.cpp:197-199
            if (fixed && cs["directLightSample"].is_number() && cs["directLightSample"].as_int() > 0)
                scene.setDirectLightSample(cs["directLightSample"].as_int());  // Scene.hpp:114 (no caller in the shipped main)
            if (cs["includeShadow"].is_boolean()) scene.enableShadow(cs["includeShadow"].as_bool());
            if (cs["RussianRouletteRate"].is_number()) scene.setRrRate(cs["RussianRouletteRate"].as_float());
            if (!cs["envMap"].is_null()) {
                if (cs["envMap"].is_string()) scene.loadEnvMap(cs["envMap"].as_string());
                else if (is_v3(cs["envMap"])) scene.backgroundColor = v3(cs["envMap"]);
            }
            if (is_v3(cs["kingPosition"])) kingPosition = v3(cs["kingPosition"]);
            if (cs["kingMaterial"].is_string()) kingMaterial = named(cs["kingMaterial"].as_string());
            if (cs.contains("soldierLeftRowPosition") && cs.contains("soldierRightRowPosition") && cs.contains("soldierMaterials")) {
                const Json &lrow = cs["soldierLeftRowPosition"], &rrow = cs["soldierRightRowPosition"];
                const float xs = cs["soldierXSpacing"].as_float(), ys = cs["soldierYSpacing"].as_float(), zs = cs["soldierZSpacing"].as_float();
                const int count = cs["soldierCountPerRow"].as_int();
                const Json &names = cs["soldierMaterials"];
                for (int i = 0; i < count; i++) {  // main.cpp:248-271
                    const float xo = i * xs, yo = i * ys, zo = i * zs;
                    const Vector3f lp(lrow[0].as_float() + xo, lrow[1].as_float() + yo, lrow[2].as_float() + zo);
                    const Vector3f rp(rrow[0].as_float() + xo, rrow[1].as_float() + yo, rrow[2].as_float() + zo);
                    Material *lm = ((size_t)i < names.size()) ? named(names[i].as_string()) : materials["rough_plastic"];
                    Material *rm = ((size_t)(i + count) < names.size()) ? named(names[i + count].as_string()) : materials["rough_plastic"];
                    scene.Add(new MeshTriangle(soldier_model, lm, lp));
                    scene.Add(new MeshTriangle(soldier_model, rm, rp));
                }
            }
            if (is_v3(cs["lightPosition"])) lightPosition = v3(cs["lightPosition"]);
            if (cs["lightBrightness"].is_number_float()) brightness_scale = cs["lightBrightness"].as_float();
            if (cs["floorMaterial"].is_string()) {
                floorMaterial = named(cs["floorMaterial"].as_string());
                floorMaterial->textured = cs["floor_isTextured"].as_bool();
            }
        }
    } catch (const std::exception &e) {
        std::cerr << "Error when reading json config: " << e.what() << std::endl;
    }
    Material *light = new Material(ROUGH_CONDUCTOR, light_emission(brightness_scale));
    // main.cpp:296-316: the wall is constructed by the reference but never added
    scene.Add(new MeshTriangle(models + "/light.obj", light, lightPosition));
    scene.Add(new MeshTriangle(models + "/bottom.obj", floorMaterial, Vector3f::Zero()));
    scene.Add(new MeshTriangle(king_model, kingMaterial, kingPosition));
    if (use_diamond) scene.Add(new MeshTriangle(models + "/diamond.obj", materials["smooth_glass_gem"]));
#endif

    if (w_override > 0) w = w_override;
    if (h_override > 0) h = h_override;
    if (spp_override > 0) r.setSpp(spp_override);
    if (!out_override.empty()) r.path = out_override;
    camera.width = w;
    camera.height = h;
    camera.position = camPos;
    camera.lookAt(camTarget, camUp);
    scene.camera = camera;

    if (!dump_path.empty()) {  // test hook: the flat description, byte for byte
        std::vector<mcpt_triangle> tris;
        std::vector<mcpt_material> mats;
        std::vector<mcpt_object> objs;
        scene.flatten(tris, mats, objs);
        const mcpt_camera c = scene.cameraDesc();
        const mcpt_params p = scene.params(0);
        std::ofstream out(dump_path, std::ios::binary);
        const int32_t hdr[4] = {(int32_t)tris.size(), (int32_t)mats.size(), (int32_t)objs.size(), scene.useEnvMap ? 1 : 0};
        out.write((const char *)hdr, sizeof hdr);
        out.write((const char *)tris.data(), tris.size() * sizeof(mcpt_triangle));
        out.write((const char *)mats.data(), mats.size() * sizeof(mcpt_material));
        out.write((const char *)objs.data(), objs.size() * sizeof(mcpt_object));
        out.write((const char *)&c, sizeof c);
        out.write((const char *)&p.rr_rate, sizeof(float));
        out.write((const char *)scene.backgroundColor.data(), 3 * sizeof(float));
        out.write((const char *)&p.n_dir_sample, sizeof(int32_t));
        return out.good() ? 0 : 1;
    }

    scene.buildBVH();

    const auto start = std::chrono::system_clock::now();
    r.Render(scene);
    const auto ms = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::system_clock::now() - start).count();
    std::cout << "Rendering finished in " << ms / 3600000 << ":" << (ms / 60000) % 60 << ":" << (ms / 1000) % 60 << "." << ms % 1000 << std::endl;
    return 0;
}
