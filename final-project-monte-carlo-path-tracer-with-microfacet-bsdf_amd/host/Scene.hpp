// Host-side Scene: the public surface of the reference's Scene (src/Scene.hpp:24-152) over the C ABI.
//   Add / setRrRate / setDirectLightSample / enableShadow / loadEnvMap / backgroundColor / camera / buildBVH
// keep their names and meaning; buildBVH() flattens the objects and uploads them to the GPU (mcpt_scene_create);
// intersect() and castRay() forward to mcpt_intersect / mcpt_cast_rays.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/mcpt.h"
#include "Camera.hpp"
#include "Object.hpp"

enum WaveLenType { RED, GREEN, BLUE };  // src/WaveLen.hpp:5

struct Ray {  // src/Ray.hpp:6-19 (the inverse direction is computed on the device)
    Vector3f origin, direction;
    Ray(const Vector3f &o, const Vector3f &d) : origin(o), direction(d) {}
};

struct Intersection {  // the parts of src/Intersection.hpp the C ABI reports
    bool happened = false;
    double distance = 1.7976931348623157e308;
    int primitive = -1;
};

class Scene {
    float rrRate = 0.7f;  // Scene.hpp:25
    bool enable_shadow = true;
    int n_dir_sample = 4;  // Scene.hpp:28 (conf.json's directLightSample is never applied by the reference's main)

  public:
    Camera camera;
    Vector3f backgroundColor = Vector3f(0, 0, 0);
    bool useEnvMap = false;
    unsigned envWidth = 0, envHeight = 0;
    std::vector<float> envPixels;  // envWidth*envHeight*3 floats in [0,1]

    explicit Scene(Camera cam) : camera(cam) {}
    ~Scene();
    Scene(const Scene &) = delete;
    Scene &operator=(const Scene &) = delete;

    void loadEnvMap(const std::string &path);  // Scene.hpp:39-57: on failure prints the error and keeps the colour
    void Add(Object *object) {                 // Scene.hpp:104-109
        objects.push_back(object);
        if (object->hasEmit()) lightsObjects.push_back(object);
    }
    void setRrRate(float rr) { rrRate = rr < 0.99f ? rr : 0.99f; }  // Scene.hpp:110-113
    void setDirectLightSample(int x) { n_dir_sample = x; }
    void enableShadow(bool shadow) { enable_shadow = shadow; }
    const std::vector<Object *> &get_objects() const { return objects; }

    void buildBVH();  // Scene.cpp:14-17
    Intersection intersect(const Ray &ray) const;                                // Scene.cpp:19-21
    float castRay(const Ray &ray, int depth, const WaveLenType &wavelen) const;  // Scene.cpp:85-184 (depth must be 0)

    // used by Renderer
    mcpt_scene *handle() const { return gpu; }
    mcpt_params params(int spp) const;
    mcpt_camera cameraDesc() const;
    // flat description (also used by the tests to compare with the Python assembly)
    void flatten(std::vector<mcpt_triangle> &tris, std::vector<mcpt_material> &mats, std::vector<mcpt_object> &objs) const;

    std::vector<Object *> objects;
    std::vector<Object *> lightsObjects;

  private:
    mcpt_scene *gpu = nullptr;
};
