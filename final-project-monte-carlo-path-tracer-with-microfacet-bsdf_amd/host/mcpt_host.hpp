// mcpt_host.hpp -- C++ host side above the C ABI (include/mcpt.h).
//
// One header for the whole host mirror: small float vectors, Material, the scene objects (MeshTriangle, Sphere), Camera,
// Scene and Renderer.  Class and member names follow the reference's public interface (Renderer::Render / setSpp / path,
// Scene::Add / setRrRate / setDirectLightSample / enableShadow / loadEnvMap / buildBVH / intersect / castRay,
// MeshTriangle(filename, material, translation, zoom), Sphere(center, radius, material), Camera::lookAt) so that code
// written against the reference compiles against this header; everything that is hot runs in libmcpt_hip.so.
#pragma once
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mcpt.h"

// ============================================================================ vectors
// Minimal float vectors for the host side (the reference uses Eigen's Vector3f/Vector2f/Matrix3f for the same job).
// Only what scene assembly needs: storage, +, -, scalar *, dot/cross/normalized with Eigen's evaluation order.

struct Vector2f {
    float v[2] = {0.f, 0.f};
    Vector2f() = default;
    Vector2f(float x, float y) : v{x, y} {}
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float operator[](int i) const { return v[i]; }
};

struct Vector3f {
    float v[3] = {0.f, 0.f, 0.f};
    Vector3f() = default;
    Vector3f(float x, float y, float z) : v{x, y, z} {}
    static Vector3f Zero() { return Vector3f(0.f, 0.f, 0.f); }
    float x() const { return v[0]; }
    float y() const { return v[1]; }
    float z() const { return v[2]; }
    float operator[](int i) const { return v[i]; }
    float &operator[](int i) { return v[i]; }
    const float *data() const { return v; }
    Vector3f operator+(const Vector3f &o) const { return {v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]}; }
    Vector3f operator-(const Vector3f &o) const { return {v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2]}; }
    Vector3f operator*(float s) const { return {v[0] * s, v[1] * s, v[2] * s}; }
    float dot(const Vector3f &o) const { return v[0] * o.v[0] + (v[1] * o.v[1] + v[2] * o.v[2]); }
    Vector3f cross(const Vector3f &o) const {
        return {v[1] * o.v[2] - v[2] * o.v[1], v[2] * o.v[0] - v[0] * o.v[2], v[0] * o.v[1] - v[1] * o.v[0]};
    }
    float norm() const { return std::sqrt(dot(*this)); }
    Vector3f normalized() const {
        const float z = dot(*this);
        if (z > 0.f) {
            const float s = std::sqrt(z);
            return {v[0] / s, v[1] / s, v[2] / s};
        }
        return *this;
    }
};
inline Vector3f operator*(float s, const Vector3f &a) { return a * s; }

struct Matrix3f {  // row-major storage, column accessors as the reference uses them (Camera.hpp:21-23)
    float m[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    static Matrix3f Identity() { return Matrix3f(); }
    void setCol(int j, const Vector3f &c) {
        for (int i = 0; i < 3; ++i) m[3 * i + j] = c[i];
    }
    float operator()(int i, int j) const { return m[3 * i + j]; }
};

// ============================================================================ material

// Host-side Material: the fields and constructor defaults of the reference's Material (src/Material.hpp:157-167,245-257).
// The BSDF itself (sample/eval/pdf/fresnel) runs on the GPU (csrc/mcpt_device.h); the host only describes materials.

enum MaterialType { SMOOTH_CONDUCTOR, ROUGH_CONDUCTOR, SMOOTH_DIELECTRIC, ROUGH_DIELECTRIC };  // Material.hpp:13-18

class Material {
  public:
    MaterialType m_type;
    Vector3f m_emission;
    float iorA, iorB;
    bool textured = false;  // the reference leaves this uninitialised (Material.hpp:164); false is what its renders show
    bool isDirac;
    float roughness;
    Vector3f base_reflectance;

    explicit Material(MaterialType t = ROUGH_CONDUCTOR, Vector3f e = Vector3f(0, 0, 0)) {
        m_type = t;
        m_emission = e;
        isDirac = (t == SMOOTH_CONDUCTOR || t == SMOOTH_DIELECTRIC);
        iorA = 1.74;
        iorB = 0.1f;
        roughness = (t == ROUGH_DIELECTRIC) ? 0.2f : 1.f;
        base_reflectance = Vector3f(0, 0, 0);
    }
    MaterialType getType() const { return m_type; }
    Vector3f getEmission() const { return m_emission; }
    bool hasEmission() const { return m_emission.norm() > 1e-4f; }  // Material.hpp:262
};

// ============================================================================ scene objects

// Host-side scene objects: MeshTriangle and Sphere keep the reference's constructors (src/Triangle.hpp:83-86,
// src/Sphere.hpp:20-21); intersection, bounds and sampling live in the GPU library.


class Object {
  public:
    virtual ~Object() {}
    virtual bool hasEmit() const = 0;
    virtual float getArea() const = 0;
};

struct Triangle {  // src/Triangle.hpp:41-56: world-space vertices + texture coordinates
    Vector3f v0, v1, v2;
    Vector2f t0, t1, t2;
    float area() const { return (v1 - v0).cross(v2 - v0).norm() * 0.5f; }
};

// Per-face vertex stream of the first mesh of an OBJ file, as the reference's loader produces it.
bool load_obj_vertex_stream(const std::string &path, std::vector<Vector3f> &positions, std::vector<Vector2f> &texcoords);

class MeshTriangle : public Object {
  public:
    // Triangle.hpp:83-135: the loader's vertex stream is grouped by threes (the index buffer is ignored, so quads are
    // not triangulated); v = zoom * vert + translation; texture coordinates are copied only for textured materials.
    MeshTriangle(const std::string &filename, Material *mt = new Material(), const Vector3f &translation = Vector3f::Zero(),
                 float zoom = 1.0f);
    bool hasEmit() const override { return m->hasEmission(); }
    float getArea() const override { return area; }
    std::vector<Triangle> triangles;
    float area = 0.f;
    Material *m;
    bool loaded = false;
};

class Sphere : public Object {
  public:
    Vector3f center;
    float radius;
    Material *m;
    Sphere(const Vector3f &c, const float &r, Material *mt = new Material()) : center(c), radius(r), m(mt) {}
    bool hasEmit() const override { return m->hasEmission(); }
    float getArea() const override { return 4 * 3.141592653589793f * radius * radius; }
};

// ============================================================================ camera

// src/Camera.hpp:6-26 restated.

class Camera {
    Matrix3f orientation = Matrix3f::Identity();

  public:
    int width = 1280, height = 960;
    float fov = 40;
    bool useDOF = false;
    float focal_distance = 100;
    float aperture_radius = 5.0f;
    Camera(int w = 1280, int h = 960) : width(w), height(h) {}
    Vector3f position = {0, 0, 0};
    void lookAt(const Vector3f &target, const Vector3f &up = {0, 1, 0}) {
        const Vector3f forward = (target - position).normalized();
        const Vector3f left = up.cross(forward).normalized();
        const Vector3f new_up = forward.cross(left).normalized();
        orientation.setCol(0, left);
        orientation.setCol(1, new_up);
        orientation.setCol(2, forward);
    }
    Matrix3f getOrientation() const { return orientation; }
};

// ============================================================================ scene

// Host-side Scene: the public surface of the reference's Scene (src/Scene.hpp:24-152) over the C ABI.
//   Add / setRrRate / setDirectLightSample / enableShadow / loadEnvMap / backgroundColor / camera / buildBVH
// keep their names and meaning; buildBVH() flattens the objects and uploads them to the GPU (mcpt_scene_create);
// intersect() and castRay() forward to mcpt_intersect / mcpt_cast_rays.


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
    mcpt_scene *handle() const { return group ? mcpt_group_scene(group, 0) : gpu; }  // (with several GPUs: the replica on the first one)
    // More than one entry: the frame is rendered by all listed GPUs (mcpt_group_*: tile partition + RCCL merge inside the
    // library; main() stays single-threaded).  Call before buildBVH.  {0, 0} rehearses the schedule on one GPU.
    void setDevices(const std::vector<int> &d) { devices = d; }
    mcpt_group *groupHandle() const { return group; }
    mcpt_params params(int spp) const;
    mcpt_camera cameraDesc() const;
    // flat description (also used by the tests to compare with the Python assembly)
    void flatten(std::vector<mcpt_triangle> &tris, std::vector<mcpt_material> &mats, std::vector<mcpt_object> &objs) const;

    std::vector<Object *> objects;
    std::vector<Object *> lightsObjects;

  private:
    mcpt_scene *gpu = nullptr;
    mcpt_group *group = nullptr;
    std::vector<int> devices;
};

// ============================================================================ renderer

// src/Renderer.hpp:14-23 restated: same members, same defaults.


class Renderer {
  public:
    void Render(const Scene &scene);
    void setSpp(int s) { spp = s; }
    std::string path = "./output.png";
    // Pass-wise accumulation with a checkpoint file (the reference's render is all-or-nothing: two hours, Renderer.cpp:36-90).
    // With a checkpoint path the frame is rendered in chunks of `checkpoint_every` spp (mcpt_params.sample_offset / accumulate:
    // the same Philox keys and the same order of additions as one call, so the image is bit-identical), the running frame is
    // written after every chunk, and a later run with the same scene, size, spp and seed continues where the file stops.
    std::string checkpoint_path;
    int checkpoint_every = 64;
    int stop_after = 0;  // test hook: leave after this many spp have been rendered in total (simulates an interruption)

  private:
    int spp = 2048;
};
