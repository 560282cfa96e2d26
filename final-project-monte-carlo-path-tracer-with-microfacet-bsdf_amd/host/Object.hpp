// Host-side scene objects: MeshTriangle and Sphere keep the reference's constructors (src/Triangle.hpp:83-86,
// src/Sphere.hpp:20-21); intersection, bounds and sampling live in the GPU library.
#pragma once
#include <string>
#include <vector>

#include "Material.hpp"

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
