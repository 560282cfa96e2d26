// Internal layouts shared by the host scene builder, the kernels and the C-ABI layer.
// Everything here is resident in HBM for the lifetime of an mcpt_scene.
#pragma once
#include <cstdint>
#include <vector>

#include "../../include/mcpt.h"

namespace mcpt {

constexpr float kEps = 1e-4f;                 // EPSILON, reference Renderer.cpp:15
constexpr int kNoChild = 0x7fffffff;          // filler in placeholder records that are never traversed
constexpr int kMaxBvhHeight = 48;             // traversal stack entries per lane (LDS)
constexpr int kMaxLightTreeDepth = 64;

// BVH2 node, 64 bytes = four 16-byte loads.  Both children's boxes live in the parent so that one
// fetch decides both descents.  child >= 0: inner node index; child < 0: leaf, primitive id = ~child; every inner node
// has two children.  Default tree: one binned-SAH BVH over all primitives; MCPT_BVH=reference keeps the reference's
// two-level median-split topology (BVH.cpp:27-93) flattened into one array (a scene-level leaf that is a mesh is
// replaced by the mesh's own root).
struct alignas(16) Node {
    float lmin[3], lmax[3];
    float rmin[3], rmax[3];
    int32_t left, right;
    int32_t pad[2];
};
static_assert(sizeof(Node) == 64, "Node must be 64 bytes");

// The same node with both child boxes quantised to a 16-bit grid over the scene's root box: 32 bytes = two 16-byte loads
// instead of four.  The traversal kernels are bound by the rate of per-lane memory requests (each lane fetches its own node),
// not by arithmetic, so halving the requests per node visit is what counts.  The quantised boxes are conservative: minima are
// rounded down and maxima up, by one extra cell, so a quantised box contains the exact one and traversal can only visit
// more, never miss (the hits themselves come from the exact primitive tests).
//   words 0..5: u16 pairs {lmin.x,lmin.y} {lmin.z,lmax.x} {lmax.y,lmax.z} {rmin.x,rmin.y} {rmin.z,rmax.x} {rmax.y,rmax.z}
//   words 6,7:  left, right child references
struct alignas(16) QNode {
    uint32_t w[6];
    int32_t left, right;
};
static_assert(sizeof(QNode) == 32, "QNode must be 32 bytes");

// Traversal-side triangle record (48 bytes, three 16-byte loads): what Triangle::getIntersection reads
// (Triangle.hpp:222-252): v0, e1, e2.
// TriGeom::mat_bits / SphereRec::mat_bits / the hit record's fourth word: material index and two flags of the material, so that k_shade
// knows them with the hit instead of one dependent load later
constexpr uint32_t kMatEmissive = 0x80000000u;  // the material emits (Material::hasEmission)
constexpr uint32_t kMatTextured = 0x40000000u;  // triangles only: the material is textured, the shading needs the barycentrics (Triangle.hpp:248)
constexpr uint32_t kMatIndexMask = 0x3fffffffu;
struct alignas(16) TriGeom {
    float v0[3];
    float e1x;
    float e1yz[2];
    float e2xy[2];
    float e2z;
    uint32_t mat_bits;  // material index | kMatTextured | kMatEmissive (carried into the hit record: saves k_shade its dependent material loads)
    int32_t pad[2];
};
static_assert(sizeof(TriGeom) == 48, "TriGeom must be 48 bytes");

// Shading-side triangle record (48 bytes): normal, material, texture coords (Triangle.hpp:45-48).
struct alignas(16) TriShade {
    float n[3];
    int32_t mat;
    float t0[2], t1[2];
    float t2[2];
    int32_t pad[2];
};
static_assert(sizeof(TriShade) == 48, "TriShade must be 48 bytes");

struct alignas(16) SphereRec {  // Sphere.hpp:14-18
    float c[3];
    float radius;
    float radius2;
    int32_t mat;
    uint32_t mat_bits;  // material index | kMatEmissive
    int32_t pad;
};
static_assert(sizeof(SphereRec) == 32, "SphereRec must be 32 bytes");

struct alignas(16) MaterialRec {  // Material.hpp:157-167
    int32_t type, textured, isDirac, hasEmission;
    float roughness, iorA, iorB, pad0;
    float refl[3], pad1;
    float emit[3], pad2;
    // Per channel (WaveLen.hpp:7-18): getIor(lambda) = iorA + iorB / lambda^2 (Material.hpp:178-183) and the float (1. / ior) of
    // Material.hpp:299,318,360,393, evaluated once on the host with the reference's expressions instead of per vertex and light
    // sample on the device (a float and a double division each time)
    float ior[3], pad3;
    float inv_ior[3], pad4;
};
static_assert(sizeof(MaterialRec) == 96, "MaterialRec must be 96 bytes");

// Light table entry: one per emissive object, in Scene::Add order (Scene.hpp:106-108).
struct alignas(16) LightRec {
    float area;        // Object::getArea()
    int32_t kind;      // MCPT_OBJ_MESH / MCPT_OBJ_SPHERE
    int32_t root;      // mesh: index of the root LightNode; sphere: sphere index
    int32_t mat;
    float root_area;   // mesh: BVHBuildNode::area of the mesh root (BVH.cpp:45-46,91)
    int32_t pad[3];
};
static_assert(sizeof(LightRec) == 32, "LightRec must be 32 bytes");

// An instanced object (csrc/mcpt_scene.cpp: instancing): the traversal NODES of its prototype's subtree are shared, shifted by
// `shift` (box tests run with the ray origin moved by -shift); the primitive tests still read this object's own exact
// world-space triangles, tri_geom[first_tri + local index], so hits are those of the un-instanced scene.
struct alignas(16) InstRec {
    float shift[3];     // object position minus prototype position
    int32_t root;       // child reference of the prototype subtree's root (inner node index)
    int32_t first_tri;  // this object's first triangle (global primitive id of local index 0)
    int32_t pad[3];
};
static_assert(sizeof(InstRec) == 32, "InstRec must be 32 bytes");

// Area tree of a light mesh, for BVHAccel::getSample (BVH.cpp:118-129).
struct alignas(16) LightNode {
    float left_area;   // node->left->area
    int32_t left;      // child LightNode index, or ~(LightTri index) when the child is a leaf
    int32_t right;
    float area;        // node->area
};
static_assert(sizeof(LightNode) == 16, "LightNode must be 16 bytes");

// Triangle of an emissive mesh, for Triangle::Sample (Triangle.hpp:71-76): it interpolates the STORED
// vertices, which cannot be rebuilt exactly from TriGeom's edges, so light triangles keep their own copy.
struct alignas(16) LightTri {
    float v0[3], v1[3], v2[3];
    float n[3];
    float area;
    int32_t prim;  // global primitive id of this triangle (k_direct tests the sampled triangle itself first)
    int32_t pad[2];
};
static_assert(sizeof(LightTri) == 64, "LightTri must be 64 bytes");

// Host-side product of scene construction, ready to be copied to the device.
struct HostScene {
    std::vector<Node> nodes;
    std::vector<QNode> qnodes;            // quantised copy of `nodes` (empty when the grid would be too coarse)
    float q_origin[3] = {0, 0, 0}, q_cell[3] = {1, 1, 1};  // box coordinate = q_origin + q * q_cell
    std::vector<TriGeom> tri_geom;
    std::vector<TriShade> tri_shade;
    std::vector<SphereRec> spheres;       // indexed by object index (entries for meshes unused)
    std::vector<MaterialRec> materials;
    std::vector<LightRec> lights;
    std::vector<LightNode> light_nodes;
    std::vector<LightTri> light_tris;
    float root_min[3], root_max[3];       // scene root bounds (tested once per ray, BVH.cpp:105)
    int32_t root;                         // root child reference (inner index or leaf)
    int32_t n_triangles = 0, n_objects = 0;
    int32_t height = 0;
    int32_t builder = 0;                  // 0 host binned SAH, 1 host reference topology, 2 GPU LBVH, 3 GPU PLOC (mcpt_scene_info::builder)
    std::vector<int32_t> sphere_objects;  // object index of every sphere object (the GPU builder's primitive list)
    std::vector<InstRec> instances;       // instanced objects (empty: plain single-level tree); leaf reference ~(n_leaf_prims + k)
    int32_t n_leaf_prims = 0;             // n_triangles + n_objects: leaf indices below it are primitives, from it on instances
    float background[3];
    int32_t env_w = 0, env_h = 0;
    std::vector<float> env;
    float light_area_sum = 0.f;
    float light_center[3] = {0.f, 0.f, 0.f};  // bounding sphere of every emissive primitive
    float light_radius = 0.f;
};

// Builder choice after the environment overrides have been applied (mcpt_build_options, include/mcpt.h).
struct BuildChoice {
    int32_t builder = MCPT_BUILD_SAH;  // MCPT_BUILD_SAH | MCPT_BUILD_REFERENCE | MCPT_BUILD_GPU_LBVH | MCPT_BUILD_GPU_PLOC
    int32_t quantise = -1;             // -1 automatic, 0 never, 1 always
    int32_t instancing = -1;           // -1 automatic (large scenes with repeated meshes), 0 never, 1 whenever a mesh repeats
    int32_t ploc_radius = 16;          // MCPT_BUILD_GPU_PLOC: clusters searched on either side (MCPT_PLOC_RADIUS)
    int32_t ploc_top = 16384;          // ... and the cluster count at which the host's binned SAH takes over (MCPT_PLOC_TOP; 0: never;
                                       // at most a sixteenth of the primitives)
};
BuildChoice resolve_build_choice(const mcpt_build_options *opt);

// Binned-SAH top of a tree over the clusters a device builder stopped at (csrc/mcpt_scene.cpp; used by the PLOC builder).
void build_sah_over_clusters(int m, const float *cmin4, const float *cmax4, const int32_t *cref, const int32_t *clevels, int node_base,
                             std::vector<Node> &out, int32_t &root_ref, int32_t &height);

// Builds the flattened scene.  Returns MCPT_OK or an error code and fills `err`.  With MCPT_BUILD_GPU_LBVH / _PLOC the traversal tree is left
// out (nodes empty, root/height unset): the caller builds it on the device (csrc/mcpt_lbvh.hip); everything else is filled.
int build_host_scene(const mcpt_scene_desc &d, HostScene &out, const char **err, const BuildChoice &choice);

}  // namespace mcpt
