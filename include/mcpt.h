/*
 * mcpt.h -- C ABI of the MI355X-native path-tracing hot path (libmcpt_hip.so).
 *
 * The reference (a single C++ executable) has no plugin/FFI interface; its only seam around the hot
 * path is the C++ call  Renderer::Render(const Scene&)  (src/Renderer.hpp:16, called once at
 * src/main.cpp:333) configured through Renderer::setSpp (Renderer.hpp:18) and the Scene setters
 * (Scene.hpp:104-119).  The entry points below are exactly what a binding for that seam needs; each
 * one names the reference interface it replaces.  Plain pointers and sizes only: no C++ or torch types.
 *
 * All functions return 0 on success or an mcpt_status code; mcpt_last_error() returns a thread-local
 * description of the last failure.  A scene handle may be used by one host thread at a time.
 * The library never falls back to a CPU path: without a usable HIP device every call fails with
 * MCPT_ERR_HIP.
 */
#ifndef MCPT_H
#define MCPT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MCPT_OK = 0,
    MCPT_ERR_ARG = 1,      /* bad argument / inconsistent description */
    MCPT_ERR_HIP = 2,      /* HIP runtime error (no device, launch failure, ...) */
    MCPT_ERR_OOM = 3,      /* host or device allocation failed */
    MCPT_ERR_LIMIT = 4,    /* a compiled-in limit was exceeded (BVH depth, light-tree depth) */
    MCPT_ERR_OVERFLOW = 5  /* a path outran the clamp stack (params.max_depth); frame is still returned */
} mcpt_status;

/* MaterialType, src/Material.hpp:13-18 */
enum { MCPT_SMOOTH_CONDUCTOR = 0, MCPT_ROUGH_CONDUCTOR = 1, MCPT_SMOOTH_DIELECTRIC = 2, MCPT_ROUGH_DIELECTRIC = 3 };
enum { MCPT_OBJ_MESH = 0, MCPT_OBJ_SPHERE = 1 };

/* One Triangle of a MeshTriangle (src/Triangle.hpp:41-56, 99-124): world-space vertices, texture coords. */
typedef struct {
    float v0[3], v1[3], v2[3];
    float t0[2], t1[2], t2[2];
} mcpt_triangle; /* 60 bytes */

/* src/Material.hpp:157-167 (+ ctor defaults :245-257): the fields the hot path reads. */
typedef struct {
    int32_t type;
    int32_t textured; /* checkerboard reflectance, Material.hpp:134-151 */
    float roughness, iorA, iorB;
    float base_reflectance[3];
    float emission[3];
} mcpt_material; /* 44 bytes */

/* One Scene::Add()-ed Object (src/Scene.hpp:104-109): a MeshTriangle or a Sphere (src/Sphere.hpp:10-21). */
typedef struct {
    int32_t kind;      /* MCPT_OBJ_MESH | MCPT_OBJ_SPHERE */
    int32_t material;  /* index into materials */
    int32_t first_tri; /* mesh: first triangle in `triangles`, file order */
    int32_t n_tri;     /* mesh: triangle count */
    float center[3];   /* sphere */
    float radius;      /* sphere */
} mcpt_object; /* 32 bytes */

/* Everything Scene holds once main() has assembled it (src/Scene.hpp:32-38,147-149). Borrowed pointers:
 * the library copies what it needs into HBM; the caller keeps ownership. */
typedef struct {
    int32_t n_objects;
    int32_t n_triangles;
    int32_t n_materials;
    int32_t env_w, env_h;          /* 0,0 => constant background colour (Scene.hpp:33,61-63) */
    float background[3];
    const mcpt_object *objects;    /* in Scene::Add order */
    const mcpt_triangle *triangles;
    const mcpt_material *materials;
    const float *env_pixels;       /* env_w*env_h*3 floats in [0,1], row-major (Scene.hpp:48-56) */
} mcpt_scene_desc;

/* src/Camera.hpp:6-26 after lookAt(). */
typedef struct {
    int32_t width, height;
    float fov;            /* degrees */
    float position[3];
    float orientation[9]; /* row-major 3x3, columns = left, up, forward (Camera.hpp:21-23) */
    int32_t use_dof;
    float focal_distance, aperture_radius;
} mcpt_camera; /* 72 bytes */

/* Renderer::spp (Renderer.hpp:18,22) + the private Scene knobs (Scene.hpp:25-28,110-119) + launch shape. */
typedef struct {
    int32_t spp;           /* samples per pixel rendered by this call */
    int32_t spp_total;     /* divisor of `framebuffer += rgb/spp` (Renderer.cpp:80); 0 => spp */
    int32_t sample_offset; /* index of this call's first sample (progressive accumulation); RNG key uses offset+k */
    float rr_rate;         /* caller applies min(rr, 0.99f) as Scene::setRrRate does */
    int32_t n_dir_sample;  /* Scene::n_dir_sample (the reference always runs 4) */
    int32_t enable_shadow;
    uint32_t seed;
    int32_t accumulate;    /* 0: framebuffer is overwritten for owned pixels; 1: added to */
    /* pixel-tile partition (multi-GPU): pixel (i,j) is owned iff ((j/tile)*ceil(W/tile) + i/tile) % nranks == rank */
    int32_t tile_size, rank, nranks;
    /* launch shape; 0 => library default */
    int32_t spp_per_pass;  /* samples per pixel in flight per pass (sizes the per-pass result buffer); 0: chosen by the library so that a pass carries many pools of samples */
    int32_t pool_paths;    /* wavefront pool capacity in channel-paths */
    int32_t max_depth;     /* clamp-stack levels per path; 0 => derived from rr_rate (P[deeper] < 1e-12) */
} mcpt_params;

typedef struct {
    uint64_t samples;       /* camera samples */
    uint64_t paths;         /* channel paths = 3 * samples */
    uint64_t vertices;      /* Scene::castRay invocations the reference would execute (Scene.cpp:85) */
    uint64_t shaded;        /* vertices that reached Material::sample (Scene.cpp:109) */
    uint64_t closest_rays;  /* closest-hit rays actually traced (primary once per sample + continuations) */
    uint64_t shadow_rays;   /* shadow rays actually traced (light samples with a non-zero contribution) */
    uint64_t ref_scene_rays;/* Scene::intersect calls the reference would make for the same work */
    uint64_t iterations;    /* wavefront iterations */
    uint64_t overflow_paths;/* paths cut by max_depth */
    double ms_total;        /* wall time of the call, host clock */
    double ms_trace_closest, ms_trace_shadow, ms_shade, ms_generate, ms_resolve; /* HIP-event sums per kernel class */
    uint64_t n_trace_closest, n_trace_shadow, n_shade, n_generate, n_resolve;    /* launches per kernel class */
    double ms_direct;   /* k_direct (direct-lighting kernel) */
    uint64_t n_direct;
    uint64_t direct_vertices; /* shaded vertices whose light samples were evaluated (the rest provably contribute 0) */
} mcpt_stats;

typedef struct mcpt_scene mcpt_scene;

/* Replaces Scene::Add + Scene::buildBVH (Scene.hpp:104-109, Scene.cpp:14-17) and MeshTriangle's per-mesh
 * BVHAccel (Triangle.hpp:128-134, BVH.cpp:27-93): builds the flattened BVH and uploads the scene to HBM
 * of the current (or `device`) GPU. */
int mcpt_scene_create(const mcpt_scene_desc *desc, int device, mcpt_scene **out);
void mcpt_scene_destroy(mcpt_scene *scene);

/* The same with an explicit choice of the tree builder (mcpt_scene_create: options == NULL).  Closest-hit results do not depend on
 * the tree (equal distances go to the larger primitive id); only rays that graze a box face within float rounding can differ.
 *   MCPT_BUILD_SAH        host, binned SAH over all primitives (default)
 *   MCPT_BUILD_REFERENCE  host, the reference's two-level median-split topology (BVH.cpp:27-93), flattened
 *   MCPT_BUILD_GPU_LBVH   on the device: Morton-code linear BVH (radix sort + Karras hierarchy + bottom-up refit); milliseconds
 *   MCPT_BUILD_GPU_PLOC   on the device: parallel locally-ordered clustering over the same Morton order (merges chosen by surface area:
 *                         a tree of near-SAH quality in a few milliseconds; search radius MCPT_PLOC_RADIUS, default 16)
 *                         instead of seconds for large scenes, at a lower tree quality
 * Fields left at MCPT_BUILD_DEFAULT / -1 take the environment overrides MCPT_BVH = sah | reference | lbvh | ploc and
 * MCPT_QUANT_NODES = 0 | 1, then the defaults. */
enum { MCPT_BUILD_DEFAULT = 0, MCPT_BUILD_SAH = 1, MCPT_BUILD_REFERENCE = 2, MCPT_BUILD_GPU_LBVH = 3, MCPT_BUILD_GPU_PLOC = 4 };
typedef struct {
    int32_t builder;  /* MCPT_BUILD_* */
    int32_t quantise; /* -1 automatic, 0 float nodes (64 B), 1 quantised nodes (32 B) */
    /* Node instancing (host SAH builder only).  Meshes that are translated copies of one another -- the 14 soldiers of main.cpp:248-271
     * are one OBJ file at 14 positions -- share ONE subtree of traversal nodes, entered with the ray origin shifted; every primitive test
     * still uses the object's own world-space triangle as the reference stores it (Triangle.hpp:99-124), so hits are unchanged.  It is
     * a memory feature (the 296 k-triangle scene: 56.9 -> 31.3 MB in HBM), measured slower than the plain tree on MI355X, so the default
     * (MCPT_INSTANCING_AUTO) leaves it off. */
    int32_t instancing; /* MCPT_INSTANCING_AUTO (0) | MCPT_INSTANCING_OFF (1) | MCPT_INSTANCING_ON (2); env override MCPT_INSTANCING = 0 | 1 */
    int32_t reserved[5];
} mcpt_build_options;
enum { MCPT_INSTANCING_AUTO = 0, MCPT_INSTANCING_OFF = 1, MCPT_INSTANCING_ON = 2 };
int mcpt_scene_create_ex(const mcpt_scene_desc *desc, int device, const mcpt_build_options *options, mcpt_scene **out);

/* Replaces the pixel/spp loop of Renderer::Render (Renderer.cpp:21-91): fb_host = W*H*3 floats,
 * row-major m = j*W + i, linear radiance averaged over spp -- what `framebuffer` holds at Renderer.cpp:91.
 * Blocking.  Tone map and PNG output (Renderer.cpp:95-109) stay with the caller. */
int mcpt_render(mcpt_scene *scene, const mcpt_camera *camera, const mcpt_params *params, float *fb_host,
                mcpt_stats *stats);

/* Same, with the framebuffer left in HBM (fb_device: W*H*3 floats on the scene's device) and all work issued
 * on `hip_stream` (a hipStream_t; NULL = default stream).  Used when the caller reduces frames with RCCL. */
int mcpt_render_device(mcpt_scene *scene, const mcpt_camera *camera, const mcpt_params *params, float *fb_device,
                       void *hip_stream, mcpt_stats *stats);

/* Replaces Scene::intersect (Scene.hpp:128, Scene.cpp:19-21) for a list of rays (host pointers; n*3 floats each).
 * out_t: hit distance as the reference's double Intersection::distance (DBL_MAX on a miss);
 * out_prim: global primitive id (triangle index, or n_triangles + object index for a sphere; -1 on a miss). */
int mcpt_intersect(mcpt_scene *scene, int64_t n, const float *origins, const float *dirs, double *out_t,
                   int32_t *out_prim);

/* Replaces Scene::castRay(ray, 0, channel) (Scene.hpp:131, Scene.cpp:85-184) for a list of rays (host pointers).
 * The RNG of ray i is keyed by (params->seed, pixel[i], sample[i], channel[i]). */
int mcpt_cast_rays(mcpt_scene *scene, const mcpt_params *params, int64_t n, const float *origins, const float *dirs,
                   const uint32_t *pixel, const uint32_t *sample, const int32_t *channel, float *out);

/* Camera ray generation of Renderer.cpp:44-76 for (pixel m, sample k) pairs (host pointers); origins/dirs: n*3 floats. */
int mcpt_camera_rays(mcpt_scene *scene, const mcpt_camera *camera, uint32_t seed, int64_t n, const uint32_t *pixel,
                     const uint32_t *sample, float *origins, float *dirs);

/* Tone map of Renderer.cpp:95-103 on the GPU: rgba[4i + c] = (unsigned char) clamp(0, 255, 255 * pow(fb[3i + c], 0.45f)), alpha 255,
 * NaN -> 255 as the reference's std::min/std::max clamp gives.  std::pow is the library's own plain-IEEE pow (csrc/mcpt_fmath.h), within
 * an ulp of any libm's.  Optional: the float frame of mcpt_render is the boundary's product; callers may keep their own tone map. */
int mcpt_tonemap(mcpt_scene *scene, const float *fb_host, int64_t n_pixels, uint8_t *rgba_host);
int mcpt_tonemap_device(mcpt_scene *scene, const float *fb_device, int64_t n_pixels, uint8_t *rgba_device, void *hip_stream);

/* ---- Multi-GPU inside the boundary.  The caller stays single-threaded like the reference's main() (Renderer::Render blocks,
 * main.cpp:333): a group holds one replica of the scene per device; mcpt_group_render partitions the frame into interleaved
 * tiles over the devices (tile_size of `params`, default 32; its rank/nranks fields are ignored), drives every device from its
 * own host thread, sums the per-device frames into the first device's with one RCCL ncclReduce over xGMI and returns the
 * merged frame in fb_host.  The result is bit-identical to mcpt_render on one GPU (disjoint pixels, same Philox keys).
 * `devices`: distinct HIP device indices; as a rehearsal on a one-GPU box every entry may name the SAME device (the merge is
 * then a kernel on that device; RCCL is neither loaded nor needed).  Errors of these three calls: mcpt_group_last_error(). */
typedef struct mcpt_group mcpt_group;
int mcpt_group_create(const mcpt_scene_desc *desc, int n_devices, const int *devices, mcpt_group **out);
int mcpt_group_render(mcpt_group *group, const mcpt_camera *camera, const mcpt_params *params, float *fb_host, mcpt_stats *stats);
int mcpt_group_size(const mcpt_group *group);
/* What mcpt_group_create spent: the scene is flattened and its tree built once (build_ms) while every device is brought up on a helper
 * thread (init_ms_max), then one thread per device copies it (upload_ms_max); setup_ms is the wall clock of the whole call. */
typedef struct {
    int32_t n_devices;
    int32_t uses_rccl; /* 1: distinct devices, the frames are merged by one ncclReduce; 0: every entry names one device (rehearsal) */
    double build_ms, upload_ms_max, init_ms_max, setup_ms;
} mcpt_group_info;
int mcpt_group_get_info(const mcpt_group *group, mcpt_group_info *info);
/* The replica on the index-th device of the group (borrowed: it lives as long as the group), for calls that take a scene, e.g.
 * mcpt_tonemap after mcpt_group_render.  NULL when out of range. */
mcpt_scene *mcpt_group_scene(mcpt_group *group, int index);
void mcpt_group_destroy(mcpt_group *group);
const char *mcpt_group_last_error(void);

/* Scene statistics for reporting (BVH nodes, tree height, bytes resident in HBM). */
typedef struct {
    int32_t n_nodes, bvh_height, n_lights, n_prims;
    uint64_t scene_bytes;
    double build_ms;  /* flattening + BVH build (the data producer of BVHAccel::recursiveBuild, BVH.cpp:27-93) */
    double upload_ms; /* host -> HBM copies */
    int32_t builder;  /* 0 host binned SAH, 1 host reference topology (median split), 2 GPU LBVH, 3 GPU PLOC */
    int32_t quantised;/* 1: 32-byte nodes with 16-bit boxes are traversed */
    int32_t n_instances; /* objects whose traversal nodes are shared with a prototype (0: plain tree) */
    int32_t lds_resident; /* 1: the scene is small enough for the kernels that copy nodes, triangles, spheres and light tables into LDS */
    double init_ms;   /* first use of the device by this process (context creation, load of the library's code objects), run on a helper
                         thread beside the host build; ~0 for every later scene.  Not part of build_ms / upload_ms */
} mcpt_scene_info;
int mcpt_scene_get_info(const mcpt_scene *scene, mcpt_scene_info *info);

/* Host-only diagnostic (no GPU needed): builds the traversal tree of `desc` exactly as mcpt_scene_create does and copies it
 * out, so that the builder (the data producer of BVHAccel::recursiveBuild, BVH.cpp:27-93) can be checked on any machine.
 * Call with boxes == NULL to get the counts, then with arrays of n_nodes entries:
 *   boxes[n][12]    float  {lmin.xyz, lmax.xyz, rmin.xyz, rmax.xyz} of the two children
 *   children[n][2]  int32  child >= 0: inner node index; < 0: leaf, primitive id = ~child
 *   qboxes[n][12]   uint16 the same boxes on the 16-bit grid (only written when info->quantised)
 * Primitive ids: triangle index, or n_triangles + object index for a sphere.
 * With instancing, a leaf index (~child) >= n_leaf_prims is instance k = index - n_leaf_prims: the subtree at inst_root_first[k][0],
 * whose boxes are in the prototype's position (this object's position = prototype + inst_shift[k]) and whose leaves hold LOCAL
 * triangle indices (global id = inst_root_first[k][1] + local).  inst_* may be NULL. */
typedef struct {
    int32_t n_nodes, root, stack_entries, quantised;
    float root_min[3], root_max[3];
    float q_origin[3], q_cell[3]; /* grid coordinate q <-> q_origin + q * q_cell */
    int32_t n_instances, n_leaf_prims;
} mcpt_bvh_info;
int mcpt_bvh_dump(const mcpt_scene_desc *desc, mcpt_bvh_info *info, float *boxes, int32_t *children, uint16_t *qboxes,
                  float *inst_shift, int32_t *inst_root_first);
/* The same arrays downloaded from a live scene (whatever built its tree, the GPU builder included). */
int mcpt_scene_dump_bvh(mcpt_scene *scene, mcpt_bvh_info *info, float *boxes, int32_t *children, uint16_t *qboxes,
                        float *inst_shift, int32_t *inst_root_first);

/* Diagnostic: evaluates the path's transcendental functions (csrc/mcpt_fmath.h: the library's own plain-IEEE sin / cos /
 * atan2 / acos, used where the reference calls libm at Material.hpp:117-118, Renderer.cpp:59-60, Sphere.hpp:66-67,
 * Scene.hpp:66-67) ON THE DEVICE for n host floats, so that tests can check that kernels and a CPU build of the same
 * header agree bit for bit.  kind: 0 sin(x), 1 cos(x), 2 atan2(x, y), 3 acos(x), 4 pow(x, y), 5 tone-map byte of x (as a float);
 * y may be NULL unless kind is 2 or 4. */
int mcpt_debug_fmath(mcpt_scene *scene, int kind, int64_t n, const float *x, const float *y, float *out);

/* Diagnostic: evaluates the device's Material functions (csrc/mcpt_device.h, following Material.hpp:26-151,178-408) for n rows, so
 * that tests can compare them one by one -- not only through whole paths -- with the CPU restatement.  in: 13 floats per row
 * {a.xyz, b.xyz, c.xyz, uv.xy, u1, u2}; sel: 3 ints per row {material index, channel 0..2, is_reflect}; out: 4 floats per row.
 * kind 0 Material::eval(wi = a, wo = b, N = c, uv), 1 Material::pdf(a, b, c), 2 fresnel(I = a, N = b), 3 sample(N = a; u1, u2) -> xyz,
 * 4 refract(I = a, N = b) -> xyz, 5 the fused eval + pdf of the shading kernel -> {eval, pdf}, 6 reflect(I = a, N = b) -> xyz. */
int mcpt_debug_material(mcpt_scene *scene, int kind, int64_t n, const float *in, const int32_t *sel, float *out);

/* Diagnostic: the device's Scene::sampleLight (Scene.cpp:23-37 with MeshTriangle::Sample, BVHAccel::getSample, Triangle::Sample; kind 0:
 * in = 4 uniforms per row {light choice, triangle pick, x, y}, out = 10 floats per row {point, normal, emission, pdf}) and
 * Scene::sampleEnv (Scene.hpp:60-99; kind 1: in = a direction per row, out = rgb) on arrays. */
int mcpt_debug_scene(mcpt_scene *scene, int kind, int64_t n, const float *in, float *out);

/* Diagnostic: counters of the checking build (libmcpt_hip_check.so, compiled with -DMCPT_CHECK_DIRECT_SKIP; the traversal
 * entries are filled only by a -DMCPT_TRAVERSAL_STATS build); all zero in the product build.
 *   out[0..5]   closest-hit rays: rays, node visits, primitive tests, hits, 64 x wave iterations, -
 *   out[8..13]  shadow rays: rays, node visits, primitive tests, occluded, 64 x wave iterations, found in the window
 *   out[14]     light samples evaluated at vertices the product would have skipped as "provably zero" (direct_is_zero)
 *   out[15]     how many of those had a non-zero contribution (must be 0) */
int mcpt_debug_counters(mcpt_scene *scene, uint64_t out[16]);

const char *mcpt_last_error(void);
const char *mcpt_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MCPT_H */
