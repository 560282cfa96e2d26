/*
 * mcpt_oracle.h -- TEST INFRASTRUCTURE ONLY (parity oracle + timed CPU baseline).
 *
 * CPU restatement of the reference's rendering hot path
 * (Renderer::Render -> Scene::castRay -> BVHAccel::Intersect -> Material::sample/eval/pdf).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libmcpt_hip.so) never links, includes or calls anything declared here.
 *
 * Pinning status: the reference has no tests and no golden vectors, and it cannot be built in
 * this image (it needs Eigen3, which is neither vendored nor installed).  The only artefact the
 * reference holds for this path is cornellbox_demo.png (DEMO scene, 384x384); the oracle is
 * pinned statistically against that image (tests/test_oracle_golden.py).  Everything else is
 * "parity unpinned" at the bit level -- see DESIGN.md section 3.
 */
#ifndef MCPT_ORACLE_H
#define MCPT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Material types, Material.hpp:13-18 */
enum { ORC_SMOOTH_CONDUCTOR = 0, ORC_ROUGH_CONDUCTOR = 1, ORC_SMOOTH_DIELECTRIC = 2, ORC_ROUGH_DIELECTRIC = 3 };
enum { ORC_OBJ_MESH = 0, ORC_OBJ_SPHERE = 1 };

typedef struct {
    float v0[3], v1[3], v2[3]; /* world-space vertices, Triangle.hpp:43 */
    float t0[2], t1[2], t2[2]; /* texture coords (zeros unless the material is textured), Triangle.hpp:45,115-122 */
} orc_triangle;

typedef struct {
    int32_t type;     /* MaterialType */
    int32_t textured; /* Material.hpp:164 (reference leaves it uninitialised; callers pass 0 unless set) */
    float roughness, iorA, iorB;
    float base_reflectance[3];
    float emission[3];
} orc_material;

typedef struct {
    int32_t kind;      /* ORC_OBJ_MESH | ORC_OBJ_SPHERE */
    int32_t material;  /* index into materials */
    int32_t first_tri; /* mesh: first triangle, file order */
    int32_t n_tri;     /* mesh: triangle count */
    float center[3];   /* sphere */
    float radius;      /* sphere */
} orc_object;

typedef struct {
    int32_t n_objects;
    int32_t n_triangles;
    int32_t n_materials;
    int32_t env_w, env_h; /* 0,0 => constant background (Scene.hpp:33,61-63) */
    float background[3];
    const orc_object *objects; /* in Scene::Add order (Scene.hpp:104-109) */
    const orc_triangle *triangles;
    const orc_material *materials;
    const float *env_pixels; /* env_w*env_h*3 floats in [0,1], row-major (Scene.hpp:48-56) */
} orc_scene_desc;

typedef struct {
    int32_t width, height;
    float fov; /* degrees */
    float position[3];
    float orientation[9]; /* row-major 3x3; columns = left, up, forward (Camera.hpp:21-23) */
    int32_t use_dof;
    float focal_distance, aperture_radius;
} orc_camera;

typedef struct {
    int32_t spp;
    float rr_rate;        /* caller applies min(rr, 0.99f) as Scene.hpp:110-113 */
    int32_t n_dir_sample; /* Scene.hpp:28 (the reference always runs 4) */
    int32_t enable_shadow;
    uint32_t seed;
    int32_t n_threads; /* <=0: omp default */
    /* pixel-tile partition for multi-rank rendering: pixel (i,j) belongs to this rank iff
     * ((j/tile)*ceil(W/tile) + (i/tile)) % nranks == rank.  nranks<=1 => all pixels. */
    int32_t tile_size, rank, nranks;
} orc_params;

typedef struct {
    uint64_t samples;    /* camera samples rendered */
    uint64_t scene_rays; /* Scene::intersect calls (Scene.cpp:19) */
    uint64_t vertices;   /* Scene::castRay invocations (Scene.cpp:85) */
    uint64_t node_visits;/* Bounds3::IntersectP calls */
    uint64_t tri_tests;  /* Triangle::getIntersection calls */
    double seconds;
} orc_stats;

typedef struct orc_scene orc_scene;

int orc_scene_create(const orc_scene_desc *desc, orc_scene **out);
void orc_scene_destroy(orc_scene *s);

/* Renderer::Render restated (Renderer.cpp:21-91): fb = W*H*3 floats, linear radiance averaged
 * over spp, row-major m = j*W + i.  Pixels not owned by (rank,nranks) are left untouched. */
int orc_render(const orc_scene *s, const orc_camera *cam, const orc_params *p, float *fb, orc_stats *stats);

/* Scene::intersect (Scene.cpp:19-21) on a ray list.  out_t = distance (double, 1.797e308 on miss),
 * out_prim = global primitive id (triangles in desc order, then spheres as n_triangles + object index; -1 miss). */
int orc_intersect(const orc_scene *s, int64_t n, const float *origins, const float *dirs, double *out_t,
                  int32_t *out_prim);

/* Primary visibility of a frame: out_prim[m * spp + k] = primitive the camera ray of pixel m, sample k hits (-1: none). */
int orc_primary_hits(const orc_scene *s, const orc_camera *cam, uint32_t seed, int32_t spp, int32_t *out_prim);

/* Scene::castRay(ray, 0, channel) (Scene.cpp:85-184) on a ray list; RNG keyed by (seed, pixel[i], sample[i], channel[i]). */
int orc_cast_rays(const orc_scene *s, const orc_params *p, int64_t n, const float *origins, const float *dirs,
                  const uint32_t *pixel, const uint32_t *sample, const int32_t *channel, float *out);

/* Camera ray generation (Renderer.cpp:44-76) for (pixel m, sample k): writes origin[3], dir[3]. */
void orc_camera_ray(const orc_camera *cam, uint32_t seed, uint32_t m, uint32_t k, float *origin, float *dir);

/* Material KATs (Material.hpp:285-408, 198-242, 268-281). */
float orc_material_eval(const orc_material *m, const float *wi, const float *wo, const float *n, int channel,
                        const float *uv, int is_reflect);
float orc_material_pdf(const orc_material *m, const float *wi, const float *wo, const float *n, int channel,
                       int is_reflect);
float orc_material_fresnel(const orc_material *m, const float *I, const float *N, int channel);
void orc_material_sample(const orc_material *m, const float *n, float u1, float u2, float *out);
void orc_material_refract(const orc_material *m, const float *I, const float *N, int channel, float *out);

/* Scene::sampleLight (kind 0: 4 uniforms per row -> {coords, normal, emit, pdf}, 10 floats) and Scene::sampleEnv (kind 1: a direction
 * per row -> rgb) on arrays. */
void orc_scene_function(const orc_scene *s, int kind, int64_t n, const float *in, float *out);

/* Philox4x32-10 (Salmon et al. 2011), for KAT tests. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]);

/* The shared transcendental functions (csrc/mcpt_fmath.h) on arrays: kind 0 sin(x), 1 cos(x), 2 atan2(x, y), 3 acos(x),
 * 4 pow(x, y), 5 tone-map byte of x (as a float). */
void orc_fmath(int kind, int64_t n, const float *x, const float *y, float *out);

/* Tone map (Renderer.cpp:95-103): fb (W*H*3 float) -> rgba8 (W*H*4). */
void orc_tonemap(const float *fb, int64_t npixels, uint8_t *rgba);

#ifdef __cplusplus
}
#endif
#endif
