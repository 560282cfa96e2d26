/*
 * mcpt_oracle.c -- TEST INFRASTRUCTURE ONLY.  See mcpt_oracle.h for the usage rules.
 *
 * A structure-for-structure CPU restatement (plain C11 + OpenMP) of the reference's hot path.
 * Every function cites the reference file:line it follows (paths relative to the reference's
 * src/ directory).  It keeps the reference's shape on purpose: recursive castRay, a pointer BVH
 * with one primitive per leaf that visits BOTH children with no t-pruning, two-level (scene BVH
 * over objects, per-mesh BVH over triangles), three independent single-channel paths per sample.
 *
 * Differences from the reference, all deliberate and documented in DESIGN.md section 3:
 *   - RNG: std::mt19937 per thread (global.hpp:14,49-53) is replaced by counter-based
 *     Philox4x32-10 keyed by (seed, pixel, sample, stream, depth, block), so that a sample's
 *     draws do not depend on thread scheduling and the GPU path can reproduce them.
 *   - Hit ties (equal double distance) go to the larger global primitive id; the reference's
 *     rule (right subtree wins, BVH.cpp:115) depends on an unstable std::sort (BVH.cpp:57-73).
 *   - Uninitialised reads (Material::textured, Triangle::t0-2, Sphere tcoords) read as 0.
 * Arithmetic classes follow the reference expression by expression: float vectors, the double
 * det/u/v/t chain in Triangle::getIntersection, double literals where the source has them.
 * Eigen (absent here) only contributes single IEEE float operations plus the 3-term dot, which
 * Eigen evaluates as x0*y0 + (x1*y1 + x2*y2); build with -ffp-contract=off.
 */
#define _POSIX_C_SOURCE 199309L
#include "mcpt_oracle.h"

/* sin/cos/atan2/acos: the reference calls the platform libm (Material.hpp:117-118, Renderer.cpp:59-60, Sphere.hpp:66-67,
 * Scene.hpp:66-67), whose last bit differs from platform to platform.  The product's own plain-IEEE implementation is used
 * here as well, so that the checker and the GPU kernels make the same rounding decisions (tests/test_fmath.py pins it to
 * <= 1 ulp of glibc). */
#include "../final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd/csrc/mcpt_fmath.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ constants */
static const float EPSILON = 1e-4f;             /* Renderer.cpp:15 */
#define PI_F 3.141592653589793f                 /* global.hpp:8-9: M_PI redefined as a float */

/* ------------------------------------------------------------------ vectors */
typedef struct { float x, y, z; } v3;
typedef struct { float x, y; } v2;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 ld3(const float *p) { return V3(p[0], p[1], p[2]); }
static inline v3 add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
static inline v3 mulf(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 divf(v3 a, float s) { return V3(a.x / s, a.y / s, a.z / s); }
/* Eigen fixed-size redux for 3 terms: x0*y0 + (x1*y1 + x2*y2) */
static inline float dot(v3 a, v3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
static inline v3 cross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float norm(v3 a) { return sqrtf(dot(a, a)); }
/* Eigen normalized(): v / sqrt(squaredNorm) when squaredNorm > 0, else v unchanged */
static inline v3 normalized(v3 a) {
    float z = dot(a, a);
    if (z > 0.0f) return divf(a, sqrtf(z));
    return a;
}
static inline float comp(v3 a, int c) { return c == 0 ? a.x : (c == 1 ? a.y : a.z); }

/* global.hpp:16-18 with std::min/std::max semantics (NaN v -> hi) */
static inline float std_minf(float a, float b) { return (b < a) ? b : a; }
static inline float std_maxf(float a, float b) { return (a < b) ? b : a; }
static inline float clampf(float lo, float hi, float v) { return std_maxf(lo, std_minf(hi, v)); }

/* ------------------------------------------------------------------ RNG */
static inline void mulhilo(uint32_t a, uint32_t b, uint32_t *hi, uint32_t *lo) {
    uint64_t p = (uint64_t)a * (uint64_t)b;
    *hi = (uint32_t)(p >> 32);
    *lo = (uint32_t)p;
}

void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0, lo0, hi1, lo1;
        mulhilo(0xD2511F53u, c0, &hi0, &lo0);
        mulhilo(0xCD9E8D57u, c2, &hi1, &lo1);
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

/* Replaces get_random_float (global.hpp:49-53).  One block = 4 uniform floats in [0,1).
 * key = (seed, pixel m); counter = (sample k, depth, block, stream); stream 0..2 = R,G,B path, 3 = camera.
 * Per castRay invocation (Appendix A.9 of SURVEY.md): block 0 = (Xi.x, Xi.y, rr, rd_flect);
 * block 1+i = light sample i: (light choice, triangle pick, x, y). */
typedef struct { uint32_t seed, pixel, sample, stream; } rng_key;

static inline void rng_block(const rng_key *k, uint32_t depth, uint32_t block, float u[4]) {
    uint32_t ctr[4] = {k->sample, depth, block, k->stream};
    uint32_t key[2] = {k->seed, k->pixel};
    uint32_t o[4];
    orc_philox4x32_10(ctr, key, o);
    for (int i = 0; i < 4; ++i) u[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
}

/* ------------------------------------------------------------------ scene types */
typedef struct { v3 pMin, pMax; } bounds3;

typedef struct material {
    int type, textured, isDirac;
    float roughness, iorA, iorB;
    v3 base_reflectance, emission;
    int has_emission; /* Material.hpp:262 */
} material;

enum { K_TRIANGLE = 0, K_SPHERE = 1, K_MESH = 2 };

struct bvh_node;

typedef struct object {
    int kind;
    int prim_id; /* global primitive id for K_TRIANGLE / K_SPHERE */
    const material *m;
    /* triangle: Triangle.hpp:43-47 */
    v3 v0, v1, v2, e1, e2, normal;
    v2 t0, t1, t2;
    /* sphere: Sphere.hpp:14-18 */
    v3 center;
    float radius, radius2;
    float area;
    /* mesh: Triangle.hpp:200-211 */
    bounds3 bounding_box;
    struct object *tris;
    int n_tris;
    struct bvh_node *bvh_root;
} object;

typedef struct bvh_node { /* BVH.hpp:53-69 */
    bounds3 bounds;
    struct bvh_node *left, *right;
    object *obj;
    float area;
} bvh_node;

struct orc_scene {
    int n_objects, n_materials, n_triangles;
    object *objects;
    object **lights; /* Scene::lightsObjects, Scene.hpp:106-108 */
    int n_lights;
    material *materials;
    bvh_node *bvh_root; /* Scene::bvh, Scene.cpp:14-17 */
    v3 background;      /* Scene.hpp:33 */
    int use_env, env_w, env_h;
    float *env_pixels;
};

typedef struct { /* Intersection.hpp:12-29 */
    int happened;
    v3 coords;
    v2 tcoords;
    v3 normal;
    v3 emit;
    double distance;
    const object *obj;
    const material *m;
} intersection;

typedef struct { v3 origin, direction, direction_inv; } ray; /* Ray.hpp:6-19 */

typedef struct { uint64_t scene_rays, vertices, node_visits, tri_tests; } counters;

static inline intersection no_hit(void) { /* Intersection.hpp:13-20 */
    intersection r;
    memset(&r, 0, sizeof r);
    r.distance = DBL_MAX;
    return r;
}

static inline ray make_ray(v3 o, v3 d) { /* Ray.hpp:13-18: inverse through a double division */
    ray r;
    r.origin = o;
    r.direction = d;
    r.direction_inv = V3((float)(1. / (double)d.x), (float)(1. / (double)d.y), (float)(1. / (double)d.z));
    return r;
}

/* ------------------------------------------------------------------ Bounds3 */
static inline bounds3 bounds_empty(void) { /* Bounds3.hpp:17-22: double lowest/max overflow to -inf/+inf */
    bounds3 b;
    b.pMax = V3(-INFINITY, -INFINITY, -INFINITY);
    b.pMin = V3(INFINITY, INFINITY, INFINITY);
    return b;
}
static inline v3 vmin(v3 a, v3 b) { return V3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); } /* Bounds3.hpp:87-90 */
static inline v3 vmax(v3 a, v3 b) { return V3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); } /* Bounds3.hpp:91-94 */
static inline bounds3 bounds_pp(v3 p1, v3 p2) { bounds3 b = {vmin(p1, p2), vmax(p1, p2)}; return b; } /* Bounds3.hpp:24-29 */
static inline bounds3 bounds_union(bounds3 a, bounds3 b) { bounds3 r = {vmin(a.pMin, b.pMin), vmax(a.pMax, b.pMax)}; return r; } /* Bounds3.hpp:110-115 */
static inline bounds3 bounds_union_p(bounds3 a, v3 p) { bounds3 r = {vmin(a.pMin, p), vmax(a.pMax, p)}; return r; } /* Bounds3.hpp:117-122 */
static inline v3 bounds_centroid(bounds3 b) { return add(mulf(b.pMin, 0.5f), mulf(b.pMax, 0.5f)); } /* Bounds3.hpp:47 */
static inline int bounds_max_extent(bounds3 b) { /* Bounds3.hpp:32-40 */
    v3 d = sub(b.pMax, b.pMin);
    if (d.x > d.y && d.x > d.z) return 0;
    else if (d.y > d.z) return 1;
    else return 2;
}

/* Bounds3::IntersectP, Bounds3.hpp:95-108.  fmin/fmax per component ignore NaN; the
 * initializer-list std::max/std::min keep a NaN that sits in the first (x) slot. */
static inline int bounds_intersectP(const bounds3 *b, const ray *r) {
    v3 t1 = V3((b->pMin.x - r->origin.x) * r->direction_inv.x, (b->pMin.y - r->origin.y) * r->direction_inv.y,
               (b->pMin.z - r->origin.z) * r->direction_inv.z);
    v3 t2 = V3((b->pMax.x - r->origin.x) * r->direction_inv.x, (b->pMax.y - r->origin.y) * r->direction_inv.y,
               (b->pMax.z - r->origin.z) * r->direction_inv.z);
    v3 lo = vmin(t1, t2), hi = vmax(t1, t2);
    float tmin = lo.x;
    if (tmin < lo.y) tmin = lo.y;
    if (tmin < lo.z) tmin = lo.z;
    float tmax = hi.x;
    if (hi.y < tmax) tmax = hi.y;
    if (hi.z < tmax) tmax = hi.z;
    return (tmin - EPSILON <= tmax) && (tmax >= -EPSILON);
}

/* ------------------------------------------------------------------ Material */
static const float WAVELEN[3] = {0.700f, 0.5461f, 0.4358f}; /* WaveLen.hpp:7-18 */

/* Material.hpp:26-34 */
static float D_GGX(v3 h, v3 n, float alpha) {
    float NoH = fabsf(dot(n, h));
    if (NoH <= EPSILON && NoH >= -EPSILON) return 0.0f;
    float tanTheta = sqrtf(1.0f - NoH * NoH) / NoH;
    float alpha2 = alpha * alpha;
    float denom = (NoH * NoH) * (alpha + tanTheta * tanTheta);
    return alpha2 / (PI_F * denom * denom);
}

/* Material.hpp:38-69 */
static float G1_SmithGGX(v3 v, v3 n, float alpha) {
    float NoV = fabsf(dot(n, v));
    if (NoV <= EPSILON && NoV >= -EPSILON) return 0.0f;
    float tanTheta = sqrtf(1.0f - NoV * NoV) / NoV;
    if (tanTheta == 0.0f) return 1.0f;
    float al_tan = alpha * tanTheta;
    return (float)(2. / (1. + (double)sqrtf(1 + al_tan * al_tan)));
}

/* Material.hpp:70-77 */
static float G_SmithGGX(v3 wi, v3 wo, v3 n, float alpha) { return G1_SmithGGX(wi, n, alpha) * G1_SmithGGX(wo, n, alpha); }

/* Material.hpp:134-151 */
static float getReflectance(const material *m, v2 uv, int ch) {
    if (!m->textured) return comp(m->base_reflectance, ch);
    int col = (int)((uv.x - 0.05f) * 10);
    int row = (int)((uv.y - 0.00f) * 12);
    if (col >= 3 && col <= 5 && row <= 7) {
        int isWhite = (col + row) % 2 == 1;
        return isWhite ? 0.9f : 0.1f;
    } else {
        return 0.1f;
    }
}

/* Material.hpp:80-86 */
static float FresnelSchlick(const material *m, float cosTheta, v2 uv, int ch) {
    float f = getReflectance(m, uv, ch);
    float invc = 1.f - cosTheta;
    float c2 = invc * invc;
    return f + (1.f - f) * c2 * c2 * invc;
}

/* Material.hpp:95-106 */
static v3 tanToWorld(v3 t, v3 n) {
    v3 B, T;
    if (fabsf(n.x) > fabsf(n.y)) {
        float invLen = 1.0f / sqrtf(n.x * n.x + n.z * n.z);
        T = V3(-n.z * invLen, 0.0f, n.x * invLen);
    } else {
        float invLen = 1.0f / sqrtf(n.y * n.y + n.z * n.z);
        T = V3(0.0f, n.z * invLen, -n.y * invLen);
    }
    B = cross(n, T);
    return add(add(mulf(T, t.x), mulf(B, t.y)), mulf(n, t.z));
}

/* Material.hpp:111-123 */
static v3 ImportanceSampleGGX(float xi_x, float xi_y, float alpha, v3 n) {
    float phi = 2.0f * PI_F * xi_x;
    float cosTheta = sqrtf((1.0f - xi_y) / (1.0f + (alpha * alpha - 1.0f) * xi_y));
    float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float sp, cp;
    mcpt_sincosf(phi, &sp, &cp);
    v3 h = V3(sinTheta * cp, sinTheta * sp, cosTheta);
    return normalized(tanToWorld(h, n));
}

/* Material.hpp:178-183 */
static float getIor(const material *m, int ch) {
    float wl = WAVELEN[ch];
    return m->iorA + m->iorB / (wl * wl);
}

/* Material.hpp:195-197 */
static v3 mat_reflect(v3 I, v3 N) { return sub(mulf(N, 2 * dot(N, I)), I); }

/* Material.hpp:198-226 */
static float mat_fresnel(const material *m, v3 I, v3 N, int ch) {
    if (m->type == ORC_SMOOTH_CONDUCTOR || m->type == ORC_ROUGH_CONDUCTOR) return 1;
    float cosi = clampf(-1, 1, dot(I, N));
    float etai = 1, etat = getIor(m, ch);
    if (cosi > 0) { float t = etai; etai = etat; etat = t; }
    float sint = etai / etat * sqrtf(std_maxf(0.f, 1 - cosi * cosi));
    if (sint >= 1) {
        return 1;
    } else {
        float cost = sqrtf(std_maxf(0.f, 1 - sint * sint));
        cosi = fabsf(cosi);
        float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
        float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
        return (Rs * Rs + Rp * Rp) / 2;
    }
}

/* Material.hpp:227-242 */
static v3 mat_refract(const material *m, v3 I, v3 N, int ch) {
    float cosi = clampf(-1, 1, dot(I, N));
    float etai = 1, etat = getIor(m, ch);
    v3 n = N;
    if (cosi < 0) {
        cosi = -cosi;
    } else {
        float t = etai; etai = etat; etat = t;
        n = neg(N);
    }
    float eta = etai / etat;
    float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return V3(0, 0, 0);
    return add(mulf(I, eta), mulf(n, eta * cosi - sqrtf(k)));
}

/* Material.hpp:268-281 (+126-130).  xi = the two draws of Material.hpp:128. */
static v3 mat_sample(const material *m, v3 N, float xi_x, float xi_y) {
    switch (m->type) {
    case ORC_ROUGH_CONDUCTOR:
    case ORC_ROUGH_DIELECTRIC:
        return ImportanceSampleGGX(xi_x, xi_y, m->roughness, N);
    default:
        return N;
    }
}

/* eta as Material.hpp:299,318,360,393: float = (cond) ? ior : 1. / ior  (the division is double) */
static inline float eta_of(const material *m, v3 wi, v3 N, int ch) {
    float ior = getIor(m, ch);
    return (dot(wi, N) > 0) ? ior : (float)(1. / (double)ior);
}

/* Material.hpp:285-328 */
static float mat_pdf(const material *m, v3 wi, v3 wo, v3 N, int ch, int isReflect) {
    switch (m->type) {
    case ORC_ROUGH_CONDUCTOR:
    case ORC_ROUGH_DIELECTRIC: {
        v3 h;
        float jacobian;
        if (isReflect) {
            h = normalized(add(wi, wo));
            h = (dot(wi, N) > 0) ? h : neg(h);
            jacobian = 1.0f / (4.0f * fabsf(dot(h, wo)));
        } else {
            float eta = eta_of(m, wi, N, ch);
            v3 hv = sub(neg(wi), mulf(wo, eta));
            h = normalized(hv);
            float d1 = dot(hv, hv);
            jacobian = eta * eta * fabsf(dot(h, wo)) / d1;
        }
        float D = D_GGX(h, N, m->roughness);
        return D * dot(N, h) * jacobian;
    }
    default: {
        v3 h;
        if (isReflect) {
            h = normalized(add(wi, wo));
        } else {
            float eta = eta_of(m, wi, N, ch);
            h = normalized(sub(neg(wi), mulf(wo, eta)));
            h = dot(h, N) > 0 ? h : neg(h);
        }
        return (fabsf(dot(h, N)) > 1 - EPSILON) ? 1.0f : 0.0f;
    }
    }
}

/* Material.hpp:330-408 */
static float mat_eval(const material *m, v3 wi, v3 wo, v3 N, int ch, v2 uv, int isReflect) {
    switch (m->type) {
    case ORC_ROUGH_CONDUCTOR:
    case ORC_ROUGH_DIELECTRIC: {
        if (isReflect) {
            if (dot(wi, N) * dot(wo, N) <= 0) return 0.f;
            v3 h = normalized(add(wi, wo));
            h = dot(wi, N) > 0 ? h : neg(h);
            float F = (m->type == ORC_ROUGH_CONDUCTOR) ? FresnelSchlick(m, fabsf(dot(h, wo)), uv, ch)
                                                       : mat_fresnel(m, neg(wi), h, ch);
            float D = D_GGX(h, N, m->roughness);
            float G = G_SmithGGX(wi, wo, h, m->roughness);
            float denom = 4.0f * fabsf(dot(N, wi)) * fabsf(dot(N, wo)) + EPSILON;
            return F * D * G / denom;
        } else {
            if (m->type == ORC_ROUGH_CONDUCTOR || dot(wi, N) * dot(wo, N) >= 0) return 0.f;
            float eta = eta_of(m, wi, N, ch);
            v3 h = normalized(sub(neg(wi), mulf(wo, eta)));
            h = dot(h, N) > 0 ? h : neg(h);
            float F = mat_fresnel(m, neg(wi), h, ch);
            float D = D_GGX(h, N, m->roughness);
            float G = G_SmithGGX(wi, wo, h, m->roughness);
            float hol = dot(h, wi);
            float hov = dot(h, wo);
            float den = hol + eta * hov;
            den *= den;
            den *= fabsf(dot(N, wi) * dot(N, wo));
            return (1.0f - F) * D * G * eta * eta * fabsf(hol * hov) / den;
        }
    }
    default: { /* SMOOTH_CONDUCTOR, SMOOTH_DIELECTRIC */
        if (isReflect) {
            v3 h = normalized(add(wi, wo));
            h = (dot(wi, N) > 0) ? h : neg(h);
            if (dot(wi, N) * dot(wo, N) <= 0 || dot(h, N) < 1 - EPSILON) {
                return 0.f;
            } else {
                return (m->type == ORC_SMOOTH_CONDUCTOR) ? FresnelSchlick(m, fabsf(dot(N, wo)), uv, ch)
                                                         : mat_fresnel(m, neg(wi), N, ch);
            }
        } else {
            float eta = eta_of(m, wi, N, ch);
            v3 h = normalized(sub(neg(wi), mulf(wo, eta)));
            h = (dot(h, N) > 0) ? h : neg(h);
            if (m->type == ORC_SMOOTH_CONDUCTOR || dot(wi, N) * dot(wo, N) >= 0 || dot(h, N) < 1 - EPSILON) {
                return 0.f;
            } else {
                return (float)(1. - (double)mat_fresnel(m, neg(wi), N, ch));
            }
        }
    }
    }
}

/* ------------------------------------------------------------------ primitives */
static void triangle_init(object *t, v3 v0, v3 v1, v3 v2, const material *m) { /* Triangle.hpp:50-56 */
    memset(t, 0, sizeof *t);
    t->kind = K_TRIANGLE;
    t->v0 = v0; t->v1 = v1; t->v2 = v2; t->m = m;
    t->e1 = sub(v1, v0);
    t->e2 = sub(v2, v0);
    t->normal = normalized(cross(t->e1, t->e2));
    t->area = norm(cross(t->e1, t->e2)) * 0.5f;
}

static bounds3 object_bounds(const object *o) {
    switch (o->kind) {
    case K_TRIANGLE: return bounds_union_p(bounds_pp(o->v0, o->v1), o->v2); /* Triangle.hpp:220 */
    case K_SPHERE:                                                           /* Sphere.hpp:58-63 */
        return bounds_pp(V3(o->center.x - o->radius, o->center.y - o->radius, o->center.z - o->radius),
                         V3(o->center.x + o->radius, o->center.y + o->radius, o->center.z + o->radius));
    default: return o->bounding_box; /* Triangle.hpp:158 */
    }
}

static inline intersection closer(intersection l, intersection r) {
    /* BVH.cpp:115 `l.distance < r.distance ? l : r`, with the tie rule made tree-independent */
    if (l.distance < r.distance) return l;
    if (r.distance < l.distance) return r;
    if (l.happened && r.happened) return (l.obj->prim_id > r.obj->prim_id) ? l : r;
    return r;
}

static intersection bvh_get_intersection(const bvh_node *node, const ray *r, counters *c);

/* Triangle::getIntersection, Triangle.hpp:222-252 */
static intersection triangle_intersect(const object *t, const ray *r, counters *c) {
    intersection inter = no_hit();
    c->tri_tests++;
    double u, v, t_tmp = 0;
    v3 pvec = cross(r->direction, t->e2);
    double det = dot(t->e1, pvec);
    if (fabs(det) < EPSILON) return inter;
    double det_inv = 1. / det;
    v3 tvec = sub(r->origin, t->v0);
    u = dot(tvec, pvec) * det_inv;
    if (u < 0 || u > 1) return inter;
    v3 qvec = cross(tvec, t->e1);
    v = dot(r->direction, qvec) * det_inv;
    if (v < 0 || u + v > 1) return inter;
    t_tmp = dot(t->e2, qvec) * det_inv;
    if (t_tmp < 0) return inter;
    inter.happened = 1;
    inter.coords = add(r->origin, mulf(r->direction, (float)t_tmp)); /* Ray.hpp:21, Eigen converts the double scalar to float */
    inter.normal = t->normal;
    inter.distance = t_tmp;
    {
        float a = (float)(1 - u - v), b = (float)u, cc = (float)v;
        inter.tcoords.x = a * t->t0.x + b * t->t1.x + cc * t->t2.x;
        inter.tcoords.y = a * t->t0.y + b * t->t1.y + cc * t->t2.y;
    }
    inter.obj = t;
    inter.m = t->m;
    return inter;
}

/* global.hpp:20-35 */
static int solveQuadratic(float a, float b, float c, float *x0, float *x1) {
    float discr = b * b - 4 * a * c;
    if (discr < 0) return 0;
    else if (discr == 0) *x0 = *x1 = (float)(-0.5 * (double)b / (double)a);
    else {
        float q = (b > 0) ? (float)(-0.5 * (double)(b + sqrtf(discr))) : (float)(-0.5 * (double)(b - sqrtf(discr)));
        *x0 = q / a;
        *x1 = c / q;
    }
    if (*x0 > *x1) { float t = *x0; *x0 = *x1; *x1 = t; }
    return 1;
}

/* Sphere::getIntersection, Sphere.hpp:26-48 */
static intersection sphere_intersect(const object *s, const ray *r) {
    intersection result = no_hit();
    v3 L = sub(r->origin, s->center);
    float a = dot(r->direction, r->direction);
    float b = 2 * dot(r->direction, L);
    float c = dot(L, L) - s->radius2;
    float t0, t1;
    if (!solveQuadratic(a, b, c, &t0, &t1)) return result;
    if (t0 < 0) t0 = t1;
    if (t0 < 0) return result;
    result.happened = 1;
    result.coords = add(r->origin, mulf(r->direction, t0));
    result.normal = normalized(sub(result.coords, s->center));
    result.m = s->m;
    result.obj = s;
    result.distance = t0;
    return result;
}

static intersection object_intersect(const object *o, const ray *r, counters *c) {
    switch (o->kind) {
    case K_TRIANGLE: return triangle_intersect(o, r, c);
    case K_SPHERE: return sphere_intersect(o, r);
    default: /* MeshTriangle::getIntersection, Triangle.hpp:183-191 -> BVHAccel::Intersect, BVH.cpp:95-101 */
        if (o->bvh_root) return bvh_get_intersection(o->bvh_root, r, c);
        return no_hit();
    }
}

/* BVHAccel::getIntersection, BVH.cpp:103-116: both children are always visited, no t pruning */
static intersection bvh_get_intersection(const bvh_node *node, const ray *r, counters *c) {
    if (node == NULL) return no_hit();
    c->node_visits++;
    if (!bounds_intersectP(&node->bounds, r)) return no_hit();
    if (node->obj != NULL) return object_intersect(node->obj, r, c);
    intersection isectl = bvh_get_intersection(node->left, r, c);
    intersection isectr = bvh_get_intersection(node->right, r, c);
    return closer(isectl, isectr);
}

/* ------------------------------------------------------------------ BVH build */
static void stable_sort_objs(object **a, object **tmp, float *key, float *ktmp, int n) {
    /* bottom-up merge sort: deterministic stand-in for the std::sort of BVH.cpp:57-73 */
    for (int w = 1; w < n; w *= 2) {
        for (int i = 0; i < n; i += 2 * w) {
            int l = i, m = i + w < n ? i + w : n, h = i + 2 * w < n ? i + 2 * w : n;
            int p = l, q = m, k = l;
            while (p < m && q < h) {
                if (key[q] < key[p]) { tmp[k] = a[q]; ktmp[k++] = key[q++]; }
                else { tmp[k] = a[p]; ktmp[k++] = key[p++]; }
            }
            while (p < m) { tmp[k] = a[p]; ktmp[k++] = key[p++]; }
            while (q < h) { tmp[k] = a[q]; ktmp[k++] = key[q++]; }
        }
        memcpy(a, tmp, (size_t)n * sizeof *a);
        memcpy(key, ktmp, (size_t)n * sizeof *key);
    }
}

/* BVHAccel::recursiveBuild, BVH.cpp:27-93 */
static bvh_node *bvh_build(object **objs, int n) {
    bvh_node *node = (bvh_node *)calloc(1, sizeof *node);
    node->bounds = bounds_empty();
    if (n == 1) {
        node->bounds = object_bounds(objs[0]);
        node->obj = objs[0];
        node->area = objs[0]->area;
        return node;
    } else if (n == 2) {
        node->left = bvh_build(&objs[0], 1);
        node->right = bvh_build(&objs[1], 1);
        node->bounds = bounds_union(node->left->bounds, node->right->bounds);
        node->area = node->left->area + node->right->area;
        return node;
    }
    bounds3 cb = bounds_empty();
    for (int i = 0; i < n; ++i) cb = bounds_union_p(cb, bounds_centroid(object_bounds(objs[i])));
    int dim = bounds_max_extent(cb);
    object **tmp = (object **)malloc((size_t)n * sizeof *tmp);
    float *key = (float *)malloc((size_t)n * sizeof *key), *ktmp = (float *)malloc((size_t)n * sizeof *ktmp);
    for (int i = 0; i < n; ++i) key[i] = comp(bounds_centroid(object_bounds(objs[i])), dim);
    stable_sort_objs(objs, tmp, key, ktmp, n);
    free(tmp); free(key); free(ktmp);
    int mid = n / 2;
    node->left = bvh_build(objs, mid);
    node->right = bvh_build(objs + mid, n - mid);
    node->bounds = bounds_union(node->left->bounds, node->right->bounds);
    node->area = node->left->area + node->right->area;
    return node;
}

static void bvh_free(bvh_node *n) {
    if (!n) return;
    bvh_free(n->left);
    bvh_free(n->right);
    free(n);
}

/* ------------------------------------------------------------------ light sampling */
/* Triangle::Sample, Triangle.hpp:71-76 */
static void triangle_sample(const object *t, float ux, float uy, intersection *pos, float *pdf) {
    float x = sqrtf(ux), y = uy;
    pos->coords = add(add(mulf(t->v0, 1.0f - x), mulf(t->v1, x * (1.0f - y))), mulf(t->v2, x * y));
    pos->normal = t->normal;
    *pdf = 1.0f / t->area;
}

/* Sphere::Sample, Sphere.hpp:64-74 (unused by the shipped scenes: no emissive spheres) */
static void sphere_sample(const object *s, float u1, float u2, intersection *pos, float *pdf) {
    float theta = (float)(2.0 * (double)PI_F * (double)u1), phi = PI_F * u2;
    float sph, cph, sth, cth;
    mcpt_sincosf(phi, &sph, &cph);
    mcpt_sincosf(theta, &sth, &cth);
    v3 dir = V3(cph, sph * cth, sph * sth);
    pos->coords = add(s->center, mulf(dir, s->radius));
    pos->normal = dir;
    pos->obj = s;
    pos->m = s->m;
    *pdf = 1.0f / s->area;
}

/* BVHAccel::getSample, BVH.cpp:118-129 */
static void bvh_get_sample(const bvh_node *node, float p, float ux, float uy, intersection *pos, float *pdf) {
    if (node->left == NULL || node->right == NULL) {
        triangle_sample(node->obj, ux, uy, pos, pdf);
        *pdf *= node->area;
        return;
    }
    if (p < node->left->area) bvh_get_sample(node->left, p, ux, uy, pos, pdf);
    else bvh_get_sample(node->right, p - node->left->area, ux, uy, pos, pdf);
}

/* Scene::sampleLight, Scene.cpp:23-37.  u = {light choice, triangle pick, x, y} */
static void scene_sample_light(const orc_scene *s, const float u[4], intersection *pos, float *pdf) {
    float emit_area_sum = 0;
    for (int k = 0; k < s->n_lights; ++k) emit_area_sum += s->lights[k]->area;
    float p = u[0] * emit_area_sum;
    emit_area_sum = 0;
    for (int k = 0; k < s->n_lights; ++k) {
        const object *l = s->lights[k];
        emit_area_sum += l->area;
        if (p <= emit_area_sum) {
            if (l->kind == K_MESH) {
                /* MeshTriangle::Sample, Triangle.hpp:193-196 -> BVHAccel::Sample, BVH.cpp:131-135 */
                float pp = sqrtf(u[1]) * l->bvh_root->area;
                bvh_get_sample(l->bvh_root, pp, u[2], u[3], pos, pdf);
                *pdf /= l->bvh_root->area;
                pos->emit = l->m->emission;
            } else {
                sphere_sample(l, u[2], u[3], pos, pdf);
                /* Sphere::Sample does not set pos.emit (Sphere.hpp:64-74): it keeps its previous value */
            }
            break;
        }
    }
}

/* ------------------------------------------------------------------ Scene */
/* Scene::sampleEnv, Scene.hpp:60-99 */
static v3 scene_sample_env(const orc_scene *s, v3 dir) {
    if (!s->use_env) return s->background;
    v3 d = normalized(dir);
    float phi = mcpt_atan2f(d.z, d.x);
    float theta = mcpt_acosf(d.y);
    float u = (phi + PI_F) / (2.f * PI_F);
    float v = theta / PI_F;
    u = u - floorf(u);
    v = v < 0.f ? 0.f : (1.f < v ? 1.f : v); /* std::clamp(v, 0.f, 1.f) */
    float x = u * s->env_w - 0.5f;
    float y = v * s->env_h - 0.5f;
    int x0 = (int)floorf(x);
    int y0 = (int)floorf(y);
    int W = s->env_w, H = s->env_h;
    int X0 = x0 % W; if (X0 < 0) X0 += W;
    int X1 = (x0 + 1) % W; if (X1 < 0) X1 += W;
    int Y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    int Y1 = y0 + 1 < 0 ? 0 : (y0 + 1 > H - 1 ? H - 1 : y0 + 1);
    float sx = x - x0, sy = y - y0;
    v3 c00 = ld3(&s->env_pixels[3 * ((size_t)Y0 * W + X0)]), c10 = ld3(&s->env_pixels[3 * ((size_t)Y0 * W + X1)]);
    v3 c01 = ld3(&s->env_pixels[3 * ((size_t)Y1 * W + X0)]), c11 = ld3(&s->env_pixels[3 * ((size_t)Y1 * W + X1)]);
    v3 c0 = add(mulf(c00, 1 - sx), mulf(c10, sx));
    v3 c1 = add(mulf(c01, 1 - sx), mulf(c11, sx));
    return add(mulf(c0, 1 - sy), mulf(c1, sy));
}

/* Scene::intersect, Scene.cpp:19-21 */
static intersection scene_intersect(const orc_scene *s, const ray *r, counters *c) {
    c->scene_rays++;
    if (s->bvh_root == NULL) return no_hit();
    return bvh_get_intersection(s->bvh_root, r, c);
}

typedef struct {
    const orc_scene *s;
    float rrRate, invRr;
    int n_dir_sample, enable_shadow;
} render_ctx;

/* Scene::directLighting, Scene.cpp:56-82 */
static float scene_direct_lighting(const render_ctx *rc, v3 wo, const intersection *surf, int ch, int isReflect,
                                   const rng_key *rk, uint32_t depth, counters *c) {
    const material *m = surf->m;
    v3 p = surf->coords;
    v3 n = surf->normal;
    v2 uv = surf->tcoords;
    float l_dir = 0;
    float pdf = 0;
    intersection inter = no_hit();
    /* Without an emitter Scene::sampleLight (Scene.cpp:23-37) returns without touching `pdf`, which the reference then
     * reads uninitialised (undefined behaviour).  Convention here and in the GPU path: no emitter, no direct light. */
    if (rc->s->n_lights == 0) return 0;
    for (int i = 0; i < rc->n_dir_sample; i++) {
        float u[4];
        rng_block(rk, depth, 1u + (uint32_t)i, u);
        scene_sample_light(rc->s, u, &inter, &pdf);
        v3 p_light = inter.coords;
        v3 n_light = inter.normal;
        float emit = comp(inter.emit, ch);
        v3 ws = normalized(sub(p_light, p));
        float dist = norm(sub(p_light, p));
        ray rlight = make_ray(p, ws);
        inter = scene_intersect(rc->s, &rlight, c);
        if ((rc->enable_shadow == 0) || (inter.happened && fabs(inter.distance - (double)dist) < (double)EPSILON)) {
            l_dir += emit * mat_eval(m, ws, wo, n, ch, uv, isReflect) * (dot(ws, n)) * dot(neg(ws), n_light) /
                     (dist * dist) / pdf / rc->n_dir_sample;
        }
    }
    return l_dir;
}

/* Scene::castRay, Scene.cpp:85-184 */
static float scene_cast_ray(const render_ctx *rc, const ray *r_in, int depth, int ch, const rng_key *rk, counters *c) {
    const orc_scene *s = rc->s;
    c->vertices++;
    intersection inter = scene_intersect(s, r_in, c);
    if (!inter.happened) return comp(scene_sample_env(s, r_in->direction), ch); /* Scene.cpp:88-95 */
    v3 p = inter.coords;
    v3 n = inter.normal;
    const material *m = inter.m;
    v2 uv = inter.tcoords;
    v3 wo = neg(r_in->direction);

    if (depth == 0 && inter.obj->m->has_emission) /* Scene.cpp:102-107 */
        return clampf(0, 1, comp(inter.m->emission, ch) * fabsf(dot(wo, n)));

    float u0[4];
    rng_block(rk, (uint32_t)depth, 0u, u0);
    v3 mfn = mat_sample(m, n, u0[0], u0[1]);          /* Scene.cpp:109 */
    float kr = mat_fresnel(m, r_in->direction, mfn, ch); /* Scene.cpp:110 */
    float l_dir = 0, l_ind = 0;

    inter.coords = add(inter.coords, mulf(n, EPSILON)); /* Scene.cpp:114 */
    if (dot(wo, n) < 0) {
        l_dir = (float)((1. - (double)kr) * (double)scene_direct_lighting(rc, wo, &inter, ch, 0, rk, (uint32_t)depth, c));
    } else {
        l_dir = kr * scene_direct_lighting(rc, wo, &inter, ch, 1, rk, (uint32_t)depth, c);
    }

    float rr = u0[2];       /* Scene.cpp:121 */
    float rd_flect = u0[3]; /* Scene.cpp:122 */
    int isReflect = rd_flect < kr;
    if (isReflect) {
        if (dot(wo, mfn) < 0) p = sub(p, mulf(n, EPSILON));
        else p = add(p, mulf(n, EPSILON));
    } else {
        if (dot(wo, mfn) < 0) p = add(p, mulf(n, EPSILON));
        else p = sub(p, mulf(n, EPSILON));
    }
    if (rr >= rc->rrRate) return l_dir; /* Scene.cpp:129-131,156-158: unclamped */
    v3 wi = isReflect ? mat_reflect(wo, mfn) : mat_refract(m, r_in->direction, mfn, ch);
    ray r = make_ray(p, wi);
    inter = scene_intersect(s, &r, c);
    if (inter.happened && !inter.obj->m->has_emission) {
        if (m->isDirac) {
            l_ind = scene_cast_ray(rc, &r, depth + 1, ch, rk, c) * mat_eval(m, wi, wo, n, ch, uv, isReflect) * rc->invRr;
        } else {
            l_ind = scene_cast_ray(rc, &r, depth + 1, ch, rk, c) * mat_eval(m, wi, wo, n, ch, uv, isReflect) *
                    fabsf(dot(wo, n)) / mat_pdf(m, wi, wo, n, ch, isReflect) * rc->invRr;
        }
    } else {
        float env = comp(scene_sample_env(s, r.direction), ch); /* Scene.cpp:145-149,172-176 */
        l_ind = env * mat_eval(m, wi, wo, n, ch, uv, isReflect) * rc->invRr;
    }
    float threshold_ind = 5, threshold_dir = 15; /* Scene.cpp:180-183 */
    l_ind = clampf(0, threshold_ind, l_ind);
    l_dir = clampf(0, threshold_dir, l_dir);
    return l_dir + l_ind;
}

/* ------------------------------------------------------------------ scene construction */
int orc_scene_create(const orc_scene_desc *d, orc_scene **out) {
    if (!d || !out || d->n_objects <= 0) return 1;
    orc_scene *s = (orc_scene *)calloc(1, sizeof *s);
    s->n_objects = d->n_objects;
    s->n_materials = d->n_materials;
    s->n_triangles = d->n_triangles;
    s->materials = (material *)calloc((size_t)d->n_materials, sizeof(material));
    for (int i = 0; i < d->n_materials; ++i) {
        const orc_material *sm = &d->materials[i];
        material *m = &s->materials[i];
        m->type = sm->type;
        m->textured = sm->textured;
        m->isDirac = (sm->type == ORC_SMOOTH_CONDUCTOR || sm->type == ORC_SMOOTH_DIELECTRIC); /* Material.hpp:248-249 */
        m->roughness = sm->roughness; m->iorA = sm->iorA; m->iorB = sm->iorB;
        m->base_reflectance = ld3(sm->base_reflectance);
        m->emission = ld3(sm->emission);
        m->has_emission = norm(m->emission) > EPSILON; /* Material.hpp:262 */
    }
    s->objects = (object *)calloc((size_t)d->n_objects, sizeof(object));
    s->lights = (object **)calloc((size_t)d->n_objects, sizeof(object *));
    for (int i = 0; i < d->n_objects; ++i) {
        const orc_object *so = &d->objects[i];
        object *o = &s->objects[i];
        if (so->material < 0 || so->material >= d->n_materials) return 2;
        o->m = &s->materials[so->material];
        if (so->kind == ORC_OBJ_SPHERE) { /* Sphere.hpp:20-21 */
            o->kind = K_SPHERE;
            o->center = ld3(so->center);
            o->radius = so->radius;
            o->radius2 = so->radius * so->radius;
            o->area = 4 * PI_F * so->radius * so->radius;
            o->prim_id = d->n_triangles + i;
        } else { /* MeshTriangle::MeshTriangle, Triangle.hpp:83-135 (after the OBJ vertex stream was grouped by 3) */
            if (so->first_tri < 0 || so->n_tri <= 0 || so->first_tri + so->n_tri > d->n_triangles) return 3;
            o->kind = K_MESH;
            o->n_tris = so->n_tri;
            o->tris = (object *)calloc((size_t)so->n_tri, sizeof(object));
            v3 mn = V3(INFINITY, INFINITY, INFINITY), mx = V3(-INFINITY, -INFINITY, -INFINITY);
            o->area = 0;
            object **ptrs = (object **)malloc((size_t)so->n_tri * sizeof *ptrs);
            for (int k = 0; k < so->n_tri; ++k) {
                const orc_triangle *st = &d->triangles[so->first_tri + k];
                object *t = &o->tris[k];
                triangle_init(t, ld3(st->v0), ld3(st->v1), ld3(st->v2), o->m);
                t->prim_id = so->first_tri + k;
                t->t0.x = st->t0[0]; t->t0.y = st->t0[1];
                t->t1.x = st->t1[0]; t->t1.y = st->t1[1];
                t->t2.x = st->t2[0]; t->t2.y = st->t2[1];
                v3 vs[3] = {t->v0, t->v1, t->v2};
                for (int j = 0; j < 3; ++j) {
                    mn = V3(std_minf(mn.x, vs[j].x), std_minf(mn.y, vs[j].y), std_minf(mn.z, vs[j].z));
                    mx = V3(std_maxf(mx.x, vs[j].x), std_maxf(mx.y, vs[j].y), std_maxf(mx.z, vs[j].z));
                }
                ptrs[k] = t;
                o->area += t->area; /* Triangle.hpp:129-132 */
            }
            o->bounding_box = bounds_pp(mn, mx);
            o->bvh_root = bvh_build(ptrs, so->n_tri); /* Triangle.hpp:134 */
            free(ptrs);
        }
        if (o->m->has_emission) s->lights[s->n_lights++] = o; /* Scene.hpp:104-109 */
    }
    {
        object **ptrs = (object **)malloc((size_t)d->n_objects * sizeof *ptrs);
        for (int i = 0; i < d->n_objects; ++i) ptrs[i] = &s->objects[i];
        s->bvh_root = bvh_build(ptrs, d->n_objects); /* Scene.cpp:14-17 */
        free(ptrs);
    }
    s->background = ld3(d->background);
    if (d->env_w > 0 && d->env_h > 0 && d->env_pixels) {
        s->use_env = 1;
        s->env_w = d->env_w; s->env_h = d->env_h;
        size_t nb = (size_t)d->env_w * d->env_h * 3 * sizeof(float);
        s->env_pixels = (float *)malloc(nb);
        memcpy(s->env_pixels, d->env_pixels, nb);
    }
    *out = s;
    return 0;
}

void orc_scene_destroy(orc_scene *s) {
    if (!s) return;
    for (int i = 0; i < s->n_objects; ++i) {
        bvh_free(s->objects[i].bvh_root);
        free(s->objects[i].tris);
    }
    bvh_free(s->bvh_root);
    free(s->objects); free(s->lights); free(s->materials); free(s->env_pixels);
    free(s);
}

/* ------------------------------------------------------------------ camera + render */
static inline v3 mat3_mul(const float *M, v3 v) { /* Eigen 3x3 * 3x1: row . v with the 3-term redux */
    return V3(M[0] * v.x + (M[1] * v.y + M[2] * v.z), M[3] * v.x + (M[4] * v.y + M[5] * v.z),
              M[6] * v.x + (M[7] * v.y + M[8] * v.z));
}

static inline float deg2rad(float deg) { return (float)((double)(deg * PI_F) / 180.0); } /* Renderer.cpp:13 */

/* Renderer.cpp:25-29,44-76 */
static void camera_ray(const orc_camera *cam, float scale, float aspect, uint32_t seed, uint32_t m, uint32_t k, v3 *pos,
                       v3 *dir) {
    int i = (int)(m % (uint32_t)cam->width), j = (int)(m / (uint32_t)cam->width);
    rng_key rk = {seed, m, k, 3u};
    float u[4];
    rng_block(&rk, 0u, 0u, u);
    v3 eye = ld3(cam->position);
    float x = (1 - 2 * (i + u[0]) / (float)cam->width) * aspect * scale;
    float y = (1 - 2 * (j + u[1]) / (float)cam->height) * scale;
    if (cam->use_dof) {
        v3 focal_point = mulf(V3(x, y, 1), cam->focal_distance);
        float r = cam->aperture_radius * sqrtf(u[2]);
        float theta = 2 * PI_F * u[3];
        float st, ct;
        mcpt_sincosf(theta, &st, &ct);
        float dx = r * ct;
        float dy = r * st;
        *pos = add(eye, mat3_mul(cam->orientation, V3(dx, dy, 0)));
        *dir = normalized(sub(focal_point, V3(dx, dy, 0)));
    } else {
        *dir = normalized(V3(x, y, 1));
        *pos = eye;
    }
    *dir = mat3_mul(cam->orientation, *dir); /* Renderer.cpp:76 */
}

static inline int tile_owner(const orc_params *p, int W, int i, int j) {
    if (p->nranks <= 1 || p->tile_size <= 0) return 1;
    int tx = (W + p->tile_size - 1) / p->tile_size;
    int t = (j / p->tile_size) * tx + (i / p->tile_size);
    return (t % p->nranks) == p->rank;
}

static render_ctx make_ctx(const orc_scene *s, const orc_params *p) {
    render_ctx rc;
    rc.s = s;
    rc.rrRate = p->rr_rate;
    rc.invRr = 1 / p->rr_rate; /* Scene.hpp:110-113 */
    rc.n_dir_sample = p->n_dir_sample;
    rc.enable_shadow = p->enable_shadow;
    return rc;
}

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

void orc_camera_ray(const orc_camera *cam, uint32_t seed, uint32_t m, uint32_t k, float *origin, float *dir) {
    float scale = (float)tan((double)deg2rad(cam->fov * 0.5f));
    float aspect = cam->width / (float)cam->height;
    v3 o, d;
    camera_ray(cam, scale, aspect, seed, m, k, &o, &d);
    origin[0] = o.x; origin[1] = o.y; origin[2] = o.z;
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* Renderer::Render, Renderer.cpp:21-91 (tone map + PNG write stay with the caller) */
int orc_render(const orc_scene *s, const orc_camera *cam, const orc_params *p, float *fb, orc_stats *stats) {
    if (!s || !cam || !p || !fb || p->spp <= 0) return 1;
    render_ctx rc = make_ctx(s, p);
    const int W = cam->width, H = cam->height, spp = p->spp;
    float scale = (float)tan((double)deg2rad(cam->fov * 0.5f)); /* Renderer.cpp:25 */
    float aspect = cam->width / (float)cam->height;             /* Renderer.cpp:26 */
    uint64_t tot_rays = 0, tot_vert = 0, tot_nodes = 0, tot_tris = 0, tot_samples = 0;
    double t0 = now_s();
    int nthreads = p->n_threads;
#ifdef _OPENMP
    if (nthreads <= 0) nthreads = omp_get_max_threads();
#else
    nthreads = 1;
#endif
#pragma omp parallel for num_threads(nthreads) schedule(dynamic, 8) reduction(+ : tot_rays, tot_vert, tot_nodes, tot_tris, tot_samples)
    for (int m = 0; m < H * W; ++m) { /* Renderer.cpp:36-37 */
        int i = m % W, j = m / W;
        if (!tile_owner(p, W, i, j)) continue;
        counters c = {0, 0, 0, 0};
        v3 acc = V3(0, 0, 0);
        for (int k = 0; k < spp; k++) { /* Renderer.cpp:40 */
            v3 pos, dir;
            camera_ray(cam, scale, aspect, p->seed, (uint32_t)m, (uint32_t)k, &pos, &dir);
            ray r = make_ray(pos, dir);
            float col[3];
            for (int ch = 0; ch < 3; ++ch) { /* Renderer.cpp:77-79 */
                rng_key rk = {p->seed, (uint32_t)m, (uint32_t)k, (uint32_t)ch};
                col[ch] = scene_cast_ray(&rc, &r, 0, ch, &rk, &c);
            }
            acc = add(acc, divf(V3(col[0], col[1], col[2]), (float)spp)); /* Renderer.cpp:80 */
        }
        fb[3 * (size_t)m + 0] = acc.x;
        fb[3 * (size_t)m + 1] = acc.y;
        fb[3 * (size_t)m + 2] = acc.z;
        tot_rays += c.scene_rays; tot_vert += c.vertices; tot_nodes += c.node_visits; tot_tris += c.tri_tests;
        tot_samples += (uint64_t)spp;
    }
    if (stats) {
        stats->samples = tot_samples; stats->scene_rays = tot_rays; stats->vertices = tot_vert;
        stats->node_visits = tot_nodes; stats->tri_tests = tot_tris;
        stats->seconds = now_s() - t0;
    }
    return 0;
}

int orc_intersect(const orc_scene *s, int64_t n, const float *origins, const float *dirs, double *out_t,
                  int32_t *out_prim) {
    if (!s) return 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        counters c = {0, 0, 0, 0};
        ray r = make_ray(ld3(&origins[3 * i]), ld3(&dirs[3 * i]));
        intersection it = scene_intersect(s, &r, &c);
        out_t[i] = it.distance;
        out_prim[i] = it.happened ? it.obj->prim_id : -1;
    }
    return 0;
}

/* What the camera rays of a frame see: for pixel m and sample k < spp (Renderer.cpp:36-76, depth of field included) the primitive
 * Scene::intersect returns for the primary ray, out_prim[m * spp + k] (-1: background).  Used by the tests that pin the scene
 * assembly, camera and depth of field against the images the reference ships. */
int orc_primary_hits(const orc_scene *s, const orc_camera *cam, uint32_t seed, int32_t spp, int32_t *out_prim) {
    if (!s || !cam || !out_prim || spp <= 0) return 1;
    const int W = cam->width, H = cam->height;
    float scale = (float)tan((double)deg2rad(cam->fov * 0.5f));
    float aspect = cam->width / (float)cam->height;
#pragma omp parallel for schedule(dynamic, 64)
    for (int m = 0; m < H * W; ++m) {
        counters c = {0, 0, 0, 0};
        for (int k = 0; k < spp; ++k) {
            v3 pos, dir;
            camera_ray(cam, scale, aspect, seed, (uint32_t)m, (uint32_t)k, &pos, &dir);
            ray r = make_ray(pos, dir);
            intersection it = scene_intersect(s, &r, &c);
            out_prim[(size_t)m * (size_t)spp + (size_t)k] = it.happened ? it.obj->prim_id : -1;
        }
    }
    return 0;
}

int orc_cast_rays(const orc_scene *s, const orc_params *p, int64_t n, const float *origins, const float *dirs,
                  const uint32_t *pixel, const uint32_t *sample, const int32_t *channel, float *out) {
    if (!s || !p) return 1;
    render_ctx rc = make_ctx(s, p);
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < n; ++i) {
        counters c = {0, 0, 0, 0};
        ray r = make_ray(ld3(&origins[3 * i]), ld3(&dirs[3 * i]));
        rng_key rk = {p->seed, pixel[i], sample[i], (uint32_t)channel[i]};
        out[i] = scene_cast_ray(&rc, &r, 0, channel[i], &rk, &c);
    }
    return 0;
}

/* ------------------------------------------------------------------ KAT entry points */
static material mat_from_desc(const orc_material *sm) {
    material m;
    memset(&m, 0, sizeof m);
    m.type = sm->type; m.textured = sm->textured;
    m.isDirac = (sm->type == ORC_SMOOTH_CONDUCTOR || sm->type == ORC_SMOOTH_DIELECTRIC);
    m.roughness = sm->roughness; m.iorA = sm->iorA; m.iorB = sm->iorB;
    m.base_reflectance = ld3(sm->base_reflectance); m.emission = ld3(sm->emission);
    m.has_emission = norm(m.emission) > EPSILON;
    return m;
}
float orc_material_eval(const orc_material *sm, const float *wi, const float *wo, const float *n, int ch, const float *uv,
                        int is_reflect) {
    material m = mat_from_desc(sm);
    v2 t = {uv ? uv[0] : 0.f, uv ? uv[1] : 0.f};
    return mat_eval(&m, ld3(wi), ld3(wo), ld3(n), ch, t, is_reflect);
}
float orc_material_pdf(const orc_material *sm, const float *wi, const float *wo, const float *n, int ch, int is_reflect) {
    material m = mat_from_desc(sm);
    return mat_pdf(&m, ld3(wi), ld3(wo), ld3(n), ch, is_reflect);
}
float orc_material_fresnel(const orc_material *sm, const float *I, const float *N, int ch) {
    material m = mat_from_desc(sm);
    return mat_fresnel(&m, ld3(I), ld3(N), ch);
}
void orc_material_sample(const orc_material *sm, const float *n, float u1, float u2, float *out) {
    material m = mat_from_desc(sm);
    v3 r = mat_sample(&m, ld3(n), u1, u2);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_material_refract(const orc_material *sm, const float *I, const float *N, int ch, float *out) {
    material m = mat_from_desc(sm);
    v3 r = mat_refract(&m, ld3(I), ld3(N), ch);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* Scene::sampleLight (Scene.cpp:23-37) and Scene::sampleEnv (Scene.hpp:60-99) on arrays, for the device-vs-oracle function tests:
 * kind 0: u = 4 floats per row {light choice, triangle pick, x, y} -> out 10 floats {coords, normal, emit, pdf};
 * kind 1: in = 3 floats per row (a direction) -> out 3 floats (radiance). */
void orc_scene_function(const orc_scene *s, int kind, int64_t n, const float *in, float *out) {
    for (int64_t i = 0; i < n; ++i) {
        if (kind == 0) {
            intersection pos = no_hit();
            float pdf = 0.f;
            scene_sample_light(s, in + 4 * i, &pos, &pdf);
            float *o = out + 10 * i;
            o[0] = pos.coords.x; o[1] = pos.coords.y; o[2] = pos.coords.z;
            o[3] = pos.normal.x; o[4] = pos.normal.y; o[5] = pos.normal.z;
            o[6] = pos.emit.x; o[7] = pos.emit.y; o[8] = pos.emit.z;
            o[9] = pdf;
        } else {
            v3 c = scene_sample_env(s, ld3(in + 3 * i));
            out[3 * i] = c.x; out[3 * i + 1] = c.y; out[3 * i + 2] = c.z;
        }
    }
}

/* mcpt_fmath.h entry points for tests/test_fmath.py: kind 0 sin, 1 cos, 2 atan2(x, y), 3 acos, 4 pow(x, y), 5 tone-map byte */
void orc_fmath(int kind, int64_t n, const float *x, const float *y, float *out) {
    for (int64_t i = 0; i < n; ++i) {
        float s, c;
        switch (kind) {
        case 0: mcpt_sincosf(x[i], &s, &c); out[i] = s; break;
        case 1: mcpt_sincosf(x[i], &s, &c); out[i] = c; break;
        case 2: out[i] = mcpt_atan2f(x[i], y[i]); break;
        case 3: out[i] = mcpt_acosf(x[i]); break;
        case 4: out[i] = mcpt_powf(x[i], y[i]); break;
        default: out[i] = (float)mcpt_tonemap_byte(x[i]); break;
        }
    }
}

/* Renderer.cpp:95-103: raw = (unsigned char) clamp(0, 255, 255 * pow(c, 0.45)); alpha 255 */
void orc_tonemap(const float *fb, int64_t npixels, uint8_t *rgba) {
    const float inv_gamma = 0.45f;
    for (int64_t i = 0; i < npixels; ++i) {
        for (int c = 0; c < 3; ++c) rgba[4 * i + c] = (uint8_t)clampf(0, 255, 255 * powf(fb[3 * i + c], inv_gamma));
        rgba[4 * i + 3] = 255;
    }
}
