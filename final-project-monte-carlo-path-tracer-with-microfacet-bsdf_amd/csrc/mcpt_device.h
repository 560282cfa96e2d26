// Device-side math for the wavefront path tracer (gfx950).  Included only by mcpt_kernels.hip.
//
// Arithmetic contract: every expression keeps the reference's operand order and
// precision class -- float vectors with Eigen's 3-term dot order x0*y0 + (x1*y1 + x2*y2), the double
// det/u/v/t chain of Triangle::getIntersection, double sub-expressions wherever the reference source has a
// double literal -- and the file is compiled with -ffp-contract=off, so that the same seeds give the
// same paths as a CPU restatement that follows the same contract.  sin/cos/atan2/acos come from mcpt_fmath.h (plain IEEE
// arithmetic, identical on host and device), not from the device libm.
#pragma once
#include <hip/hip_runtime.h>

#include "mcpt_fmath.h"
#include "mcpt_internal.h"

namespace mcpt {

struct f3 {
    float x, y, z;
};
struct f2 {
    float x, y;
};

#define MCPT_DI __device__ __forceinline__

MCPT_DI f3 mk3(float x, float y, float z) { return {x, y, z}; }
MCPT_DI f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
MCPT_DI f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
MCPT_DI f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
MCPT_DI f3 operator*(f3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
MCPT_DI f3 operator/(f3 a, float s) { return {a.x / s, a.y / s, a.z / s}; }
MCPT_DI float dot(f3 a, f3 b) { return a.x * b.x + (a.y * b.y + a.z * b.z); }
MCPT_DI f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
MCPT_DI float norm(f3 a) { return sqrtf(dot(a, a)); }
MCPT_DI f3 normalized(f3 a) {  // Eigen normalized(): unchanged when the squared norm is 0
    float z = dot(a, a);
    if (z > 0.0f) return a / sqrtf(z);
    return a;
}
MCPT_DI float comp(f3 a, int c) { return c == 0 ? a.x : (c == 1 ? a.y : a.z); }
// (selects, not a[c]: a dynamically indexed member forces the whole material record into scratch/LDS)
MCPT_DI float comp(const float *a, int c) { return c == 0 ? a[0] : (c == 1 ? a[1] : a[2]); }

// std::min / std::max / clamp of reference global.hpp:16-18 (a NaN v comes out as hi)
MCPT_DI float std_min(float a, float b) { return (b < a) ? b : a; }
MCPT_DI float std_max(float a, float b) { return (a < b) ? b : a; }
MCPT_DI float clampf(float lo, float hi, float v) { return std_max(lo, std_min(hi, v)); }

constexpr float kPi = 3.141592653589793f;  // global.hpp:8-9 (a float)

// ---------------------------------------------------------------- Philox4x32-10
struct RngKey {
    uint32_t seed, pixel, sample, stream;
};

MCPT_DI void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul_hi / mul_lo pair
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0;
        c1 = lo1;
        c2 = n2;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

// One block = 4 uniforms in [0,1).  key = (seed, pixel); counter = (sample, depth, block, stream).
MCPT_DI void rng_block(const RngKey &k, uint32_t depth, uint32_t block, float u[4]) {
    uint32_t o[4];
    philox4x32_10(k.sample, depth, block, k.stream, k.seed, k.pixel, o);
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(o[i] >> 8) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------- scene view passed to kernels by value
struct DevScene {
    const Node *nodes;
    const QNode *qnodes;  // quantised nodes, or nullptr (then `nodes` is traversed)
    const float4 *pnodes; // SMALL kernels only: their LDS copy of the quantised nodes as floats relative to q_origin (4 float4 per node), or nullptr
    float q_origin[3], q_cell[3];
    const TriGeom *tri_geom;
    const TriShade *tri_shade;
    const SphereRec *spheres;
    const MaterialRec *mats;
    const LightRec *lights;
    const LightNode *light_nodes;
    const LightTri *light_tris;
    const InstRec *inst;   // instanced objects, or nullptr (plain tree); leaf index n_leaf_prims + k refers to inst[k]
    int32_t n_leaf_prims;  // leaf indices below it are primitives
    const float *env;
    float root_min[3], root_max[3];
    float background[3];
    float light_center[3], light_radius, light_area_sum;
    int32_t root, n_tri, n_lights, env_w, env_h, height;
    // Small-scene flavour (kernels' SMALL template flag, mcpt_kernels.hip): the whole traversal data set -- nodes, TriGeom, spheres -- and
    // the light tables fit a few KB, and every workgroup copies them into LDS once.  The counts say how much there is to copy.
    int32_t small, n_inner, n_sphere_slots, n_mats, n_light_nodes, n_light_tris;
    unsigned long long *dbg;  // traversal statistics (only written by -DMCPT_TRAVERSAL_STATS builds)
};

// ---------------------------------------------------------------- ray / box / primitive tests
struct Ray {
    f3 o, d, inv;
};

// Ray.hpp:13-18 computes the inverse as (float)(1. / (double)d).  For a single division of float operands,
// rounding to double first and to float second is innocuous (double has 53 >= 2*24 + 2 significand bits), so
// the result is bit-identical to the correctly rounded f32 division below (hipcc's default for `/`);
// tests/test_gpu_parity.py::test_intersect_bit_exact keeps this honest.
MCPT_DI Ray make_ray(f3 o, f3 d) {
    Ray r;
    r.o = o;
    r.d = d;
    r.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    return r;
}

// Bounds3::IntersectP, Bounds3.hpp:95-108.  fmin/fmax ignore a NaN operand; the initializer-list
// std::max/std::min keep a NaN that sits in the x slot.
// FAST: the caller guarantees that the ray's three reciprocals are finite.  Box corners and the origin are finite,
// so every slab product is then finite or +-inf (overflow) and no NaN exists; on NaN-free inputs the chained
// compare/selects of the reference equal plain max/min, which compile to v_max3_f32 / v_min3_f32.
template <bool FAST>
MCPT_DI bool box_from_slabs(float t1x, float t1y, float t1z, float t2x, float t2y, float t2z, float &tmin_out, float &tmax_out) {
    const float lx = fminf(t1x, t2x), ly = fminf(t1y, t2y), lz = fminf(t1z, t2z);
    const float hx = fmaxf(t1x, t2x), hy = fmaxf(t1y, t2y), hz = fmaxf(t1z, t2z);
    float tmin, tmax;
    if (FAST) {
        tmin = fmaxf(fmaxf(lx, ly), lz);
        tmax = fminf(fminf(hx, hy), hz);
    } else {
        tmin = lx;
        if (tmin < ly) tmin = ly;
        if (tmin < lz) tmin = lz;
        tmax = hx;
        if (hy < tmax) tmax = hy;
        if (hz < tmax) tmax = hz;
    }
    tmin_out = tmin;
    tmax_out = tmax;
    return (tmin - kEps <= tmax) && (tmax >= -kEps);
}

template <bool FAST>
MCPT_DI bool box_hit(const float mn[3], const float mx[3], const Ray &r, float &tmin_out, float &tmax_out) {
    const float t1x = (mn[0] - r.o.x) * r.inv.x, t1y = (mn[1] - r.o.y) * r.inv.y, t1z = (mn[2] - r.o.z) * r.inv.z;
    const float t2x = (mx[0] - r.o.x) * r.inv.x, t2y = (mx[1] - r.o.y) * r.inv.y, t2z = (mx[2] - r.o.z) * r.inv.z;
    return box_from_slabs<FAST>(t1x, t1y, t1z, t2x, t2y, t2z, tmin_out, tmax_out);
}

// Quantised child box (QNode): coordinate = q_origin + q * q_cell.  For rays with finite reciprocals the slab product
// (coordinate - o) * inv is evaluated as q * (q_cell * inv) + (q_origin - o) * inv with one FMA per bound; its rounding error
// is about 2 % of the one-cell margin the builder added, so the box still contains the exact one.
struct QRay {
    f3 a, b;  // a = q_cell * inv, b = (q_origin - o) * inv
};
MCPT_DI QRay make_qray(const DevScene &S, const Ray &r) {
    QRay q;
    q.a = mk3(S.q_cell[0] * r.inv.x, S.q_cell[1] * r.inv.y, S.q_cell[2] * r.inv.z);
    q.b = mk3((S.q_origin[0] - r.o.x) * r.inv.x, (S.q_origin[1] - r.o.y) * r.inv.y, (S.q_origin[2] - r.o.z) * r.inv.z);
    return q;
}
template <bool FAST>
MCPT_DI bool qbox_hit(const DevScene &S, const Ray &r, const QRay &qr, uint32_t mnx, uint32_t mny, uint32_t mnz, uint32_t mxx, uint32_t mxy,
                      uint32_t mxz, float &tmin_out, float &tmax_out) {
    if (FAST) {
        const float t1x = __builtin_fmaf((float)mnx, qr.a.x, qr.b.x), t1y = __builtin_fmaf((float)mny, qr.a.y, qr.b.y);
        const float t1z = __builtin_fmaf((float)mnz, qr.a.z, qr.b.z);
        const float t2x = __builtin_fmaf((float)mxx, qr.a.x, qr.b.x), t2y = __builtin_fmaf((float)mxy, qr.a.y, qr.b.y);
        const float t2z = __builtin_fmaf((float)mxz, qr.a.z, qr.b.z);
        return box_from_slabs<true>(t1x, t1y, t1z, t2x, t2y, t2z, tmin_out, tmax_out);
    }
    const float mn[3] = {S.q_origin[0] + (float)mnx * S.q_cell[0], S.q_origin[1] + (float)mny * S.q_cell[1], S.q_origin[2] + (float)mnz * S.q_cell[2]};
    const float mx[3] = {S.q_origin[0] + (float)mxx * S.q_cell[0], S.q_origin[1] + (float)mxy * S.q_cell[1], S.q_origin[2] + (float)mxz * S.q_cell[2]};
    return box_hit<false>(mn, mx, r, tmin_out, tmax_out);
}

// Prepared child box (the SMALL kernels' LDS nodes): x' = q * q_cell as a float, so coordinate = q_origin + x' and the slab bound of a ray
// with finite reciprocals is x' * inv + (q_origin - o) * inv: one FMA, no conversion.  x' carries one rounding (6e-8 of the scene's extent,
// 0.4 % of a grid cell), the FMA one more: far inside the one-cell margin of the quantised boxes, so the box still contains the exact one.
template <bool FAST>
MCPT_DI bool pbox_hit(const DevScene &S, const Ray &r, const QRay &qr, float mnx, float mny, float mnz, float mxx, float mxy, float mxz,
                      float &tmin_out, float &tmax_out) {
    if (FAST) {
        const float t1x = __builtin_fmaf(mnx, r.inv.x, qr.b.x), t1y = __builtin_fmaf(mny, r.inv.y, qr.b.y), t1z = __builtin_fmaf(mnz, r.inv.z, qr.b.z);
        const float t2x = __builtin_fmaf(mxx, r.inv.x, qr.b.x), t2y = __builtin_fmaf(mxy, r.inv.y, qr.b.y), t2z = __builtin_fmaf(mxz, r.inv.z, qr.b.z);
        return box_from_slabs<true>(t1x, t1y, t1z, t2x, t2y, t2z, tmin_out, tmax_out);
    }
    // (the same float values the quantised path dequantises to: q_origin + (float)q * q_cell)
    const float mn[3] = {S.q_origin[0] + mnx, S.q_origin[1] + mny, S.q_origin[2] + mnz};
    const float mx[3] = {S.q_origin[0] + mxx, S.q_origin[1] + mxy, S.q_origin[2] + mxz};
    return box_hit<false>(mn, mx, r, tmin_out, tmax_out);
}

MCPT_DI bool ray_is_plain(const Ray &r) {  // all three reciprocals finite (no zero / denormal direction component)
    return (fabsf(r.inv.x) < INFINITY) && (fabsf(r.inv.y) < INFINITY) && (fabsf(r.inv.z) < INFINITY);
}

// Triangle::getIntersection, Triangle.hpp:222-252 (two-sided Moller-Trumbore, double det chain).
//
// The reference divides first (`det_inv = 1. / det`, a double division) and rejects afterwards; about two thirds of the
// candidates are rejected on u or v.  Every rejection that can be decided WITHOUT the quotient is decided first, with the
// reference's outcome exactly:
//   * det, u's numerator a = tvec.pvec, v's numerator b = dir.qvec and t's numerator c = e2.qvec are float dot products;
//     the reference's doubles are their exact widenings.  det_inv = RN(1/det) has the sign of det and, like the products
//     a * det_inv etc., cannot underflow (|det| >= 1e-4, |numerator| >= 2^-149), so  u < 0  <=>  a and det have opposite
//     signs (a == 0 gives u = +-0, not < 0); the same holds for v < 0 and t < 0.  |det| < EPSILON is the same comparison
//     in float as in double.
//   * u > 1 and u + v > 1 are certain when the numerators exceed |det| by a factor 1.00001 (float rounding of the
//     comparison's operands is 6e-8 relative, the double chain's 3e-16); anything closer goes through the exact chain.
// Only candidates that survive these tests pay for the division.
MCPT_DI bool tri_hit(const TriGeom &g, const Ray &r, double &t_out, double &u_out, double &v_out) {
    const f3 v0 = mk3(g.v0[0], g.v0[1], g.v0[2]);
    const f3 e1 = mk3(g.e1x, g.e1yz[0], g.e1yz[1]);
    const f3 e2 = mk3(g.e2xy[0], g.e2xy[1], g.e2z);
    const f3 pvec = cross(r.d, e2);
    const float det_f = dot(e1, pvec);
    if (fabsf(det_f) < kEps) return false;  // also false for a NaN det, as fabs(det) < EPSILON is
    const f3 tvec = r.o - v0;
    const float a_f = dot(tvec, pvec);
    const float ad = fabsf(det_f), lim = ad * 1.00001f;
    const float a_s = (det_f < 0.f) ? -a_f : a_f;  // numerator of u with the sign of det folded in: u = a_s / |det|
    if (a_s < 0.f || a_s > lim) return false;    // u < 0, or u > 1 for certain
    const f3 qvec = cross(tvec, e1);
    const float b_f = dot(r.d, qvec);
    const float b_s = (det_f < 0.f) ? -b_f : b_f;
    if (b_s < 0.f || a_s + b_s > lim) return false;  // v < 0, or u + v > 1 for certain
    const float c_f = dot(e2, qvec);
    const float c_s = (det_f < 0.f) ? -c_f : c_f;
    if (c_s < 0.f) return false;                   // t < 0
    // the reference's chain for the survivors (NaN numerators arrive here too and fail the comparisons below as they do there)
    const double det = (double)det_f;
    const double det_inv = 1. / det;
    const double u = (double)a_f * det_inv;
    if (u < 0 || u > 1) return false;
    const double v = (double)b_f * det_inv;
    if (v < 0 || u + v > 1) return false;
    const double t = (double)c_f * det_inv;
    if (t < 0) return false;
    t_out = t;
    u_out = u;
    v_out = v;
    return true;
}

// solveQuadratic, global.hpp:20-35
MCPT_DI bool solve_quadratic(float a, float b, float c, float &x0, float &x1) {
    const float discr = b * b - 4 * a * c;
    if (discr < 0) return false;
    if (discr == 0) {
        x0 = x1 = (float)(-0.5 * (double)b / (double)a);
    } else {
        const float q = (b > 0) ? (float)(-0.5 * (double)(b + sqrtf(discr))) : (float)(-0.5 * (double)(b - sqrtf(discr)));
        x0 = q / a;
        x1 = c / q;
    }
    if (x0 > x1) {
        const float t = x0;
        x0 = x1;
        x1 = t;
    }
    return true;
}

// Sphere::getIntersection, Sphere.hpp:26-48
MCPT_DI bool sphere_hit(const SphereRec &s, const Ray &r, float &t_out) {
    const f3 L = r.o - mk3(s.c[0], s.c[1], s.c[2]);
    const float a = dot(r.d, r.d);
    const float b = 2 * dot(r.d, L);
    const float c = dot(L, L) - s.radius2;
    float t0, t1;
    if (!solve_quadratic(a, b, c, t0, t1)) return false;
    if (t0 < 0) t0 = t1;
    if (t0 < 0) return false;
    t_out = t0;
    return true;
}

// ---------------------------------------------------------------- Material (Material.hpp)
MCPT_DI float wavelen(int ch) { return ch == 0 ? 0.700f : (ch == 1 ? 0.5461f : 0.4358f); }  // WaveLen.hpp:7-18

MCPT_DI float D_GGX(f3 h, f3 n, float alpha) {  // Material.hpp:26-34
    const float NoH = fabsf(dot(n, h));
    if (NoH <= kEps && NoH >= -kEps) return 0.0f;
    const float tanTheta = sqrtf(1.0f - NoH * NoH) / NoH;
    const float alpha2 = alpha * alpha;
    const float denom = (NoH * NoH) * (alpha + tanTheta * tanTheta);
    return alpha2 / (kPi * denom * denom);
}

MCPT_DI float G1_SmithGGX(f3 v, f3 n, float alpha) {  // Material.hpp:38-69
    const float NoV = fabsf(dot(n, v));
    if (NoV <= kEps && NoV >= -kEps) return 0.0f;
    const float tanTheta = sqrtf(1.0f - NoV * NoV) / NoV;
    if (tanTheta == 0.0f) return 1.0f;
    const float al_tan = alpha * tanTheta;
    return (float)(2. / (1. + (double)sqrtf(1 + al_tan * al_tan)));
}

MCPT_DI float G_SmithGGX(f3 wi, f3 wo, f3 n, float alpha) { return G1_SmithGGX(wi, n, alpha) * G1_SmithGGX(wo, n, alpha); }  // :70-77

MCPT_DI float get_reflectance(const MaterialRec &m, f2 uv, int ch) {  // Material.hpp:134-151
    if (!m.textured) return comp(m.refl, ch);
    const int col = (int)((uv.x - 0.05f) * 10);
    const int row = (int)((uv.y - 0.00f) * 12);
    if (col >= 3 && col <= 5 && row <= 7) {
        const bool isWhite = (col + row) % 2 == 1;
        return isWhite ? 0.9f : 0.1f;
    }
    return 0.1f;
}

MCPT_DI float fresnel_schlick(const MaterialRec &m, float cosTheta, f2 uv, int ch) {  // Material.hpp:80-86
    const float f = get_reflectance(m, uv, ch);
    const float invc = 1.f - cosTheta;
    const float c2 = invc * invc;
    return f + (1.f - f) * c2 * c2 * invc;
}

MCPT_DI f3 tan_to_world(f3 t, f3 n) {  // Material.hpp:95-106
    f3 T;
    if (fabsf(n.x) > fabsf(n.y)) {
        const float invLen = 1.0f / sqrtf(n.x * n.x + n.z * n.z);
        T = mk3(-n.z * invLen, 0.0f, n.x * invLen);
    } else {
        const float invLen = 1.0f / sqrtf(n.y * n.y + n.z * n.z);
        T = mk3(0.0f, n.z * invLen, -n.y * invLen);
    }
    const f3 B = cross(n, T);
    return (T * t.x + B * t.y) + n * t.z;
}

MCPT_DI f3 importance_sample_ggx(float xi_x, float xi_y, float alpha, f3 n) {  // Material.hpp:111-123
    const float phi = 2.0f * kPi * xi_x;
    const float cosTheta = sqrtf((1.0f - xi_y) / (1.0f + (alpha * alpha - 1.0f) * xi_y));
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    float sp, cp;
    mcpt_sincosf(phi, &sp, &cp);  // std::cos / std::sin, Material.hpp:117-118
    const f3 h = mk3(sinTheta * cp, sinTheta * sp, cosTheta);
    return normalized(tan_to_world(h, n));
}

MCPT_DI float get_ior(const MaterialRec &m, int ch) { return comp(m.ior, ch); }  // Material.hpp:178-183, tabulated per channel by the host

MCPT_DI f3 mat_reflect(f3 I, f3 N) { return N * (2 * dot(N, I)) - I; }  // Material.hpp:195-197

MCPT_DI float mat_fresnel(const MaterialRec &m, f3 I, f3 N, int ch) {  // Material.hpp:198-226
    if (m.type == MCPT_SMOOTH_CONDUCTOR || m.type == MCPT_ROUGH_CONDUCTOR) return 1;
    float cosi = clampf(-1, 1, dot(I, N));
    float etai = 1, etat = get_ior(m, ch);
    if (cosi > 0) {
        const float t = etai;
        etai = etat;
        etat = t;
    }
    const float sint = etai / etat * sqrtf(std_max(0.f, 1 - cosi * cosi));
    if (sint >= 1) return 1;
    const float cost = sqrtf(std_max(0.f, 1 - sint * sint));
    cosi = fabsf(cosi);
    const float Rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
    const float Rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
    return (Rs * Rs + Rp * Rp) / 2;
}

MCPT_DI f3 mat_refract(const MaterialRec &m, f3 I, f3 N, int ch) {  // Material.hpp:227-242
    float cosi = clampf(-1, 1, dot(I, N));
    float etai = 1, etat = get_ior(m, ch);
    f3 n = N;
    if (cosi < 0) {
        cosi = -cosi;
    } else {
        const float t = etai;
        etai = etat;
        etat = t;
        n = -N;
    }
    const float eta = etai / etat;
    const float k = 1 - eta * eta * (1 - cosi * cosi);
    if (k < 0) return mk3(0, 0, 0);
    return I * eta + n * (eta * cosi - sqrtf(k));
}

MCPT_DI f3 mat_sample(const MaterialRec &m, f3 N, float xi_x, float xi_y) {  // Material.hpp:268-281
    if (m.type == MCPT_ROUGH_CONDUCTOR || m.type == MCPT_ROUGH_DIELECTRIC) return importance_sample_ggx(xi_x, xi_y, m.roughness, N);
    return N;
}

MCPT_DI float eta_of(const MaterialRec &m, f3 wi, f3 N, int ch) {  // Material.hpp:299,318,360,393
    return (dot(wi, N) > 0) ? comp(m.ior, ch) : comp(m.inv_ior, ch);
}

MCPT_DI float mat_pdf(const MaterialRec &m, f3 wi, f3 wo, f3 N, int ch, bool isReflect) {  // Material.hpp:285-328 (rough branch)
    f3 h;
    float jacobian;
    if (isReflect) {
        h = normalized(wi + wo);
        h = (dot(wi, N) > 0) ? h : -h;
        jacobian = 1.0f / (4.0f * fabsf(dot(h, wo)));
    } else {
        const float eta = eta_of(m, wi, N, ch);
        const f3 hv = (-wi) - wo * eta;
        h = normalized(hv);
        const float d1 = dot(hv, hv);
        jacobian = eta * eta * fabsf(dot(h, wo)) / d1;
    }
    const float D = D_GGX(h, N, m.roughness);
    return D * dot(N, h) * jacobian;
}

MCPT_DI float mat_eval(const MaterialRec &m, f3 wi, f3 wo, f3 N, int ch, f2 uv, bool isReflect) {  // Material.hpp:330-408
    const bool rough = (m.type == MCPT_ROUGH_CONDUCTOR || m.type == MCPT_ROUGH_DIELECTRIC);
    if (rough) {
        if (isReflect) {
            if (dot(wi, N) * dot(wo, N) <= 0) return 0.f;
            f3 h = normalized(wi + wo);
            h = dot(wi, N) > 0 ? h : -h;
            const float F = (m.type == MCPT_ROUGH_CONDUCTOR) ? fresnel_schlick(m, fabsf(dot(h, wo)), uv, ch) : mat_fresnel(m, -wi, h, ch);
            const float D = D_GGX(h, N, m.roughness);
            const float G = G_SmithGGX(wi, wo, h, m.roughness);
            const float denom = 4.0f * fabsf(dot(N, wi)) * fabsf(dot(N, wo)) + kEps;
            return F * D * G / denom;
        }
        if (m.type == MCPT_ROUGH_CONDUCTOR || dot(wi, N) * dot(wo, N) >= 0) return 0.f;
        const float eta = eta_of(m, wi, N, ch);
        f3 h = normalized((-wi) - wo * eta);
        h = dot(h, N) > 0 ? h : -h;
        const float F = mat_fresnel(m, -wi, h, ch);
        const float D = D_GGX(h, N, m.roughness);
        const float G = G_SmithGGX(wi, wo, h, m.roughness);
        const float hol = dot(h, wi);
        const float hov = dot(h, wo);
        float den = hol + eta * hov;
        den *= den;
        den *= fabsf(dot(N, wi) * dot(N, wo));
        return (1.0f - F) * D * G * eta * eta * fabsf(hol * hov) / den;
    }
    if (isReflect) {
        f3 h = normalized(wi + wo);
        h = (dot(wi, N) > 0) ? h : -h;
        if (dot(wi, N) * dot(wo, N) <= 0 || dot(h, N) < 1 - kEps) return 0.f;
        return (m.type == MCPT_SMOOTH_CONDUCTOR) ? fresnel_schlick(m, fabsf(dot(N, wo)), uv, ch) : mat_fresnel(m, -wi, N, ch);
    }
    const float eta = eta_of(m, wi, N, ch);
    f3 h = normalized((-wi) - wo * eta);
    h = (dot(h, N) > 0) ? h : -h;
    if (m.type == MCPT_SMOOTH_CONDUCTOR || dot(wi, N) * dot(wo, N) >= 0 || dot(h, N) < 1 - kEps) return 0.f;
    return (float)(1. - (double)mat_fresnel(m, -wi, N, ch));
}

// Material::eval and Material::pdf of a ROUGH material for the same (wi, wo) in one go (Scene.cpp:140-143,167-170 calls both): the two
// functions build the same half vector and the same D_GGX term (D depends on |N.h| only, so eval's sign-flipped h and pdf's unflipped one
// give the same D); sharing them saves a normalisation and a D_GGX per continuing rough vertex.  Every expression is the one of
// mat_eval / mat_pdf above, so the values are bit-identical to calling the two separately.
MCPT_DI void mat_eval_pdf_rough(const MaterialRec &m, f3 wi, f3 wo, f3 N, int ch, f2 uv, bool isReflect, float &ev, float &pd) {
    const float wiN = dot(wi, N), woN = dot(wo, N);
    if (isReflect) {
        const f3 hb = normalized(wi + wo);
        const f3 h = (wiN > 0) ? hb : -hb;
        const float D = D_GGX(h, N, m.roughness);
        pd = D * dot(N, h) * (1.0f / (4.0f * fabsf(dot(h, wo))));   // Material.hpp:293-308
        if (wiN * woN <= 0) {
            ev = 0.f;
            return;
        }
        const float F = (m.type == MCPT_ROUGH_CONDUCTOR) ? fresnel_schlick(m, fabsf(dot(h, wo)), uv, ch) : mat_fresnel(m, -wi, h, ch);
        const float G = G_SmithGGX(wi, wo, h, m.roughness);
        const float denom = 4.0f * fabsf(dot(N, wi)) * fabsf(dot(N, wo)) + kEps;
        ev = F * D * G / denom;                                      // Material.hpp:338-352
        return;
    }
    const float eta = eta_of(m, wi, N, ch);
    const f3 hv = (-wi) - wo * eta;
    const f3 hb = normalized(hv);
    const float D = D_GGX(hb, N, m.roughness);
    const float d1 = dot(hv, hv);
    pd = D * dot(N, hb) * (eta * eta * fabsf(dot(hb, wo)) / d1);     // Material.hpp:309-327
    if (m.type == MCPT_ROUGH_CONDUCTOR || wiN * woN >= 0) {
        ev = 0.f;
        return;
    }
    const f3 h = dot(hb, N) > 0 ? hb : -hb;
    const float F = mat_fresnel(m, -wi, h, ch);
    const float G = G_SmithGGX(wi, wo, h, m.roughness);
    const float hol = dot(h, wi);
    const float hov = dot(h, wo);
    float den = hol + eta * hov;
    den *= den;
    den *= fabsf(dot(N, wi) * dot(N, wo));
    ev = (1.0f - F) * D * G * eta * eta * fabsf(hol * hov) / den;   // Material.hpp:354-372
}

// ---------------------------------------------------------------- environment, Scene.hpp:60-99
MCPT_DI f3 sample_env(const DevScene &S, f3 dir) {
    if (S.env_w <= 0) return mk3(S.background[0], S.background[1], S.background[2]);
    const f3 d = normalized(dir);
    const float phi = mcpt_atan2f(d.z, d.x);  // Scene.hpp:66-67
    const float theta = mcpt_acosf(d.y);
    float u = (phi + kPi) / (2.f * kPi);
    float v = theta / kPi;
    u = u - floorf(u);
    v = v < 0.f ? 0.f : (1.f < v ? 1.f : v);
    const float x = u * S.env_w - 0.5f;
    const float y = v * S.env_h - 0.5f;
    const int x0 = (int)floorf(x);
    const int y0 = (int)floorf(y);
    const int W = S.env_w, H = S.env_h;
    int X0 = x0 % W;
    if (X0 < 0) X0 += W;
    int X1 = (x0 + 1) % W;
    if (X1 < 0) X1 += W;
    const int Y0 = y0 < 0 ? 0 : (y0 > H - 1 ? H - 1 : y0);
    const int Y1 = y0 + 1 < 0 ? 0 : (y0 + 1 > H - 1 ? H - 1 : y0 + 1);
    const float sx = x - x0, sy = y - y0;
    const float *e = S.env;
    const f3 c00 = mk3(e[3 * ((size_t)Y0 * W + X0)], e[3 * ((size_t)Y0 * W + X0) + 1], e[3 * ((size_t)Y0 * W + X0) + 2]);
    const f3 c10 = mk3(e[3 * ((size_t)Y0 * W + X1)], e[3 * ((size_t)Y0 * W + X1) + 1], e[3 * ((size_t)Y0 * W + X1) + 2]);
    const f3 c01 = mk3(e[3 * ((size_t)Y1 * W + X0)], e[3 * ((size_t)Y1 * W + X0) + 1], e[3 * ((size_t)Y1 * W + X0) + 2]);
    const f3 c11 = mk3(e[3 * ((size_t)Y1 * W + X1)], e[3 * ((size_t)Y1 * W + X1) + 1], e[3 * ((size_t)Y1 * W + X1) + 2]);
    const f3 c0 = c00 * (1 - sx) + c10 * sx;
    const f3 c1 = c01 * (1 - sx) + c11 * sx;
    return c0 * (1 - sy) + c1 * sy;
}

}  // namespace mcpt
