// Sky-pixel culling: pixels none of whose camera rays can hit anything are finished without tracing a single ray.
//
// Renderer::Render traces every sample of every pixel (Renderer.cpp:36-80); for a pixel that sees only the background every one of
// its spp samples returns the same constant, castRay's miss value (Scene.cpp:88-95, Scene.hpp:61-63 without an environment map), and
// the pixel's value is that constant summed spp times (Renderer.cpp:80).  Which pixels those are is decided CONSERVATIVELY, per pixel,
// before the wavefront loop starts:
//
//   Every camera ray of pixel (i, j) runs from a lens point L = eye + O (dx, dy, 0), |(dx, dy)| <= R (aperture radius; 0 without
//   depth of field), through a focal point P = eye + O (x F, y F, F) with (x, y) inside the pixel's jitter square (Renderer.cpp:44-76).
//   Against the central ray X0(s) = eye + s O fp0 (lens centre, pixel centre), a point X(s) = L + s (P - L) of any sample ray satisfies
//       |X(s) - X0(s)| = |(1 - s)(dx, dy, 0) + s (fp - fp0)| <= |1 - s| R + s h,     h = F sqrt((aspect scale / W)^2 + (scale / H)^2),
//   for the SAME parameter s >= 0.  Inside the scene's root box s <= s_far, so the deviation is at most
//       rho = max(R, |1 - s_far| R + s_far h).
//   Hence: if a sample ray enters a box B, the central ray (s >= 0) enters B widened by rho on every side.  The classifier walks the
//   tree with the central ray against boxes widened by 1.05 rho + a rounding allowance; reaching any leaf marks the pixel "may hit".
//   A pixel that reaches no leaf cannot have a sample ray inside any leaf box, i.e. every sample misses every primitive.
//
// The widening is generous (a dozen units in the 5000-unit chess scene), so a ring of pixels around every silhouette is traced as
// before; about 60 % of the chess frame is culled.  Culled pixels get exactly the value the wavefront would give them: spp additions of
// background[c] / spp_total in sample order (k_sky_fill = k_accumulate's arithmetic).  Not used with an environment map (the miss
// value then depends on each sample's direction) -- nothing is culled there.  tests/test_gpu_cull.py: frames bit-identical with the
// culling on and off, thin geometry and depth of field included; MCPT_SKY_CULL=0 switches it off.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>

#include "mcpt_cull.h"

namespace mcpt {

namespace {

constexpr int kB = 256;
inline uint32_t nblocks(uint32_t n) { return (n + kB - 1) / kB; }

struct CullConst {
    float eye[3], orient[9];
    float scale, aspect, focal, lens;  // focal distance (1 without DoF), aperture radius (0 without DoF)
    int32_t width, height;
    float rho;  // widening of every box
};

// central ray (origin o, direction d, parameter s >= 0) against [mn - rho, mx + rho]
__device__ __forceinline__ bool beam_box(const float o[3], const float d[3], const float mn[3], const float mx[3], float rho) {
    float smin = 0.f, smax = INFINITY;
    for (int a = 0; a < 3; ++a) {
        const float lo = mn[a] - rho, hi = mx[a] + rho;
        if (fabsf(d[a]) > 1e-30f) {
            const float s1 = (lo - o[a]) / d[a], s2 = (hi - o[a]) / d[a];
            smin = fmaxf(smin, fminf(s1, s2));
            smax = fminf(smax, fmaxf(s1, s2));
        } else if (o[a] < lo || o[a] > hi) {
            return false;
        }
    }
    return smax * 1.0001f + 1e-6f >= smin;  // (the slack errs towards "enters")
}

// Besides the sky / may-hit flag, the walk records WHICH leaves the widened central ray reaches when they are few (at most four
// primitives): every primitive a sample ray of the pixel can hit is among them, so k_primary tests just those primitives instead of
// walking the tree (a pixel that sees only the floor: two triangles).  cand.x == kCandTraverse: too many (or an instance): traverse.
__global__ __launch_bounds__(kB) void k_classify(DevScene S, CullConst C, const uint32_t *__restrict__ pixels, uint32_t n, uint8_t *__restrict__ may_hit,
                                                  int4 *__restrict__ cand) {
    __shared__ int32_t stk[kMaxBvhHeight + 2][kB];
    const uint32_t g = blockIdx.x * kB + threadIdx.x;
    if (g >= n) return;
    const int tid = threadIdx.x;
    const uint32_t m = pixels[g];
    const int i = (int)(m % (uint32_t)C.width), j = (int)(m / (uint32_t)C.width);
    const float x0 = (1.f - 2.f * (i + 0.5f) / (float)C.width) * C.aspect * C.scale;
    const float y0 = (1.f - 2.f * (j + 0.5f) / (float)C.height) * C.scale;
    const float fp[3] = {x0 * C.focal, y0 * C.focal, C.focal};
    float d[3];
    for (int a = 0; a < 3; ++a) d[a] = C.orient[3 * a] * fp[0] + C.orient[3 * a + 1] * fp[1] + C.orient[3 * a + 2] * fp[2];
    const float *o = C.eye;
    bool hit = false;
    int32_t c[4] = {kCandNone, kCandNone, kCandNone, kCandNone};
    int n_cand = 0;
    if (beam_box(o, d, S.root_min, S.root_max, C.rho)) {
        int32_t cur = S.root;
        int sp = 0;
        while (true) {
            if (cur < 0) {  // a leaf (a primitive or an instance): a sample ray might hit it
                hit = true;
                const bool prim_leaf = (uint32_t)(~cur) < (uint32_t)S.n_leaf_prims || S.inst == nullptr;
                if (!prim_leaf || n_cand == 4) {  // an instance, or a fifth primitive: this pixel's rays walk the tree
                    n_cand = 5;
                    break;
                }
                c[n_cand++] = cur;
                if (sp == 0) break;
                cur = stk[--sp][tid];
                continue;
            }
            const Node N = S.nodes[cur];
            const bool hl = beam_box(o, d, N.lmin, N.lmax, C.rho), hr = beam_box(o, d, N.rmin, N.rmax, C.rho);
            if (hl && hr) {
                stk[sp++][tid] = N.right;
                cur = N.left;
            } else if (hl) {
                cur = N.left;
            } else if (hr) {
                cur = N.right;
            } else {
                if (sp == 0) break;
                cur = stk[--sp][tid];
            }
        }
    }
    may_hit[g] = hit ? 1 : 0;
    cand[g] = n_cand > 4 ? make_int4(kCandTraverse, 0, 0, 0) : make_int4(c[0], c[1], c[2], c[3]);
}

// framebuffer[m] += background / spp, spp times, in order (Renderer.cpp:80 with every sample equal to the miss value)
__global__ __launch_bounds__(kB) void k_sky_fill(const uint32_t *__restrict__ sky_pixels, uint32_t n_sky, float3 background, int32_t spp, float spp_total,
                                                  float *__restrict__ fb) {
    const uint32_t g = blockIdx.x * kB + threadIdx.x;
    if (g >= n_sky * 3u) return;
    const uint32_t m = sky_pixels[g / 3u], c = g % 3u;
    const float v = c == 0 ? background.x : (c == 1 ? background.y : background.z);
    float acc = fb[(size_t)m * 3 + c];
    for (int k = 0; k < spp; ++k) acc += v / spp_total;
    fb[(size_t)m * 3 + c] = acc;
}

}  // namespace

hipError_t cull_sky_pixels(const DevScene &S, const CameraConst &cam, const uint32_t *d_pixels, uint32_t n, uint32_t *d_out, uint8_t *d_flags,
                           int4 *d_cand_tmp, int4 *d_cand_out, void *d_temp, size_t temp_bytes, uint32_t *d_count, uint32_t *n_trace, hipStream_t st) {
    CullConst C;
    for (int a = 0; a < 3; ++a) C.eye[a] = cam.eye[a];
    for (int a = 0; a < 9; ++a) C.orient[a] = cam.orient[a];
    C.scale = cam.scale;
    C.aspect = cam.aspect;
    C.focal = cam.use_dof ? cam.focal_distance : 1.0f;
    C.lens = cam.use_dof ? fabsf(cam.aperture_radius) : 0.0f;
    C.width = cam.width;
    C.height = cam.height;
    // the bound takes |O v| = |v| (Camera::lookAt builds an orthonormal matrix, Camera.hpp:17-24): any other matrix, no culling
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b) {
            double dotp = 0;
            for (int k = 0; k < 3; ++k) dotp += (double)C.orient[3 * k + a] * C.orient[3 * k + b];
            if (std::fabs(dotp - (a == b ? 1.0 : 0.0)) > 1e-4) return hipSuccess;
        }
    // h, s_far, rho (see the header comment), all on the host in double
    const double h = std::fabs((double)C.focal) * std::sqrt(std::pow((double)C.aspect * C.scale / C.width, 2) + std::pow((double)C.scale / C.height, 2));
    double centre[3], half = 0, eye_c = 0, eye_n = 0;
    for (int a = 0; a < 3; ++a) {
        centre[a] = 0.5 * ((double)S.root_min[a] + S.root_max[a]);
        half += std::pow(0.5 * ((double)S.root_max[a] - S.root_min[a]), 2);
        eye_c += std::pow((double)C.eye[a] - centre[a], 2);
        eye_n += (double)C.eye[a] * C.eye[a];
    }
    half = std::sqrt(half);
    const double reach = std::sqrt(eye_c) + half + C.lens;            // farthest scene point from any lens point
    const double fmin = std::fabs((double)C.focal) - h - C.lens;      // shortest |P - L| (|fp0| >= focal)
    if (!(fmin > 0.05 * std::fabs((double)C.focal)) || !std::isfinite(reach) || !std::isfinite(h)) return hipSuccess;  // odd camera: no culling
    const double s_far = reach / fmin;
    const double rho = std::max((double)C.lens, std::fabs(1.0 - s_far) * C.lens + s_far * h);
    C.rho = (float)(1.05 * rho + 1e-4 * (half + std::sqrt(eye_n) + std::sqrt(eye_c)) + 1e-3);
    if (!std::isfinite(C.rho)) return hipSuccess;
    hipLaunchKernelGGL(k_classify, dim3(nblocks(n)), dim3(kB), 0, st, S, C, d_pixels, n, d_flags, d_cand_tmp);
    hipError_t e = hipcub::DevicePartition::Flagged(d_temp, temp_bytes, d_pixels, d_flags, d_out, d_count, (int)n, st);
    if (e != hipSuccess) return e;
    e = hipcub::DevicePartition::Flagged(d_temp, temp_bytes, d_cand_tmp, d_flags, d_cand_out, d_count, (int)n, st);  // the same order
    if (e != hipSuccess) return e;
    uint32_t cnt = 0;
    e = hipMemcpyAsync(&cnt, d_count, sizeof cnt, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) return e;
    *n_trace = cnt;
    return hipSuccess;
}

size_t cull_temp_bytes(uint32_t n) {
    size_t a = 0, b = 0;
    (void)hipcub::DevicePartition::Flagged(nullptr, a, (const uint32_t *)nullptr, (const uint8_t *)nullptr, (uint32_t *)nullptr, (uint32_t *)nullptr, (int)n);
    (void)hipcub::DevicePartition::Flagged(nullptr, b, (const int4 *)nullptr, (const uint8_t *)nullptr, (int4 *)nullptr, (uint32_t *)nullptr, (int)n);
    return a > b ? a : b;
}

void launch_sky_fill(const uint32_t *sky_pixels, uint32_t n_sky, const float background[3], int32_t spp, float spp_total, float *fb, hipStream_t st) {
    if (n_sky == 0) return;
    hipLaunchKernelGGL(k_sky_fill, dim3(nblocks(n_sky * 3u)), dim3(kB), 0, st, sky_pixels, n_sky, make_float3(background[0], background[1], background[2]), spp,
                       spp_total, fb);
}

}  // namespace mcpt
