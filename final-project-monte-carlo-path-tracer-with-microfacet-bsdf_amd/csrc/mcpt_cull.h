// Sky-pixel culling (csrc/mcpt_cull.hip): conservative per-pixel classification before the wavefront loop.
#pragma once
#include <hip/hip_runtime.h>

#include "mcpt_kernels.h"

namespace mcpt {

// Partitions the n owned pixels `d_pixels` into d_out[0, *n_trace) (pixels whose rays may hit something, original order) and
// d_out[*n_trace, n) (pixels that can only see the background).  d_flags: n bytes, d_temp / temp_bytes: cull_temp_bytes(n),
// d_count: one uint32.  Leaves *n_trace == n (nothing culled, d_out untouched) for cameras the bound does not cover.  Synchronises `st`.
// d_cand_out (aligned with d_out): per pixel the at most four primitives its rays can hit (leaf references, kCandNone = unused), or
// kCandTraverse in .x; d_cand_tmp: n entries of scratch.
hipError_t cull_sky_pixels(const DevScene &S, const CameraConst &cam, const uint32_t *d_pixels, uint32_t n, uint32_t *d_out, uint8_t *d_flags,
                           int4 *d_cand_tmp, int4 *d_cand_out, void *d_temp, size_t temp_bytes, uint32_t *d_count, uint32_t *n_trace, hipStream_t st);
size_t cull_temp_bytes(uint32_t n);
// framebuffer[m][c] += background[c] / spp_total, spp times in order, for the culled pixels (what the wavefront would accumulate)
void launch_sky_fill(const uint32_t *sky_pixels, uint32_t n_sky, const float background[3], int32_t spp, float spp_total, float *fb, hipStream_t st);

}  // namespace mcpt
