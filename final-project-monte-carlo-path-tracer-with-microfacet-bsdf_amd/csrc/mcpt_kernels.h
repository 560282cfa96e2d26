// Kernel-launch interface between the C-ABI layer (mcpt_api.cpp) and the HIP kernels (mcpt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>

#include "mcpt_device.h"

namespace mcpt {

// Path record flags (rec0.z): depth in the low 16 bits.
constexpr uint32_t kFresh = 1u << 16;      // no pending vertex: the record's ray is the path's first ray
constexpr uint32_t kTerminate = 1u << 17;  // Russian roulette failed at the pending vertex (Scene.cpp:129,156)
constexpr uint32_t kInside = 1u << 18;     // wo.n < 0 at the pending vertex (Scene.cpp:115)
constexpr uint32_t kPassBit = 1u << 20;    // which of the two passes in flight the path belongs to (pass index & 1)
constexpr uint32_t kNoDirect = 1u << 19;   // every light sample of the pending vertex contributes exactly 0: contrib[] not written

// Device counters.  Every hot word sits on its own 128-byte line: the allocation atomics of different
// queues then never serialise on one L2 line / memory channel.
struct alignas(128) HotCounter {
    uint32_t v;
    uint32_t pad[31];
};
// The shadow-ray queue is written by k_direct without workgroup barriers: every WAVE reserves its entries with one atomic per kind, on the
// counter of its shard (wave index modulo kShadowShards), so that no single counter sees more than 1/32 of the atomics.  Shard s owns the
// region [s R, (s + 1) R) of the queue arrays, R = shadow_region(capacity): entries whose light sample was already found in the window
// fill it from the front, the others from the back; k_trace_shadow turns a queue position into an entry with the prefix sums of the 32
// counts.  (A compact queue with one counter per workgroup needed two barriers per round of k_direct: 40 % of the kernel.)
constexpr uint32_t kShadowShards = 32;
__host__ __device__ inline uint32_t shadow_region(uint32_t capacity) {  // entries per shard: whole waves, enough for every wave that maps to it
    return (((capacity + 63u) / 64u + kShadowShards - 1u) / kShadowShards) * 64u;
}
struct Counters {
    HotCounter n_paths[2];  // records in path list 0 / 1
    HotCounter n_rays[2];   // entries in closest-hit queue 0 / 1
    HotCounter n_direct[2]; // vertices that need direct lighting: length of the k_direct work list (same indexing)
    // Free clamp-stack slots: a ring of a power-of-two number of entries.  k_primary pops at `free_head`, k_shade pushes at
    // `free_tail` (both only ever increase; entry = counter & mask), so the two kernels can run concurrently: the host only
    // lets k_primary pop entries that were pushed by kernels which have already completed.
    HotCounter free_head, free_tail;
    HotCounter n_prays[2];  // new samples of list 0 / 1 (Wave::fresh); their rays are stored from the END of the ray arrays
    HotCounter live[2];     // unfinished paths of the pass with that parity (a pass is complete when it reaches 0)
    HotCounter pushes;      // recursion levels entered (castRay depth+1 calls); folded into tot_pushes by k_bookkeep
    HotCounter overflow;    // cumulative: paths cut by max_depth
    HotCounter ended;       // vertices shaded and finished in the same k_shade call (no record); folded into tot_ended
    // cumulative totals kept on the device by k_bookkeep (the host does not see every iteration's counts)
    unsigned long long tot_shaded, tot_direct, tot_shadow, tot_cont, tot_iterations, tot_pushes, tot_ended;
    uint32_t last_shadow;  // length of the shadow queue k_bookkeep cleared last (the host sizes the next k_trace_shadow grid from it)
    // ---- everything above is what the host reads back every iteration (kCountersHeadBytes); the sharded counters stay on the device
    HotCounter n_shadow[2][kShadowShards];    // per shard: shadow rays whose light sample was found (indexed like the list the rays belong to)
    HotCounter n_shadow_w[2][kShadowShards];  // per shard: shadow rays that still need the window search; stored from the END of the shard's region
};
constexpr size_t kCountersHeadBytes = offsetof(Counters, n_shadow);

// One side of the double-buffered wavefront state (all SoA, 16-byte records, indexed by list position).
struct Wave {
    uint4 *rec0;      // {pid, ray index, depth|flags, kr bits}
    float4 *rec1;     // {eval, |wo.n| or -1 for Dirac, pdf, clamp-stack slot bits}; not written for kTerminate records,
                      // which keep the slot in rec0.y (they have no ray), nor for fresh ones (slot in rec0.w)
    float4 *ray_o;    // closest-hit queue: origin
    float4 *ray_d;    // closest-hit queue: direction
    uint4 *hit;       // closest-hit results: {t lo, t hi, prim, 0}
    float *contrib;   // n_dir per path record: light-sample contributions; zeroed by the shadow kernel when invisible
    uint2 *fresh;     // new samples with a surface hit, entry k (its ray and hit: entry ray_cap-1-k of the ray arrays):
                      // {sample | pass parity << 31, position of its three slots in the free ring}.  k_shade gives each one
                      // three lanes (the channel paths) after the records of the list: no per-path record is written for them
};

// Per-iteration scratch between k_shade, k_direct and k_trace<shadow> (single-buffered: produced and consumed
// inside one iteration).
struct Scratch {
    float4 *vtx0;   // k_direct work list: {q.xyz (offset shading point, Scene.cpp:114), uv.x}
    float4 *vtx1;   // {n.xyz, uv.y}
    float4 *vtx2;   // {wo.xyz, bits: material | channel << 16 | inside << 18}
    uint32_t *vtx_j;  // index of the vertex's record in the next path list
    float4 *shq_o;  // compacted shadow queue: {origin.xyz, bits: index into contrib}.  Rays whose light sample was found by
                    // k_direct's own test of the sampled primitive fill it from the front (occluder search only), the
                    // others from the back (window search first)
    float4 *shq_d;  // {direction.xyz, distance to the light sample}
};

// Per-pixel candidate primitives for the primary rays (csrc/mcpt_cull.hip)
constexpr int32_t kCandNone = 0x7fffffff;      // unused entry
constexpr int32_t kCandTraverse = 0x7ffffffe;  // in .x: no short list for this pixel, walk the tree

struct RenderConst {
    float rr_rate, inv_rr;
    int32_t n_dir, enable_shadow;
    uint32_t seed;
    int32_t mode;  // 0: pid -> (pixel list, sample); 1: explicit per-path keys (mcpt_cast_rays)
    const uint32_t *pixel_list;
    const int4 *pixel_cand;  // aligned with pixel_list, or nullptr (every primary ray walks the tree)
    int32_t s_pass[2], sample_offset[2];  // per pass parity: two passes can be in flight
    int32_t s_pass_shift[2];              // log2(s_pass) when it is a power of two (the usual 256: a shift instead of a division), else -1
    const uint32_t *key_pixel, *key_sample;
    const int32_t *key_channel;
    int32_t max_depth;
    int32_t track_live;  // maintain Counters::live (only needed when several passes are pipelined)
    uint32_t pool;  // clamp-stack row length (slots)
    float4 *stack;  // [level][slot] = {clamp(0,15,l_dir), eval, |wo.n| or -1, pdf}
    float *result[2];  // per pass parity, per pid: castRay(ray, 0, channel)
    uint32_t *free_slots;
    uint32_t free_mask;  // ring size - 1
    uint32_t ray_cap;    // entries of the ray / hit arrays
    Counters *counters;
};

struct CameraConst {
    int32_t width, height, use_dof;
    float scale, aspect, focal_distance, aperture_radius;
    float eye[3];
    float orient[9];
};

// Small-scene flavour (DevScene::small): what the SMALL kernels can hold in LDS -- nodes (64-B or 32-B records), TriGeom, sphere records
// (one slot per object), materials and light tables -- and the depth of their traversal stack.  The Cornell configurations are 31-34
// nodes, 32 triangles, 9 objects, 7-8 materials, one two-triangle light.
constexpr int kSmallNodes = 64, kSmallTris = 64, kSmallSphereSlots = 16;
constexpr int kSmallMats = 12, kSmallLights = 4, kSmallLightNodes = 8, kSmallLightTris = 8;
constexpr int kSmallStk = 8;  // LDS stack entries of the SMALL kernels: trees of up to 9 levels

// Retrace list of one traversal kernel (retry flavour of the traversal stack, MCPT_STK_PUSH in mcpt_kernels.hip): the rays of a launch that
// lost a stack entry; the retrace kernel launched right behind traces them again with the scratch stack and clears the list.  `cap` is the
// largest number of rays one launch can hold, so the list cannot overflow.
struct RetryList {
    uint32_t *count;  // entries
    uint32_t *done;   // workgroups of the retrace kernel that have finished (the last one clears both counters)
    uint32_t *items;  // ray index (closest), queue position (shadow), sample index (primary)
    uint32_t cap;
};
struct RetryLists {
    RetryList closest, shadow, primary;
};

// Traversal-stack entries per lane the traversal kernels provide for a tree of this height (a ray holds at most one child reference
// per inner ancestor, i.e. height - 1).  0: the tree is too deep for any instantiation.
#ifndef MCPT_STK_B
#define MCPT_STK_B 20
#endif
constexpr int kStkB = MCPT_STK_B;  // entries for trees of height 18..20.  (19 would be exact for height 20 and lets 8 instead of 7 workgroups of k_primary /
// k_trace_shadow share a CU: they gain what k_direct beside them loses -- frame rate +1 %, +0.1 %, -0.8 % in three A/B runs: no change.)
#ifndef MCPT_PLAIN_MAX_HEIGHT
#define MCPT_PLAIN_MAX_HEIGHT 24
#endif
// Deepest tree traversed with an LDS stack that holds every entry (24 entries for 21-24 levels).  Measured on the 296 k-triangle SAH
// tree (24 levels): 24 LDS entries 3970 Msamples/s, retry flavour 3925 -- its traversal kernels are faster with 8 instead of 6
// workgroups per CU, but k_direct beside them loses more than they gain.
constexpr int kPlainMaxHeight = MCPT_PLAIN_MAX_HEIGHT;
static_assert(kPlainMaxHeight == 20 || kPlainMaxHeight == 24, "plain stack classes: 16, kStkB (<= 20 levels), 24");
// true: the traversal kernels of a tree of this height run the retry flavour and need their retrace lists (RetryList)
inline bool stack_uses_retry(int height) {
#if defined(MCPT_LDS_ONLY_STACKS)
    (void)height;
    return false;
#elif defined(MCPT_FORCE_RETRY)
    (void)height;
    return true;
#else
    return height > kPlainMaxHeight;
#endif
}
inline int traversal_stack_entries(int height) {
    return height <= 17 ? 16 : (height <= 20 ? kStkB : (height <= kPlainMaxHeight ? 24 : (height <= kMaxBvhHeight ? kMaxBvhHeight : 0)));  // (deeper: 16 in LDS, and the scratch stack of kMaxBvhHeight entries for the rays that need more)
}

// The two halves of mcpt_scene_create (csrc/mcpt_api.hip), exposed for mcpt_group_create (csrc/mcpt_multi.hip), which flattens the scene
// and builds its tree once and then uploads it to every device from one thread per device.
struct HostBuild {
    HostScene hs;
    BuildChoice choice;
    double build_ms = 0.0;  // flattening + host tree build
    double init_ms = 0.0;   // first use of the device by this process (context, code objects), overlapped with the host build
};
int build_scene_host(const mcpt_scene_desc *desc, const mcpt_build_options *options, HostBuild &hb);
int upload_scene(const mcpt_scene_desc *desc, HostBuild &hb, int device, mcpt_scene **out);
double warm_up_device(int device);
void set_device_sharers(mcpt_scene *sc, int n);  // replicas of a group on one device divide its free memory between them

void launch_init_free(uint32_t *free_slots, Counters *c, uint32_t pool, uint32_t start, uint32_t mask, hipStream_t s);
// After k_shade(cur -> next): adds this iteration's list lengths to the cumulative totals and clears the counters of
// list `cur` (consumed; it is the next iteration's output list), in one launch.
void launch_bookkeep(Counters *c, int cur_idx, bool from_host, uint32_t n_next, uint32_t n_cont, uint32_t n_direct, hipStream_t s);
// Camera ray + closest hit for `n_samples` new samples, fused: a miss or a depth-0 emitter hit writes the
// three channel results directly; any other hit appends one ray/hit entry and three fresh path records to
// wave `next` (list index `next_idx`).
void launch_primary(const DevScene &S, const CameraConst &cam, const RenderConst &C, Wave next, int next_idx, int parity,
                    uint32_t first_sample, uint32_t n_samples, const RetryList &rl, hipStream_t s);
void launch_generate_explicit(const RenderConst &C, Wave next, int next_idx, uint32_t n, hipStream_t s);
void launch_camera_rays(const CameraConst &cam, uint32_t seed, uint32_t n, const uint32_t *pixel, const uint32_t *sample,
                        float4 *o, float4 *d, hipStream_t s);
// n_dev != nullptr: the ray count is read on the device (n then only sizes the grid, which strides over the queue).
void launch_trace_closest(const DevScene &S, uint32_t n, const uint32_t *n_dev, const float4 *ray_o, const float4 *ray_d, uint4 *hit,
                          const RetryList &rl, hipStream_t s);
// Direct lighting (Scene::directLighting, Scene.cpp:56-82) for the n_vertices entries of the k_direct work list:
// one lane per (vertex, light sample); fills next.contrib and appends the non-zero samples to the shadow queue.  The list
// length is read on the device; the grid (n_vertices_grid vertices) strides over it.
void launch_direct(const DevScene &S, const RenderConst &C, Wave next, Scratch X, int next_idx, uint32_t n_vertices_grid, uint32_t small_per_cu, hipStream_t s);
// Shadow queue (lengths in counters->n_shadow / n_shadow_w, together at most n_max; arrays of `cap` entries): zeroes
// contrib[] of invisible samples.
void launch_trace_shadow(const DevScene &S, const Counters *counters, int next_idx, uint32_t n_max, uint32_t cap, Scratch X,
                         float *contrib, uint32_t grid_per_cu, const RetryList &rl, hipStream_t s);
// Shades list `cur_idx` (at most n_cur_max records; the true count is read from the device counter) into the other list.
void launch_shade(const DevScene &S, const RenderConst &C, Wave cur, Wave next, Scratch X, int cur_idx, uint32_t n_cur_max,
                  hipStream_t s);
// Multi-GPU merge helpers (csrc/mcpt_multi.hip): zero the pixels rank `rank` does not own (tile rule of mcpt_params); a += b.
void launch_mask_unowned(float *fb, int width, int height, int tile, int rank, int nranks, hipStream_t s);
void launch_add_frame(float *a, const float *b, uint32_t n, hipStream_t s);
// Tone map of Renderer.cpp:95-103: n_pix RGB float triples -> n_pix RGBA8 (csrc/mcpt_fmath.h: mcpt_tonemap_byte).
void launch_tonemap(const float *fb, uint32_t n_pix, unsigned char *rgba, hipStream_t s);
void launch_debug_fmath(int kind, uint32_t n, const float *x, const float *y, float *out, hipStream_t s);
void launch_debug_scene(const DevScene &S, int kind, uint32_t n, const float *in, float *out, hipStream_t s);
void launch_debug_material(const DevScene &S, int kind, uint32_t n, const float *in, const int32_t *sel, float *out, hipStream_t s);
void launch_accumulate(const float *result, const uint32_t *pixel_list, uint32_t n_pix, int32_t s_pass, float spp_total,
                       float *fb, hipStream_t s);

}  // namespace mcpt
