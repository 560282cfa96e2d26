// HIP kernels of the wavefront path tracer for gfx950 (MI355X, CDNA4; 64-wide wavefronts).
//
// Pipeline per wavefront iteration (host loop in mcpt_api.cpp):
//   k_shade          one lane per path record: resolves the pending vertex (direct-light sum, continuation
//                    hit), finishes the path or pushes a clamp-stack level, samples the BSDF at the next vertex and
//                    emits a vertex record + at most one continuation ray; survivors are stream-compacted into
//                    the next list with ballot/popcount wave-aggregated atomics.
//   k_primary        camera ray + closest hit for new samples, fused (one primary ray feeds the three channel
//                    paths); sky misses and depth-0 emitter hits are finished here and never become records.
//   k_direct         direct lighting, one lane per (vertex, light sample); non-zero samples enter the shadow queue.
//   k_trace_closest  closest hit for continuation rays.
//   k_trace_shadow   shadow queue (persistent grid): the distance-equality visibility of Scene.cpp:75.
//   k_accumulate     per pass: framebuffer[m] += rgb/spp in sample order (Renderer.cpp:80).
//
// Reference logic covered: Renderer.cpp:39-80, Scene.cpp:19-37,56-184, BVH.cpp:95-135,
// Bounds3.hpp:95-108, Triangle.hpp:71-76,193-196,222-252, Sphere.hpp:26-48, Material.hpp:26-408.
#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "mcpt_kernels.h"

namespace mcpt {

namespace {

constexpr int kBlock = 256;
#ifndef MCPT_SHADE_BLOCK
#define MCPT_SHADE_BLOCK 512
#endif
constexpr int kShadeBlock = MCPT_SHADE_BLOCK;  // k_shade: larger workgroups = fewer allocation atomics per hot counter

MCPT_DI uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// Block-aggregated queue allocation.  Up to kMaxAlloc counters are served by ONE round of atomics per
// workgroup (lane k of wave 0 adds the block total of request k), instead of one returning atomic per
// wave and counter: the hot counters are the only cross-workgroup contention points of the pipeline.
// Must be called by every thread of the block (it contains barriers).
constexpr int kMaxAlloc = 9;
constexpr int kMaxWaves = 16;
struct BlockAllocShared {
    uint32_t cnt[kMaxWaves][kMaxAlloc];  // in: requests of wave w; out (after the first barrier): requests of the waves before w
    uint32_t base[kMaxAlloc];
};

// begin: ballots, per-wave counts to LDS, one barrier, then lanes 0..N-1 of wave 0 issue the atomics and publish the
// bases.  end: second barrier, per-lane indices.  Work that does not need the indices can sit between the two calls:
// the other waves then compute while wave 0 waits for its atomics to return.
template <int N>
MCPT_DI void block_alloc_begin(BlockAllocShared &sh, const bool (&want)[N], const uint32_t (&mult)[N], uint32_t *const (&counter)[N],
                               const bool (&subtract)[N], uint32_t (&prefix)[N]) {
    static_assert(N <= kMaxAlloc, "too many allocation requests");
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const unsigned long long mask = __ballot(want[k]);
        prefix[k] = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (lane == 0) sh.cnt[wave][k] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) {
        if (threadIdx.x == (unsigned)k) {  // lanes 0..N-1 of wave 0 issue their atomics in the same instruction
            uint32_t total = 0;
            for (uint32_t w = 0; w < n_waves; ++w) {  // counts -> exclusive prefix over the waves, in place
                const uint32_t c = sh.cnt[w][k];
                sh.cnt[w][k] = total;
                total += c;
            }
            uint32_t base = 0;
            if (total) {
                const uint32_t amount = total * mult[k];
                base = subtract[k] ? (atomicSub(counter[k], amount) - amount) : atomicAdd(counter[k], amount);
            }
            sh.base[k] = base;
        }
    }
}

template <int N>
MCPT_DI void block_alloc_end(BlockAllocShared &sh, const uint32_t (&mult)[N], const uint32_t (&prefix)[N], uint32_t (&index)[N]) {
    const uint32_t wave = threadIdx.x >> 6;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < N; ++k) index[k] = sh.base[k] + (sh.cnt[wave][k] + prefix[k]) * mult[k];
}

template <int N>
MCPT_DI void block_alloc(BlockAllocShared &sh, const bool (&want)[N], const uint32_t (&mult)[N], uint32_t *const (&counter)[N],
                         const bool (&subtract)[N], uint32_t (&index)[N]) {
    uint32_t prefix[N];
    block_alloc_begin<N>(sh, want, mult, counter, subtract, prefix);
    block_alloc_end<N>(sh, mult, prefix, index);
}

MCPT_DI f3 ld3(float4 v) { return mk3(v.x, v.y, v.z); }

// (An XCD-aware block-id remap -- cdna_hip_programming.md T1: every XCD gets a contiguous eighth of the rays, so that neighbouring
// pixels share one L2 -- was measured 24 % SLOWER on the chess frame (3380 vs 4440 Msamples/s): hardware block ids are dealt
// round-robin to the XCDs, which spreads the cheap sky tiles and the expensive glass tiles evenly over them; a contiguous eighth per
// XCD turns the frame's spatial cost variation into XCD load imbalance.  The scene fits every L2 anyway.)

// ------------------------------------------------------------------------------------------------
// Small scenes (the Cornell configurations: 31-34 nodes, 32 triangles, 3 spheres, 7 KB in all): the SMALL instantiations of the
// traversal kernels copy nodes, TriGeom and sphere records into LDS once per workgroup and repoint their by-value DevScene at the
// copy, so every node visit and primitive test reads LDS instead of the vector cache; the traversal stack is 8 entries (trees of up
// to 9 levels).  A wave-uniform template flavour: there is no LDS / global select inside the loop (that select is what sank
// the top-of-tree staging experiment on the chess scene).  The pointers are generic ones derived from __shared__
// arrays; after inlining the compiler's address-space inference turns the loads into ds_read_b128 (tools/kernel_resources.py shows
// no flat_load in the SMALL kernels' loops).
// ------------------------------------------------------------------------------------------------
// (limits: kSmall* in mcpt_kernels.h; mcpt_scene_create decides `DevScene::small` from them)
struct alignas(16) SmallGeomLds {
    uint4 nodes[kSmallNodes * 4];  // Node records (64 B), or QNode records (32 B) in the first half
    TriGeom tri[kSmallTris];
    SphereRec sph[kSmallSphereSlots];
};
struct alignas(16) SmallLightLds {
    MaterialRec mats[kSmallMats];
    LightRec lights[kSmallLights];
    LightNode nodes[kSmallLightNodes];
    LightTri tris[kSmallLightTris];
};
MCPT_DI void lds_copy16(void *dst, const void *src, uint32_t n16) {  // n16 16-byte words, by the whole workgroup
    uint4 *d = reinterpret_cast<uint4 *>(dst);
    const uint4 *g = reinterpret_cast<const uint4 *>(src);
    for (uint32_t i = threadIdx.x; i < n16; i += blockDim.x) d[i] = g[i];
}
// (the caller's __syncthreads follows: a kernel that stages both blocks pays for one barrier)
MCPT_DI void stage_small_geom(DevScene &S, SmallGeomLds &L) {
    // Quantised nodes become "prepared" nodes (traverse_loop, NF = 2): the 16-bit grid coordinates as floats relative to the grid origin,
    // x' = (float)q * cell, in the layout of Node.  Float nodes are copied as they are.  (Every node pointer the SMALL kernels follow ends up
    // as an LDS pointer or null on every path, so that the address-space inference sees no global / LDS mix.)
    const bool quant = S.qnodes != nullptr;
    if (quant) {
        for (uint32_t i = threadIdx.x; i < (uint32_t)S.n_inner; i += blockDim.x) {
            const uint4 *q = reinterpret_cast<const uint4 *>(S.qnodes + i);
            const uint4 a = q[0], b = q[1];
            const float cx = S.q_cell[0], cy = S.q_cell[1], cz = S.q_cell[2];
            float4 *o = reinterpret_cast<float4 *>(L.nodes) + 4 * i;
            o[0] = make_float4((float)(a.x & 0xffffu) * cx, (float)(a.x >> 16) * cy, (float)(a.y & 0xffffu) * cz, (float)(a.y >> 16) * cx);
            o[1] = make_float4((float)(a.z & 0xffffu) * cy, (float)(a.z >> 16) * cz, (float)(a.w & 0xffffu) * cx, (float)(a.w >> 16) * cy);
            o[2] = make_float4((float)(b.x & 0xffffu) * cz, (float)(b.x >> 16) * cx, (float)(b.y & 0xffffu) * cy, (float)(b.y >> 16) * cz);
            o[3] = make_float4(__int_as_float((int32_t)b.z), __int_as_float((int32_t)b.w), 0.f, 0.f);
        }
    } else {
        lds_copy16(L.nodes, S.nodes, (uint32_t)S.n_inner * (uint32_t)(sizeof(Node) / 16));
    }
    S.nodes = reinterpret_cast<const Node *>(L.nodes);
    S.pnodes = quant ? reinterpret_cast<const float4 *>(L.nodes) : nullptr;
    lds_copy16(L.tri, S.tri_geom, (uint32_t)S.n_tri * (uint32_t)(sizeof(TriGeom) / 16));
    lds_copy16(L.sph, S.spheres, (uint32_t)S.n_sphere_slots * (uint32_t)(sizeof(SphereRec) / 16));
    S.tri_geom = L.tri;
    S.spheres = L.sph;
}
MCPT_DI void stage_small_lights(DevScene &S, SmallLightLds &L) {
    lds_copy16(L.mats, S.mats, (uint32_t)S.n_mats * (uint32_t)(sizeof(MaterialRec) / 16));
    lds_copy16(L.lights, S.lights, (uint32_t)S.n_lights * (uint32_t)(sizeof(LightRec) / 16));
    lds_copy16(L.nodes, S.light_nodes, (uint32_t)S.n_light_nodes * (uint32_t)(sizeof(LightNode) / 16));
    lds_copy16(L.tris, S.light_tris, (uint32_t)S.n_light_tris * (uint32_t)(sizeof(LightTri) / 16));
    S.mats = L.mats;
    S.lights = L.lights;
    S.light_nodes = L.nodes;
    S.light_tris = L.tris;
}

// ------------------------------------------------------------------------------------------------
// Traversal.  One lane per ray; the per-lane stack of child references lives in LDS as
// stk[level][thread] so that the 64 lanes of a wave hit 64 consecutive banks.
//
// Result equivalence with BVHAccel::getIntersection (BVH.cpp:103-116), which visits both children and
// never prunes: a primitive is tested here only if every ancestor box test of the reference passes
// (same box test, same tree), children are visited near-first, and a subtree is skipped only when its
// entry distance exceeds the best hit by a margin far above float rounding (closest hit) or lies beyond
// the light sample (shadow rays).  Equal distances go to the larger primitive id.
// ------------------------------------------------------------------------------------------------
struct TraceResult {
    double t;
    int32_t prim;
    uint32_t mat_bits;  // closest hit: material index | kMatTextured | kMatEmissive (TriGeom::mat_bits)
    bool visible;       // shadow queries only
    bool dropped;       // retry flavour: a stack entry was lost, the result is void (the ray goes to the retrace list)
};

// One traversal loop, three query kinds:
//   kClosest   closest hit (BVH.cpp:95-116); subtrees entered beyond the best hit (+margin) are skipped.
//   kWindow    shadow phase A: only subtrees whose [tmin,tmax] overlaps [dist-m, dist+m] are entered.  Finds
//              every hit the visibility test |t - dist| < EPSILON (Scene.cpp:75) could accept; a hit with
//              t <= dist - EPSILON met on the way proves occlusion at once.
//   kOccluder  shadow phase B: any hit with t <= dist - EPSILON ends the search (subtrees entered beyond dist skipped).
// The margin m = 1e-4*dist + 1e-2 is far above the float rounding of the slab test.
enum { kClosest = 0, kWindow = 1, kOccluder = 2 };

struct TraceState {
    double best_t;
    int32_t best_prim;
    uint32_t best_mat;
    bool occluded, found;
    bool dropped;  // a stack entry was lost (retry flavour): the result is void
#ifdef MCPT_TRAVERSAL_STATS
    unsigned nv, nt, iters, maxsp;
#endif
};

// The per-lane traversal stack: STK entries in LDS (column `tid` of stk[][kBlock]).  A ray holds at most one entry per inner ancestor
// (tree height - 1), but the deepest stack any ray of the chess frames reaches is 11-12 entries (SAH trees of height 20-24) or 14-15
// (LBVH, height 27-36; tools/traversal_stats_env.py), while every LDS entry costs 1 KB per workgroup and, beyond 19, resident
// workgroups (up to 19 entries: 8 per CU, 20-22: 7, 23-26: 6, 32: 5, 48: 3).  Three flavours:
//   plain   trees of up to 24 levels: STK >= height - 1 LDS entries, a push can never fail.
//   retry   deeper trees (RETRY): 16 LDS entries; a push onto a full stack drops the entry and marks the ray (`dropped`).  The walk goes on
//           (it only visits less) and its result is thrown away: the ray goes to the kernel's RETRACE LIST (RetryList), and a small
//           kernel launched right behind (k_retrace_closest / k_retrace_shadow / k_primary_retrace) traces the listed rays again with the
//   scratch flavour (STK = 0, SCR): the whole stack is a per-lane array of kMaxBvhHeight entries in scratch memory -- slow and exact.
// The hot kernels carry no second copy of the loop: a retrace inlined behind the first walk was measured first and cost them 6-8 %
// (registers, scratch set-up, instruction cache); so did an overflow array behind the LDS entries inside the loop (a compare per pop).
// Results never depend on the stack size: the checking build (-DMCPT_FORCE_RETRY -DMCPT_STK_RETRY=4) sends most rays of every scene
// through the lists and renders the same frames (tests/test_gpu_checks.py).  Measured, chess frame with the GPU-built tree (27
// levels): 32 LDS entries 3700 Msamples/s, retry flavour 4160 (8 workgroups per CU instead of 5).
#ifndef MCPT_STK_RETRY
#define MCPT_STK_RETRY 16
#endif
constexpr int kStkRetry = MCPT_STK_RETRY;  // LDS entries of the retry flavour (the checking build: 4, and every tree uses it)
// (Macros, not functions: the plain flavour must compile to exactly the statements it had before the other flavours existed.)
#define MCPT_STK_PUSH(v)                                                                                           \
    do {                                                                                                           \
        if (SCR) {                                                                                                 \
            if (sp < kMaxBvhHeight) scr[sp++] = (v); /* never full: mcpt_scene_create refuses deeper trees */      \
        } else {                                                                                                   \
            if (sp < STK) stk[sp++][tid] = (v); /* plain: never full (STK >= height - 1, asserted at creation) */  \
            else if (MARK) st.dropped = true;   /* retry: the entry is lost, the ray is traced again */            \
        }                                                                                                          \
    } while (0)
#define MCPT_STK_POP() (SCR ? scr[--sp] : stk[--sp][tid]) /* sp > 0 */

// Speculative while-while loop (Aila & Laine 2009, "Understanding the efficiency of ray traversal on GPUs").  A plain
// `if (inner) node-step else leaf-test` loop makes a wave pay for BOTH bodies in nearly every iteration (with 64 lanes, some lane
// always holds a leaf).  Here a round has two phases:
//   phase 1  inner nodes only.  A lane that reaches a leaf PARKS it and keeps descending from its stack (speculatively: the
//            parked leaf might have shortened the ray); a lane that reaches a second leaf, or runs out of work, waits.  The phase ends
//            when at most kLeafVote lanes of the wave are still looking for their first leaf.
//   phase 2  every lane tests its parked leaf; a second leaf waiting in `cur` is parked for the next round.
// Measured on the chess frame (A/B on one box, same build otherwise): plain loop 4190 Msamples/s; this loop with vote 0: 4375,
// 4: 4515, 8: 4540, 12: 4540, 16: 4525; testing the second leaf in the same round instead of parking it: 4430.  The serialised
// k_trace_closest went from 82.5 to 72 ms per 2 x 256 spp.  Same tests, same results: the order of primitive tests does not
// matter (ties go to the larger primitive id), and the pruning margins are unchanged.
//
// INST (scenes with instanced objects, csrc/mcpt_scene.cpp): a leaf index >= n_leaf_prims is an instance.  Entering it moves the
// origin used by the BOX tests by -shift, pushes an exit marker and continues in the prototype's shared subtree, whose leaves hold
// local triangle indices; primitive tests always use the world ray and the object's own world-space triangle
// (first_tri + local index), so hits are exactly those of the un-instanced tree.  Popping the marker restores the origin.
constexpr int32_t kNoWork = (int32_t)0x80000000;   // neither an inner node (>= 0) nor a leaf (~index, index < 2^31 - 2)
constexpr int32_t kInstExit = (int32_t)0x80000001; // stack marker: the subtree of the current instance is exhausted
#ifndef MCPT_LEAF_VOTE
#define MCPT_LEAF_VOTE 12
#endif
constexpr int kLeafVote = MCPT_LEAF_VOTE;
// NF, the node format: 0 float boxes (the exact ones: the reference's own box semantics), 1 quantised (QNode), 2 "prepared" -- the
// SMALL kernels' LDS copy of the quantised nodes, converted once per workgroup to floats RELATIVE to the grid origin (x' = q * cell), so that
// a slab bound is ONE fma, x' * inv + (origin - o) * inv, with no integer-to-float conversion per visit (12 of the ~55 vector instructions
// of a visit).  Same grid, same conservative boxes as format 1 (the error of the form is 0.4 % of the one-cell margin).
template <int MODE, int STK, bool SCR, bool MARK, bool FAST, int NF, bool INST>
MCPT_DI void traverse_loop(const DevScene &S, const Ray &r, float dist, int32_t (*stk)[kBlock], int32_t *scr, int tid, TraceState &st) {
    constexpr bool QUANT = NF != 0;
    static_assert(NF != 2 || !INST, "prepared nodes: small scenes, never instanced");
    QRay qr;
    if (QUANT) qr = make_qray(S, r);
    Ray rb = r;               // the ray of the box tests (origin shifted inside an instance)
    int32_t prim_base = 0;    // first triangle of the current instance (0 at the top level: leaf indices are primitive ids)
    const uint32_t n_leaf_prims = INST ? (uint32_t)S.n_leaf_prims : 0x7ffffffeu;
    const float margin = dist * 1e-4f + 1e-2f;
    float lim = (MODE == kClosest) ? INFINITY : (dist + margin);
    const float lo = dist - margin;
    float tm, tx;
    int32_t cur = S.root;
    if (!box_hit<FAST>(S.root_min, S.root_max, r, tm, tx)) return;
    int sp = 0;
    int32_t leaf = kNoWork;
    if (cur < 0 && (uint32_t)(~cur) < n_leaf_prims) {  // the root is a leaf (a scene of one primitive)
        leaf = cur;
        cur = kNoWork;
    }
    while (true) {
        // ---- phase 1: inner nodes
        while (true) {
#ifdef MCPT_TRAVERSAL_STATS
            st.iters++;
#endif
            if (INST) {
                if (cur == kInstExit) {  // back to the top level
                    rb.o = r.o;
                    prim_base = 0;
                    if (QUANT) qr.b = make_qray(S, r).b;
                    cur = (sp == 0) ? kNoWork : MCPT_STK_POP();
                }
                if (cur > kInstExit && cur < 0 && (uint32_t)(~cur) >= n_leaf_prims) {  // an instance: enter its prototype's subtree
                    const InstRec I = S.inst[(uint32_t)(~cur) - n_leaf_prims];
                    rb.o = mk3(r.o.x - I.shift[0], r.o.y - I.shift[1], r.o.z - I.shift[2]);
                    prim_base = I.first_tri;
                    if (QUANT) qr.b = make_qray(S, rb).b;
                    MCPT_STK_PUSH(kInstExit);
                    cur = I.root;
                }
            }
            if (cur >= 0) {
#ifdef MCPT_TRAVERSAL_STATS
                st.nv++;
#endif
                int32_t left, right;
                float tl = 0.f, tr = 0.f, txl = 0.f, txr = 0.f;
                bool hl, hr;
                // every inner node has two children (the builders only emit an inner node for >= 2 primitives)
                if (NF == 2) {  // prepared node (LDS): float boxes relative to the grid origin
                    const float4 *np = S.pnodes + 4 * cur;
                    const float4 a = np[0], b = np[1], c = np[2], e = np[3];
                    left = __float_as_int(e.x);
                    right = __float_as_int(e.y);
                    hl = pbox_hit<FAST>(S, rb, qr, a.x, a.y, a.z, a.w, b.x, b.y, tl, txl);
                    hr = pbox_hit<FAST>(S, rb, qr, b.z, b.w, c.x, c.y, c.z, c.w, tr, txr);
                } else if (QUANT) {  // 32-byte node: two 16-byte requests per lane instead of four
                    const uint4 *np = reinterpret_cast<const uint4 *>(S.qnodes + cur);
                    const uint4 a = np[0], b = np[1];
                    left = (int32_t)b.z;
                    right = (int32_t)b.w;
                    hl = qbox_hit<FAST>(S, rb, qr, a.x & 0xffffu, a.x >> 16, a.y & 0xffffu, a.y >> 16, a.z & 0xffffu, a.z >> 16, tl, txl);
                    hr = qbox_hit<FAST>(S, rb, qr, a.w & 0xffffu, a.w >> 16, b.x & 0xffffu, b.x >> 16, b.y & 0xffffu, b.y >> 16, tr, txr);
                } else {
                    const float4 *np = reinterpret_cast<const float4 *>(S.nodes + cur);
                    const float4 a = np[0], b = np[1], c = np[2], e = np[3];
                    const float lmin[3] = {a.x, a.y, a.z}, lmax[3] = {a.w, b.x, b.y};
                    const float rmin[3] = {b.z, b.w, c.x}, rmax[3] = {c.y, c.z, c.w};
                    left = __float_as_int(e.x);
                    right = __float_as_int(e.y);
                    hl = box_hit<FAST>(lmin, lmax, rb, tl, txl);
                    hr = box_hit<FAST>(rmin, rmax, rb, tr, txr);
                }
                hl = hl && !(tl > lim);
                hr = hr && !(tr > lim);
                if (MODE == kWindow) {
                    hl = hl && !(txl < lo);
                    hr = hr && !(txr < lo);
                }
                if (hl && hr) {
                    const bool swap = tr < tl;
                    const int32_t nearc = swap ? right : left, farc = swap ? left : right;
#ifdef MCPT_TRAVERSAL_STATS
                    const int sp_before = sp;
#endif
                    MCPT_STK_PUSH(farc);
#ifdef MCPT_TRAVERSAL_STATS
                    st.maxsp = max(st.maxsp, sp == sp_before ? 1000u : (unsigned)sp);  // (1000: a dropped entry)
#endif
                    cur = nearc;
                } else if (hl) {
                    cur = left;
                } else if (hr) {
                    cur = right;
                } else {
                    cur = (sp == 0) ? kNoWork : MCPT_STK_POP();
                }
                if (cur < 0 && (uint32_t)(~cur) < n_leaf_prims && leaf == kNoWork) {  // first leaf of the round: park it and keep traversing
                    leaf = ~(prim_base + ~cur);  // (a global primitive id from here on)
                    cur = (sp == 0) ? kNoWork : MCPT_STK_POP();
                }
            }
            // a lane can still make progress on nodes if it holds an inner node (or, INST, an instance / exit marker)
            const bool workable = INST ? (cur >= 0 || (cur > kNoWork && (uint32_t)(~cur) >= n_leaf_prims)) : (cur >= 0);
            if (__popcll(__ballot(leaf == kNoWork && workable)) <= kLeafVote) break;
        }
        // ---- phase 2: the parked leaf
        if (leaf != kNoWork) {
#ifdef MCPT_TRAVERSAL_STATS
            st.nt++;
#endif
            const int32_t prim = ~leaf;
            double t = 0, u, v;
            bool h;
            uint32_t mb;
            if (prim < S.n_tri) {
                const TriGeom g = S.tri_geom[prim];
                mb = g.mat_bits;
                h = tri_hit(g, r, t, u, v);
            } else {
                float ts = 0.f;
                const SphereRec sph = S.spheres[prim - S.n_tri];
                mb = sph.mat_bits;
                h = sphere_hit(sph, r, ts);
                t = (double)ts;
            }
            if (h) {
                if (MODE != kClosest) {
                    const double dd = t - (double)dist;
                    if (dd <= -(double)kEps) {
                        st.occluded = true;
                        return;
                    }
                    if (fabs(dd) < (double)kEps) st.found = true;
                } else if (t < st.best_t || (t == st.best_t && prim > st.best_prim)) {
                    st.best_t = t;
                    st.best_prim = prim;
                    st.best_mat = mb;
                    lim = (float)(t + (fabs(t) * 1e-4 + 1e-2));
                }
            }
            leaf = kNoWork;
            if (cur < 0 && (uint32_t)(~cur) < n_leaf_prims) {  // a second leaf was waiting: park it for the next round
                leaf = ~(prim_base + ~cur);
                cur = (sp == 0) ? kNoWork : MCPT_STK_POP();
            }
        }
        if (cur == kNoWork && leaf == kNoWork) return;
    }
}

// Primitive test shared with the leaf branch of the loop (k_direct tests the sampled light primitive with it).
MCPT_DI bool prim_hit(const DevScene &S, int32_t prim, const Ray &r, double &t) {
    if (prim < S.n_tri) {
        double u, v;
        return tri_hit(S.tri_geom[prim], r, t, u, v);
    }
    float ts = 0.f;
    const bool h = sphere_hit(S.spheres[prim - S.n_tri], r, ts);
    t = (double)ts;
    return h;
}

// The dispatch of one query over the loop's instantiations: slab-test flavour (wave-uniform: the exact NaN-faithful chain only when some
// lane of the wave has a non-finite reciprocal, i.e. a zero direction component; otherwise the bit-identical max3/min3 form), node
// format, instancing.  Shadow queries, Scene.cpp:74-75: a light sample counts iff the CLOSEST hit lies within EPSILON of the light
// distance, i.e. iff some hit lies in the window AND no hit lies at or below dist - EPSILON; `found`: the window search is already
// settled (k_direct found the sampled primitive in the window).
#define MCPT_TL(MODE, FAST, STKN, SCRB, MARKB, SCRP)                                                                  \
    do {                                                                                                              \
        if (PREP && S.pnodes) { /* SMALL kernels of a scene with quantised nodes */                                   \
            traverse_loop<MODE, STKN, SCRB, MARKB, FAST, (PREP ? 2 : 1), false>(S, r, dist, stk, SCRP, tid, st);      \
        } else if (S.inst) {                                                                                          \
            if (S.qnodes) traverse_loop<MODE, STKN, SCRB, MARKB, FAST, 1, true>(S, r, dist, stk, SCRP, tid, st);      \
            else traverse_loop<MODE, STKN, SCRB, MARKB, FAST, 0, true>(S, r, dist, stk, SCRP, tid, st);               \
        } else {                                                                                                      \
            if (S.qnodes) traverse_loop<MODE, STKN, SCRB, MARKB, FAST, 1, false>(S, r, dist, stk, SCRP, tid, st);     \
            else traverse_loop<MODE, STKN, SCRB, MARKB, FAST, 0, false>(S, r, dist, stk, SCRP, tid, st);              \
        }                                                                                                             \
    } while (0)
#define MCPT_QUERY(STKN, SCRB, MARKB, SCRP)                                        \
    do {                                                                           \
        const bool plain = __all(ray_is_plain(r)) != 0;                            \
        if (SHADOW) {                                                              \
            if (plain) {                                                           \
                if (!found) MCPT_TL(kWindow, true, STKN, SCRB, MARKB, SCRP);       \
                if (st.found && !st.occluded) MCPT_TL(kOccluder, true, STKN, SCRB, MARKB, SCRP);  \
            } else {                                                               \
                if (!found) MCPT_TL(kWindow, false, STKN, SCRB, MARKB, SCRP);      \
                if (st.found && !st.occluded) MCPT_TL(kOccluder, false, STKN, SCRB, MARKB, SCRP); \
            }                                                                      \
        } else if (plain) {                                                        \
            MCPT_TL(kClosest, true, STKN, SCRB, MARKB, SCRP);                      \
        } else {                                                                   \
            MCPT_TL(kClosest, false, STKN, SCRB, MARKB, SCRP);                     \
        }                                                                          \
    } while (0)

// The scratch flavour: the ray again, from the start, with the whole stack in a per-lane array (see MCPT_STK_PUSH).
template <bool SHADOW>
MCPT_DI void traverse_again(const DevScene &S, const Ray &r, float dist, int32_t (*stk)[kBlock], int tid, bool found, TraceState &st) {
    constexpr bool PREP = false;
    int32_t scr[kMaxBvhHeight];
    st.best_t = DBL_MAX;
    st.best_prim = -1;
    st.best_mat = 0;
    st.occluded = false;
    st.found = found;
    st.dropped = false;
    MCPT_QUERY(0, true, false, scr);
}

template <bool SHADOW, int STK, bool RETRY, bool PREP = false>
MCPT_DI TraceResult traverse(const DevScene &S, const Ray &r, float dist, int32_t (*stk)[kBlock], int tid, bool found = false) {
    TraceState st;
    st.best_t = DBL_MAX;
    st.best_prim = -1;
    st.best_mat = 0;
    st.occluded = false;
    st.found = found;
    if (RETRY) st.dropped = false;
#ifdef MCPT_TRAVERSAL_STATS
    st.nv = st.nt = st.iters = st.maxsp = 0;
#endif
    MCPT_QUERY(STK, false, RETRY, nullptr);
#ifdef MCPT_TRAVERSAL_STATS
    if (S.dbg) {  // [kind*8 + {rays, node visits, prim tests, occluded/hit, wave-iterations*64, found}]
        const int base = SHADOW ? 8 : 0;
        atomicAdd(&S.dbg[base + 0], 1ull);
        atomicAdd(&S.dbg[base + 1], (unsigned long long)st.nv);
        atomicAdd(&S.dbg[base + 2], (unsigned long long)st.nt);
        atomicAdd(&S.dbg[base + 3], (unsigned long long)(SHADOW ? (st.occluded ? 1 : 0) : (st.best_prim >= 0 ? 1 : 0)));
        unsigned mx = st.iters;
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, o));
        if (lane_id() == 0) atomicAdd(&S.dbg[base + 4], (unsigned long long)mx * 64ull);
        atomicAdd(&S.dbg[base + 5], (unsigned long long)(SHADOW ? (st.found ? 1 : 0) : 0));
        atomicMax(&S.dbg[SHADOW ? 7 : 6], (unsigned long long)st.maxsp);  // deepest stack of any ray (1000: an entry was dropped)
    }
#endif
    return TraceResult{st.best_t, st.best_prim, st.best_mat, !st.occluded && st.found, RETRY ? st.dropped : false};
}

// The retrace of one listed ray (scratch flavour).
template <bool SHADOW>
MCPT_DI TraceResult traverse_scratch(const DevScene &S, const Ray &r, float dist, int32_t (*stk)[kBlock], int tid, bool found = false) {
    TraceState st;
#ifdef MCPT_TRAVERSAL_STATS
    st.nv = st.nt = st.iters = st.maxsp = 0;
#endif
    traverse_again<SHADOW>(S, r, dist, stk, tid, found, st);
    return TraceResult{st.best_t, st.best_prim, st.best_mat, !st.occluded && st.found, false};
}

MCPT_DI void retry_append(const RetryList &rl, uint32_t v) {  // (one atomic per lost ray: there are next to none)
    const uint32_t k = atomicAdd(rl.count, 1u);
    if (k < rl.cap) rl.items[k] = v;
}
// end of a retrace kernel: the last workgroup to finish clears the list for the next launch
MCPT_DI void retry_finish(const RetryList &rl) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        if (atomicAdd(rl.done, 1u) == gridDim.x - 1u) {
            *rl.count = 0u;
            *rl.done = 0u;
        }
    }
}

MCPT_DI uint4 pack_hit(double t, int32_t prim, uint32_t mat_bits) {  // {t lo, t hi, prim, TriGeom::mat_bits}
    const unsigned long long tb = (unsigned long long)__double_as_longlong(t);
    return make_uint4((uint32_t)tb, (uint32_t)(tb >> 32), (uint32_t)prim, mat_bits);
}

template <int STK, bool RETRY, bool SMALL>
__global__ __launch_bounds__(kBlock) void k_trace_closest(DevScene S, uint32_t n_host, const uint32_t *__restrict__ n_dev,
                                                          const float4 *__restrict__ ray_o, const float4 *__restrict__ ray_d,
                                                          uint4 *__restrict__ hit, RetryList rl) {
    __shared__ int32_t stk[STK][kBlock];
    if constexpr (SMALL) {
        __shared__ SmallGeomLds small_geom;
        stage_small_geom(S, small_geom);
        __syncthreads();
    }
    const int tid = threadIdx.x;
    const uint32_t n = n_dev ? *n_dev : n_host;
    // grid-stride: when the length is only known on the device the host sizes the grid from an estimate
    for (uint32_t i = blockIdx.x * kBlock + tid; i < n; i += gridDim.x * kBlock) {
        const Ray r = make_ray(ld3(ray_o[i]), ld3(ray_d[i]));
        const TraceResult tr = traverse<false, STK, RETRY, SMALL>(S, r, 0.f, stk, tid);
        if (RETRY && tr.dropped) retry_append(rl, i);
        else hit[i] = pack_hit(tr.t, tr.prim, tr.mat_bits);
    }
}

// The rays k_trace_closest / k_trace_closest_refill put on their retrace list, with the scratch stack.
__global__ __launch_bounds__(kBlock) void k_retrace_closest(DevScene S, const float4 *__restrict__ ray_o, const float4 *__restrict__ ray_d,
                                                            uint4 *__restrict__ hit, RetryList rl) {
    __shared__ int32_t stk[1][kBlock];
    const uint32_t n = min(*rl.count, rl.cap);
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const uint32_t i = rl.items[k];
        const Ray r = make_ray(ld3(ray_o[i]), ld3(ray_d[i]));
        const TraceResult tr = traverse_scratch<false>(S, r, 0.f, stk, threadIdx.x);
        hit[i] = pack_hit(tr.t, tr.prim, tr.mat_bits);
    }
    retry_finish(rl);
}

// k_trace_closest with lane refill.  Rays of one wave finish at different times (59 % lane utilisation with one ray per lane,
// profiles/r02_traversal_stats.txt).  Here a wave owns a chunk of kRaysPerLane x 64 consecutive rays and, at the top of every round
// of the speculative loop, lanes whose ray has finished store its hit and -- once at least kRefillMin lanes are idle -- take the next
// rays of the chunk (ballot + prefix count: no atomics, no persistent grid: the grid is still one workgroup per 1024 rays).
// The common configuration only (quantised nodes, no instancing); the rays of the chunk that have a zero direction component are traced
// by the generic path before the loop starts (round 3: tracing them inside a refill, with the rest of the wave parked in the middle of
// its own walks, lost the rest of the chunk when a refill handed out nothing but such rays -- tests/test_gpu_small_scene.py has the
// case).  Same tests per ray as traverse_loop<kClosest, STK, true, true, false>, hence the same hits.
#ifndef MCPT_REFILL_MIN
#define MCPT_REFILL_MIN 16
#endif
#ifndef MCPT_RAYS_PER_LANE
#define MCPT_RAYS_PER_LANE 4
#endif
constexpr uint32_t kRaysPerLane = MCPT_RAYS_PER_LANE, kRefillMin = MCPT_REFILL_MIN;
// Occupancy: the refill state costs registers.  Round 2: 79 VGPRs unconstrained (6 waves per SIMD); bounded to 7 waves (72 VGPRs, 14 spilled)
// was the optimum then: frame 4617 (one ray per lane) -> 4645 (unconstrained) -> 4695 (7 waves) -> 4620 (8 waves, 42 spills).  Round 3 moved the
// generic walk of zero-component rays out of the loop, which freed registers: 69 VGPRs unconstrained, 64 with 4 spills at 8 waves -- frame
// +0.6..1.1 % over 7 waves (A/B on one box: 4847 -> 4888); 8 rays per lane instead of 4: +0.5 %, refill threshold 8 instead of 16: +0.1 %.
// (Stacks deeper than 20 entries: LDS bounds the occupancy below 7 anyway, so no register bound there.)
#ifndef MCPT_REFILL_WAVES
#define MCPT_REFILL_WAVES 8
#endif
template <int STK, bool RETRY, bool SMALL>
__global__ __launch_bounds__(kBlock, (STK <= 20 ? MCPT_REFILL_WAVES : 1)) void k_trace_closest_refill(DevScene S, uint32_t n_host, const uint32_t *__restrict__ n_dev,
                                                                 const float4 *__restrict__ ray_o, const float4 *__restrict__ ray_d,
                                                                 uint4 *__restrict__ hit, RetryList rl) {
    __shared__ int32_t stk[STK][kBlock];
    if constexpr (SMALL) {
        __shared__ SmallGeomLds small_geom;
        stage_small_geom(S, small_geom);
        __syncthreads();
    }
    const int tid = threadIdx.x;
    const uint32_t wave = (uint32_t)tid >> 6;
    const uint32_t n = n_dev ? *n_dev : n_host;
    constexpr uint32_t kChunk = 64u * kRaysPerLane;
    for (uint32_t chunk = blockIdx.x * (kBlock / 64) + wave; (unsigned long long)chunk * kChunk < n; chunk += gridDim.x * (kBlock / 64)) {
        uint32_t next = chunk * kChunk;
        const uint32_t end = min(n, next + kChunk);
        int32_t my = -1;  // index of the ray this lane is tracing
        Ray r = make_ray(mk3(0, 0, 0), mk3(0, 0, 1));
        QRay qr = make_qray(S, r);
        float lim = INFINITY;
        double best_t = DBL_MAX;
        int32_t best_prim = -1;
        uint32_t best_mat = 0;
        int32_t cur = kNoWork, leaf = kNoWork;
        int sp = 0;
        bool dropped = false;  // RETRY: this ray lost a stack entry (see MCPT_STK_PUSH)
        // Rays with a zero direction component (non-finite reciprocals: rare) need the NaN-faithful slab chain, which the loop below
        // does not carry.  They are traced first, by the generic path, with the whole wave at one place (a wave without such a ray
        // pays for four direction loads per lane, which the refills below then find in the cache); the refill skips them.
        for (uint32_t k = 0; k < kRaysPerLane; ++k) {
            const uint32_t idx = next + k * 64u + lane_id();
            bool odd = false;
            Ray ro = r;
            if (idx < end) {
                ro = make_ray(ld3(ray_o[idx]), ld3(ray_d[idx]));
                odd = !ray_is_plain(ro);
            }
            if (__any(odd)) {
                if (odd) {
                    const TraceResult tr = traverse<false, STK, RETRY, SMALL>(S, ro, 0.f, stk, tid);
                    if (RETRY && tr.dropped) retry_append(rl, idx);
                    else hit[idx] = pack_hit(tr.t, tr.prim, tr.mat_bits);
                }
            }
        }
        while (true) {
            // ---- finished lanes: store, refill
            const bool idle = cur == kNoWork && leaf == kNoWork;
            if (idle && my >= 0) {
                if (RETRY && dropped) retry_append(rl, (uint32_t)my);  // (k_retrace_closest traces it again)
                else hit[my] = pack_hit(best_t, best_prim, best_mat);
                my = -1;
            }
            const unsigned long long im = __ballot(idle);
            const uint32_t n_idle = (uint32_t)__popcll(im);
            if (next < end && n_idle >= kRefillMin) {
                const uint32_t idx = next + __builtin_amdgcn_mbcnt_hi((uint32_t)(im >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)im, 0u));
                if (idle && idx < end) {
                    r = make_ray(ld3(ray_o[idx]), ld3(ray_d[idx]));
                    if (ray_is_plain(r)) {  // (the others were traced before the loop: the lane stays idle and is refilled again)
                        my = (int32_t)idx;
                        qr = make_qray(S, r);
                        lim = INFINITY;
                        best_t = DBL_MAX;
                        best_prim = -1;
                        best_mat = 0;
                        sp = 0;
                        dropped = false;
                        float tm, tx;
                        if (box_hit<true>(S.root_min, S.root_max, r, tm, tx)) {
                            if (S.root >= 0) cur = S.root;
                            else leaf = S.root;  // a scene of one primitive
                        }
                    }
                }
                next += n_idle;
            }
            if (!__any(my >= 0)) {
                if (next < end) continue;  // every ray just handed out had been traced before the loop: hand out the next ones
                break;
            }
            // ---- phase 1: inner nodes (see traverse_loop)
            while (true) {
                if (cur >= 0) {
                    int32_t left, right;
                    float tl = 0.f, tr = 0.f, txl = 0.f, txr = 0.f;
                    bool hl, hr;
                    if constexpr (SMALL) {  // prepared nodes in LDS (stage_small_geom)
                        const float4 *np = S.pnodes + 4 * cur;
                        const float4 a = np[0], b = np[1], c = np[2], e = np[3];
                        left = __float_as_int(e.x);
                        right = __float_as_int(e.y);
                        hl = pbox_hit<true>(S, r, qr, a.x, a.y, a.z, a.w, b.x, b.y, tl, txl);
                        hr = pbox_hit<true>(S, r, qr, b.z, b.w, c.x, c.y, c.z, c.w, tr, txr);
                    } else {
                        const uint4 *np = reinterpret_cast<const uint4 *>(S.qnodes + cur);
                        const uint4 a = np[0], b = np[1];
                        left = (int32_t)b.z;
                        right = (int32_t)b.w;
                        hl = qbox_hit<true>(S, r, qr, a.x & 0xffffu, a.x >> 16, a.y & 0xffffu, a.y >> 16, a.z & 0xffffu, a.z >> 16, tl, txl);
                        hr = qbox_hit<true>(S, r, qr, a.w & 0xffffu, a.w >> 16, b.x & 0xffffu, b.x >> 16, b.y & 0xffffu, b.y >> 16, tr, txr);
                    }
                    hl = hl && !(tl > lim);
                    hr = hr && !(tr > lim);
                    if (hl && hr) {
                        const bool swap = tr < tl;
                        const int32_t nearc = swap ? right : left, farc = swap ? left : right;
                        if (sp < STK) stk[sp++][tid] = farc;
                        else if (RETRY) dropped = true;
                        cur = nearc;
                    } else if (hl) {
                        cur = left;
                    } else if (hr) {
                        cur = right;
                    } else {
                        cur = (sp == 0) ? kNoWork : stk[--sp][tid];
                    }
                    if (cur < 0 && cur != kNoWork && leaf == kNoWork) {
                        leaf = cur;
                        cur = (sp == 0) ? kNoWork : stk[--sp][tid];
                    }
                }
                if (__popcll(__ballot(leaf == kNoWork && cur >= 0)) <= kLeafVote) break;
            }
            // ---- phase 2: the parked leaf
            if (leaf != kNoWork) {
                const int32_t prim = ~leaf;
                double t = 0, u, v;
                bool h;
                uint32_t mb;
                if (prim < S.n_tri) {
                    const TriGeom g = S.tri_geom[prim];
                    mb = g.mat_bits;
                    h = tri_hit(g, r, t, u, v);
                } else {
                    float ts = 0.f;
                    const SphereRec sph = S.spheres[prim - S.n_tri];
                    mb = sph.mat_bits;
                    h = sphere_hit(sph, r, ts);
                    t = (double)ts;
                }
                if (h && (t < best_t || (t == best_t && prim > best_prim))) {
                    best_t = t;
                    best_prim = prim;
                    best_mat = mb;
                    lim = (float)(t + (fabs(t) * 1e-4 + 1e-2));
                }
                leaf = kNoWork;
                if (cur < 0 && cur != kNoWork) {
                    leaf = cur;
                    cur = (sp == 0) ? kNoWork : stk[--sp][tid];
                }
            }
        }
    }
}

// Queue position -> entry of the sharded shadow queue (Counters): positions [0, pf[K]) are the entries whose light sample was found, shard
// by shard; the next pw[K] positions the others.  pf / pw: exclusive prefix sums of the shards' counts (K + 1 values each, in LDS).
MCPT_DI void shadow_prefix(const Counters *counters, int next_idx, uint32_t *pf, uint32_t *pw) {  // all threads of the workgroup call it
    if (threadIdx.x < 64) {
        const uint32_t t = threadIdx.x;
        uint32_t cf = t < kShadowShards ? counters->n_shadow[next_idx][t].v : 0u;
        uint32_t cw = t < kShadowShards ? counters->n_shadow_w[next_idx][t].v : 0u;
        uint32_t sf = cf, sw = cw;
        for (int o = 1; o < 64; o <<= 1) {  // inclusive scan over the wave
            const uint32_t a = (uint32_t)__shfl_up((int)sf, o), b = (uint32_t)__shfl_up((int)sw, o);
            if ((int)t >= o) {
                sf += a;
                sw += b;
            }
        }
        if (t < kShadowShards) {
            pf[t] = sf - cf;
            pw[t] = sw - cw;
        }
        if (t == kShadowShards - 1u) {
            pf[kShadowShards] = sf;
            pw[kShadowShards] = sw;
        }
    }
    __syncthreads();
}
MCPT_DI uint32_t shadow_entry(const uint32_t *pf, const uint32_t *pw, uint32_t i, uint32_t region, bool &found) {
    const uint32_t nf = pf[kShadowShards];
    found = i < nf;
    const uint32_t *p = found ? pf : pw;
    const uint32_t j = found ? i : i - nf;
    uint32_t s = 0;  // the largest shard index with p[s] <= j (an empty shard shares its prefix with the next one: the later one wins)
    for (uint32_t step = kShadowShards / 2; step; step >>= 1)
        if (p[s + step] <= j) s += step;
    const uint32_t off = j - p[s];
    return found ? s * region + off : (s + 1u) * region - 1u - off;
}

// Shadow queue consumer: a fixed grid strides over the queue, whose length is only known on the device.
// Items [0, n_found) are the front of the arrays (light sample already found: occluder search only), the next n_window
// items are read from the back (window search first), so that the long occluder searches fill whole waves.
template <int STK, bool RETRY, bool SMALL>
__global__ __launch_bounds__(kBlock) void k_trace_shadow(DevScene S, const Counters *__restrict__ counters, int next_idx, uint32_t cap,
                                                         const float4 *__restrict__ shq_o, const float4 *__restrict__ shq_d,
                                                         float *__restrict__ contrib, RetryList rl) {
    __shared__ int32_t stk[STK][kBlock];
    __shared__ uint32_t pf[kShadowShards + 1], pw[kShadowShards + 1];
    const int tid = threadIdx.x;
    if constexpr (SMALL) {  // (made visible by the barrier at the end of shadow_prefix)
        __shared__ SmallGeomLds small_geom;
        stage_small_geom(S, small_geom);
    }
    shadow_prefix(counters, next_idx, pf, pw);
    const uint32_t n = pf[kShadowShards] + pw[kShadowShards];
    const uint32_t region = shadow_region(cap);
    for (uint32_t i = blockIdx.x * kBlock + tid; i < n; i += gridDim.x * kBlock) {
        bool found;
        const uint32_t e = shadow_entry(pf, pw, i, region, found);
        const float4 o = shq_o[e], d = shq_d[e];
        const Ray r = make_ray(ld3(o), ld3(d));
        const TraceResult tr = traverse<true, STK, RETRY, SMALL>(S, r, d.w, stk, tid, found);
        if (RETRY && tr.dropped) retry_append(rl, i);  // (undecided: k_retrace_shadow)
        else if (!tr.visible) contrib[__float_as_uint(o.w)] = 0.f;  // Scene.cpp:74-79: an invisible sample adds nothing
    }
}

// The shadow rays k_trace_shadow put on its retrace list (queue positions), with the scratch stack.
__global__ __launch_bounds__(kBlock) void k_retrace_shadow(DevScene S, const Counters *__restrict__ counters, int next_idx, uint32_t cap,
                                                           const float4 *__restrict__ shq_o, const float4 *__restrict__ shq_d,
                                                           float *__restrict__ contrib, RetryList rl) {
    __shared__ int32_t stk[1][kBlock];
    __shared__ uint32_t pf[kShadowShards + 1], pw[kShadowShards + 1];
    shadow_prefix(counters, next_idx, pf, pw);
    const uint32_t region = shadow_region(cap);
    const uint32_t n = min(*rl.count, rl.cap);
    for (uint32_t k = blockIdx.x * kBlock + threadIdx.x; k < n; k += gridDim.x * kBlock) {
        const uint32_t i = rl.items[k];
        bool found;
        const uint32_t e = shadow_entry(pf, pw, i, region, found);
        const float4 o = shq_o[e], d = shq_d[e];
        const Ray r = make_ray(ld3(o), ld3(d));
        const TraceResult tr = traverse_scratch<true>(S, r, d.w, stk, threadIdx.x, found);
        if (!tr.visible) contrib[__float_as_uint(o.w)] = 0.f;
    }
    retry_finish(rl);
}

// ------------------------------------------------------------------------------------------------
// Path keys
// ------------------------------------------------------------------------------------------------
// s -> (position in the pixel list, sample index within the pass): s / s_pass and s % s_pass, by shift and mask when s_pass is a
// power of two (a uniform branch; a 32-bit division by a run-time value costs ~25 vector instructions per lane)
MCPT_DI void split_sample(const RenderConst &C, int q, uint32_t s, uint32_t &pl, uint32_t &k) {
    const uint32_t sp = (uint32_t)(q ? C.s_pass[1] : C.s_pass[0]);
    const int sh = q ? C.s_pass_shift[1] : C.s_pass_shift[0];
    if (sh >= 0) {
        pl = s >> sh;
        k = s & (sp - 1u);
    } else {
        pl = s / sp;
        k = s - pl * sp;
    }
}

MCPT_DI void path_key(const RenderConst &C, uint32_t pid, int q, RngKey &key, int &ch) {
    if (C.mode == 0) {
        const uint32_t s = pid / 3u;
        ch = (int)(pid - 3u * s);
        uint32_t pl, k;
        split_sample(C, q, s, pl, k);
        key.pixel = C.pixel_list ? C.pixel_list[pl] : pl;
        key.sample = (uint32_t)(q ? C.sample_offset[1] : C.sample_offset[0]) + k;
    } else {
        ch = C.key_channel[pid];
        key.pixel = C.key_pixel[pid];
        key.sample = C.key_sample[pid];
    }
    key.seed = C.seed;
    key.stream = (uint32_t)ch;
}

// ------------------------------------------------------------------------------------------------
// Camera rays, Renderer.cpp:44-76
// ------------------------------------------------------------------------------------------------
MCPT_DI f3 mat3_mul(const float *M, f3 v) {
    return mk3(M[0] * v.x + (M[1] * v.y + M[2] * v.z), M[3] * v.x + (M[4] * v.y + M[5] * v.z), M[6] * v.x + (M[7] * v.y + M[8] * v.z));
}

MCPT_DI void camera_ray(const CameraConst &cam, uint32_t seed, uint32_t m, uint32_t k, f3 &pos, f3 &dir) {
    const int i = (int)(m % (uint32_t)cam.width), j = (int)(m / (uint32_t)cam.width);
    RngKey rk{seed, m, k, 3u};
    float u[4];
    rng_block(rk, 0u, 0u, u);
    const f3 eye = mk3(cam.eye[0], cam.eye[1], cam.eye[2]);
    const float x = (1 - 2 * (i + u[0]) / (float)cam.width) * cam.aspect * cam.scale;
    const float y = (1 - 2 * (j + u[1]) / (float)cam.height) * cam.scale;
    if (cam.use_dof) {
        const f3 focal_point = mk3(x, y, 1) * cam.focal_distance;
        const float r = cam.aperture_radius * sqrtf(u[2]);
        const float theta = 2 * kPi * u[3];
        float st, ct;
        mcpt_sincosf(theta, &st, &ct);  // Renderer.cpp:59-60
        const float dx = r * ct;
        const float dy = r * st;
        pos = eye + mat3_mul(cam.orient, mk3(dx, dy, 0));
        dir = normalized(focal_point - mk3(dx, dy, 0));
    } else {
        dir = normalized(mk3(x, y, 1));
        pos = eye;
    }
    dir = mat3_mul(cam.orient, dir);  // Renderer.cpp:76
}

// The camera ray of sample s and its closest hit among the pixel's candidate primitives, when it has a short list (csrc/mcpt_cull.hip);
// returns false when the ray has to walk the tree.
MCPT_DI bool primary_ray(const DevScene &S, const CameraConst &cam, const RenderConst &C, int q, uint32_t s, f3 &pos, f3 &dir, TraceResult &tr) {
    uint32_t pl, ks;
    split_sample(C, q, s, pl, ks);
    const uint32_t m = C.pixel_list ? C.pixel_list[pl] : pl;
    const uint32_t k = (uint32_t)(q ? C.sample_offset[1] : C.sample_offset[0]) + ks;
    camera_ray(cam, C.seed, m, k, pos, dir);
    const int4 cd = C.pixel_cand ? C.pixel_cand[pl] : make_int4(kCandTraverse, 0, 0, 0);
    if (cd.x == kCandTraverse) return false;
    // every primitive a ray of this pixel can hit is in the list: test those, with the traversal's tie rule
    const Ray r = make_ray(pos, dir);
    const int32_t cs[4] = {cd.x, cd.y, cd.z, cd.w};
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        if (cs[q4] == kCandNone) continue;
        const int32_t prim = ~cs[q4];
        double t = 0;
        if (prim_hit(S, prim, r, t) && (t < tr.t || (t == tr.t && prim > tr.prim))) {
            tr.t = t;
            tr.prim = prim;
            tr.mat_bits = prim < S.n_tri ? S.tri_geom[prim].mat_bits : S.spheres[prim - S.n_tri].mat_bits;
        }
    }
    return true;
}

// What follows the closest hit of a new sample (all lanes of the workgroup call it; `valid` lanes hold a sample):
//   miss (Scene.cpp:88-95) ............ result[3 channels] = environment, no records
//   depth-0 emitter (Scene.cpp:102-107)  result[3 channels] = clamp(0,1, emission * |wo.n|), no records
//   surface ........................... one ray + hit entry, three fresh path records, three clamp-stack slots
MCPT_DI void primary_finish(const DevScene &S, const RenderConst &C, const Wave &next, int next_idx, int q, uint32_t s, bool valid, f3 pos, f3 dir,
                            const TraceResult &tr, BlockAllocShared &sh) {
    bool surface = false;
    float *const result = q ? C.result[1] : C.result[0];
    if (valid) {
        if (tr.prim < 0) {
            const f3 env = sample_env(S, dir);
            result[(size_t)s * 3 + 0] = env.x;
            result[(size_t)s * 3 + 1] = env.y;
            result[(size_t)s * 3 + 2] = env.z;
        } else {
            const int mat = (int)(tr.mat_bits & kMatIndexMask);
            if (tr.mat_bits >> 31) {  // depth-0 emitter, Scene.cpp:102-107
                const MaterialRec &M = S.mats[mat];
                f3 n;
                if (tr.prim < S.n_tri) {
                    const TriShade ts = S.tri_shade[tr.prim];
                    n = mk3(ts.n[0], ts.n[1], ts.n[2]);
                } else {
                    const SphereRec sp = S.spheres[tr.prim - S.n_tri];
                    const f3 p = pos + dir * (float)tr.t;
                    n = normalized(p - mk3(sp.c[0], sp.c[1], sp.c[2]));
                }
                const float c = fabsf(dot(-dir, n));
                result[(size_t)s * 3 + 0] = clampf(0, 1, M.emit[0] * c);
                result[(size_t)s * 3 + 1] = clampf(0, 1, M.emit[1] * c);
                result[(size_t)s * 3 + 2] = clampf(0, 1, M.emit[2] * c);
            } else {
                surface = true;
            }
        }
    }
    const bool want[3] = {surface, surface, surface && C.track_live};
    const uint32_t mult[3] = {1u, 3u, 3u};
    uint32_t *const ctr[3] = {&C.counters->n_prays[next_idx].v, &C.counters->free_head.v, &C.counters->live[q].v};  // three more unfinished paths of this pass
    const bool sub[3] = {false, false, false};
    uint32_t idx[3];
    block_alloc<3>(sh, want, mult, ctr, sub, idx);
    if (!surface) return;
    // the ray is already traced: it goes to the back of the ray arrays, away from the continuation rays that k_shade appends
    // at the front for k_trace_closest.  The three channel paths get no records: k_shade derives them from the sample entry.
    const uint32_t k = idx[0], ri = C.ray_cap - 1u - k;
    next.ray_o[ri] = make_float4(pos.x, pos.y, pos.z, 0.f);
    next.ray_d[ri] = make_float4(dir.x, dir.y, dir.z, 0.f);
    next.hit[ri] = pack_hit(tr.t, tr.prim, tr.mat_bits);
    next.fresh[k] = make_uint2(s | (q ? 0x80000000u : 0u), idx[1]);
}

// k_primary: Renderer.cpp:44-79 up to the first Scene::intersect of castRay (Scene.cpp:87) for new samples.
// The three channel paths of a sample share the primary ray, so it is generated and traced once.
template <int STK, bool RETRY, bool SMALL>
__global__ __launch_bounds__(kBlock) void k_primary(DevScene S, CameraConst cam, RenderConst C, Wave next, int next_idx, int q,
                                                    uint32_t first_sample, uint32_t n_samples, RetryList rl) {
    __shared__ int32_t stk[STK][kBlock];
    __shared__ BlockAllocShared sh;
    if constexpr (SMALL) {
        __shared__ SmallGeomLds small_geom;
        stage_small_geom(S, small_geom);
        __syncthreads();
    }
    const int tid = threadIdx.x;
    const uint32_t j = blockIdx.x * kBlock + tid;
    bool valid = j < n_samples;
    const uint32_t s = first_sample + j;
    f3 pos = mk3(0, 0, 0), dir = mk3(0, 0, 1);
    TraceResult tr{DBL_MAX, -1, 0u, false, false};
    if (valid) {
        if (!primary_ray(S, cam, C, q, s, pos, dir, tr)) {
            tr = traverse<false, STK, RETRY, SMALL>(S, make_ray(pos, dir), 0.f, stk, tid);
            if (RETRY && tr.dropped) {  // the sample is finished by k_primary_retrace
                retry_append(rl, s);
                valid = false;
            }
        }
    }
#ifdef MCPT_TRAVERSAL_STATS
    if (S.dbg) {  // how many DIFFERENT primitives the 64 primary rays of a wave hit: what a wave-shared (packet) walk would have to visit at least
        const int myp = (valid && tr.prim >= 0) ? tr.prim : -2;
        bool first = myp >= 0;
        for (int l = 0; l < 64; ++l) {
            const int op = __shfl(myp, l);
            if ((uint32_t)l < lane_id() && op == myp) first = false;
        }
        const unsigned long long m = __ballot(first), mh = __ballot(myp >= 0);
        if (lane_id() == 0 && mh) {
            atomicAdd(&S.dbg[14], (unsigned long long)__popcll(m));
            atomicAdd(&S.dbg[15], 1ull);
        }
    }
#endif
    primary_finish(S, C, next, next_idx, q, s, valid, pos, dir, tr, sh);
}

// The samples k_primary put on its retrace list: the same, with the scratch stack (a fixed grid strides over the list).
__global__ __launch_bounds__(kBlock) void k_primary_retrace(DevScene S, CameraConst cam, RenderConst C, Wave next, int next_idx, int q, RetryList rl) {
    __shared__ int32_t stk[1][kBlock];
    __shared__ BlockAllocShared sh;
    const uint32_t n = min(*rl.count, rl.cap);
    for (uint32_t base = blockIdx.x * kBlock; base < n; base += gridDim.x * kBlock) {  // uniform trip count per workgroup
        const uint32_t j = base + threadIdx.x;
        const bool valid = j < n;
        const uint32_t s = valid ? rl.items[j] : 0u;
        f3 pos = mk3(0, 0, 0), dir = mk3(0, 0, 1);
        TraceResult tr{DBL_MAX, -1, 0u, false, false};
        if (valid && !primary_ray(S, cam, C, q, s, pos, dir, tr)) tr = traverse_scratch<false>(S, make_ray(pos, dir), 0.f, stk, threadIdx.x);
        primary_finish(S, C, next, next_idx, q, s, valid, pos, dir, tr, sh);
        __syncthreads();  // `sh` is reused by the next round
    }
    retry_finish(rl);
}

// mcpt_cast_rays: caller-supplied rays, one fresh record per ray; the rays are traced by k_trace_closest.
__global__ __launch_bounds__(kBlock) void k_generate_explicit(Wave next, Counters *c, int next_idx, uint32_t n) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j == 0) {
        c->n_paths[next_idx].v = n;
        c->n_rays[next_idx].v = n;
        c->free_head.v += n;  // the first n ring entries hold slots 0 .. n-1 (record j owns slot j)
    }
    if (j >= n) return;
    next.rec0[j] = make_uint4(j, j, kFresh, j);
}

__global__ __launch_bounds__(kBlock) void k_camera_rays(CameraConst cam, uint32_t seed, uint32_t n, const uint32_t *pixel,
                                                        const uint32_t *sample, float4 *o, float4 *d) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    f3 pos, dir;
    camera_ray(cam, seed, pixel[j], sample[j], pos, dir);
    o[j] = make_float4(pos.x, pos.y, pos.z, 0.f);
    d[j] = make_float4(dir.x, dir.y, dir.z, 0.f);
}

// `start`: initial value of the ring's head counter (0; a test hook starts it just below 2^32 to exercise the wrap).
__global__ __launch_bounds__(kBlock) void k_init_free(uint32_t *free_slots, Counters *c, uint32_t pool, uint32_t start, uint32_t mask) {
    const uint32_t j = blockIdx.x * kBlock + threadIdx.x;
    if (j < pool) free_slots[(start + j) & mask] = j;
    if (j == 0) {
        c->n_paths[0].v = c->n_paths[1].v = 0;
        c->n_rays[0].v = c->n_rays[1].v = 0;
        c->free_head.v = start;
        c->free_tail.v = start + pool;
        c->n_prays[0].v = c->n_prays[1].v = 0;
        c->live[0].v = c->live[1].v = 0;
        for (uint32_t k = 0; k < kShadowShards; ++k) {
            c->n_shadow[0][k].v = c->n_shadow[1][k].v = 0;
            c->n_shadow_w[0][k].v = c->n_shadow_w[1][k].v = 0;
        }
        c->n_direct[0].v = c->n_direct[1].v = 0;
        c->pushes.v = 0;
        c->overflow.v = 0;
        c->ended.v = 0;
        c->last_shadow = 0;
        c->tot_shaded = c->tot_direct = c->tot_shadow = c->tot_cont = c->tot_iterations = c->tot_pushes = c->tot_ended = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// Light sampling: Scene::sampleLight (Scene.cpp:23-37) -> MeshTriangle::Sample (Triangle.hpp:193-196)
// -> BVHAccel::Sample/getSample (BVH.cpp:118-135) -> Triangle::Sample (Triangle.hpp:71-76).
// u = {light choice, triangle pick, x, y}.  Returns false when no light was selected.
// ------------------------------------------------------------------------------------------------
MCPT_DI bool sample_light(const DevScene &S, const float u[4], f3 &x_l, f3 &n_l, f3 &emit, float &pdf, int32_t &prim) {
    float area_sum = S.light_area_sum;  // Scene.cpp:24-27: the sum over the emitters in insertion order (computed once, on the host)
    const float p = u[0] * area_sum;
    area_sum = 0.f;
    for (int k = 0; k < S.n_lights; ++k) {
        const LightRec L = S.lights[k];
        area_sum += L.area;
        if (p <= area_sum) {
            const MaterialRec &m = S.mats[L.mat];
            if (L.kind == MCPT_OBJ_MESH) {
                float pp = sqrtf(u[1]) * L.root_area;  // BVH.cpp:132 (the biased pick is reproduced)
                int32_t ref = L.root;
                while (ref >= 0) {  // BVH.cpp:118-129
                    const LightNode N = S.light_nodes[ref];
                    if (pp < N.left_area) {
                        ref = N.left;
                    } else {
                        pp = pp - N.left_area;
                        ref = N.right;
                    }
                }
                const LightTri T = S.light_tris[~ref];
                const float x = sqrtf(u[2]), y = u[3];  // Triangle.hpp:72
                const f3 v0 = mk3(T.v0[0], T.v0[1], T.v0[2]), v1 = mk3(T.v1[0], T.v1[1], T.v1[2]), v2 = mk3(T.v2[0], T.v2[1], T.v2[2]);
                x_l = (v0 * (1.0f - x) + v1 * (x * (1.0f - y))) + v2 * (x * y);
                n_l = mk3(T.n[0], T.n[1], T.n[2]);
                prim = T.prim;
                pdf = 1.0f / T.area;   // Triangle.hpp:75
                pdf *= T.area;         // BVH.cpp:121
                pdf /= L.root_area;    // BVH.cpp:134
                emit = mk3(m.emit[0], m.emit[1], m.emit[2]);  // Triangle.hpp:195
            } else {  // Sphere::Sample, Sphere.hpp:64-74 (leaves pos.emit untouched: zero here)
                const SphereRec s = S.spheres[L.root];
                const float theta = (float)(2.0 * (double)kPi * (double)u[2]), phi = kPi * u[3];
                float sph, cph, sth, cth;
                mcpt_sincosf(phi, &sph, &cph);
                mcpt_sincosf(theta, &sth, &cth);
                const f3 dir = mk3(cph, sph * cth, sph * sth);
                x_l = mk3(s.c[0], s.c[1], s.c[2]) + dir * s.radius;
                n_l = dir;
                prim = S.n_tri + L.root;
                pdf = 1.0f / L.area;
                emit = mk3(0.f, 0.f, 0.f);
            }
            return true;
        }
    }
    return false;
}

// ------------------------------------------------------------------------------------------------
// direct_is_zero: true only if EVERY light sample at this vertex has eval() == 0 exactly, so that
// Scene::directLighting (Scene.cpp:56-82) returns +0 whatever the draws and the visibilities are.
//   * no emitter in the scene;
//   * a conductor seen from inside: directLighting is called with isReflect = false (Scene.cpp:115-116) and
//     eval(.., false) returns 0 for both conductor types (Material.hpp:355-357,396-399);
//   * a Dirac BSDF (Material.hpp:375-403) is non-zero only if h.N >= 1 - EPSILON, i.e. the half vector lies within
//     acos(1 - 1e-4) = 0.01414 rad of N.  Reflection: ws then lies within 2 * 0.01414 rad of the mirror direction
//     r = 2 (N.wo) N - wo.  Refraction (dielectric seen from inside, ws outside): ws = -ior*wo - lambda*h, so it
//     lies within |lambda| * 0.01414 / cos(theta_i) <= 0.094 rad of the Snell direction when ior * sin(theta_o) <= 0.9.
//     Every light sample lies inside the cone of half-angle asin(R / D) around the direction to the centre of the
//     emitters' bounding sphere (radius R, distance D > 1.01 R).  If r is farther from that cone than the tolerance
//     (0.06 rad for reflection, 0.15 rad for refraction: > 1.5x the bounds above) no sample can give a non-zero eval.
// tests: the -DMCPT_CHECK_DIRECT_SKIP build evaluates the skipped vertices anyway and counts non-zero contributions.
// ------------------------------------------------------------------------------------------------
MCPT_DI bool direct_is_zero(const DevScene &S, const MaterialRec &m, f3 q, f3 n, f3 wo, bool inside, int ch) {
    if (S.n_lights == 0) return true;
    const bool conductor = (m.type == MCPT_SMOOTH_CONDUCTOR || m.type == MCPT_ROUGH_CONDUCTOR);
    if (inside && conductor) return true;
    if (!m.isDirac) return false;
    const f3 L = mk3(S.light_center[0], S.light_center[1], S.light_center[2]) - q;
    const float D2 = dot(L, L), R = S.light_radius;
    if (!(D2 > R * R * 1.0201f)) return false;
    const float D = sqrtf(D2);
    const float sl = R / D, cl = sqrtf(1.0f - sl * sl);
    f3 r;
    float cm, sm;
    if (!inside) {
        r = n * (2 * dot(n, wo)) - wo;
        cm = 0.99820054f;  // cos(0.06)
        sm = 0.05996400f;  // sin(0.06)
    } else {
        const float ior = get_ior(m, ch);
        const float won = dot(wo, n);
        const f3 wot = wo - n * won;
        const float sin2 = ior * ior * dot(wot, wot);
        if (!(sin2 < 0.81f)) return false;
        r = wot * (-ior) + n * sqrtf(1.0f - sin2);
        cm = 0.98877108f;  // cos(0.15)
        sm = 0.14943813f;  // sin(0.15)
    }
    const float rl = norm(r);
    if (!(rl > 0.5f && rl < 2.0f)) return false;
    const float cos_a = dot(r, L) / (rl * D);
    return cos_a < (cl * cm - sl * sm) - 1e-3f;
}

// ------------------------------------------------------------------------------------------------
// k_shade: Scene::castRay (Scene.cpp:85-184) turned inside out.
//
// The recursion  L_d = clamp(0,15,l_dir_d) + clamp(0,5, L_{d+1} * f_d)  is not multiplicative (per-level
// clamps, unclamped early returns), so a path cannot carry a scalar throughput.  Each level that
// recurses pushes {clamp(0,15,l_dir), eval, |wo.n| (or -1 for Dirac), pdf} on a per-path clamp stack in
// HBM; when the path ends the stack is unwound with exactly the reference's float expressions.
// ------------------------------------------------------------------------------------------------
MCPT_DI float unwind(const RenderConst &C, uint32_t slot, uint32_t depth, float X) {
    for (int lvl = (int)depth - 1; lvl >= 0; --lvl) {
        const float4 e = C.stack[(size_t)lvl * C.pool + slot];
        float l_ind;
        if (e.z < 0.f) l_ind = X * e.y * C.inv_rr;            // Scene.cpp:137-138,164-165 (isDirac)
        else l_ind = X * e.y * e.z / e.w * C.inv_rr;           // Scene.cpp:140-143,167-170
        X = e.x + clampf(0, 5, l_ind);                         // Scene.cpp:180-183
    }
    return X;
}

__global__ __launch_bounds__(kShadeBlock, 8) void k_shade(DevScene S, RenderConst C, Wave cur, Wave next, Scratch Xs, int cur_idx) {
    __shared__ BlockAllocShared sh;
#ifdef MCPT_TRAVERSAL_STATS
    const long long t_wave0 = clock64();
#endif
    const uint32_t i = blockIdx.x * kShadeBlock + threadIdx.x;
    // the grid is an upper bound; the list lengths live on the device.  Lanes [0, n_rec) take the records of the list, the next
    // 3 * n_new lanes the new samples (one lane per channel path).
    const uint32_t n_rec = C.counters->n_paths[cur_idx].v, n_new = C.counters->n_prays[cur_idx].v;
    const bool valid = i < n_rec + 3u * n_new;
    const bool is_new = valid && i >= n_rec;
    const int next_idx = cur_idx ^ 1;

    uint32_t pid = 0, slot = 0, depth = 0, ray_idx = 0;
    int ch = 0, pq = 0;  // pq: parity of the pass the path belongs to
    RngKey key{0, 0, 0, 0};
    bool do_shade = false, finished = false, pushed = false, overflow = false;
    float X = 0.f;
    double hit_t = 0;
    int32_t hit_prim = -1;
    uint32_t hit_mat = 0;
    float4 ro4 = make_float4(0.f, 0.f, 0.f, 0.f), rd4 = make_float4(0.f, 0.f, 1.f, 0.f);

    if (valid) {
        uint4 r0;
        if (is_new) {  // a new sample: {sample | parity, ring position of its slots}; its ray is entry ray_cap-1-k
            const uint32_t k = (i - n_rec) / 3u, c = (i - n_rec) % 3u;
            const uint2 f = cur.fresh[k];
            r0 = make_uint4((f.x & 0x7fffffffu) * 3u + c, C.ray_cap - 1u - k, kFresh | ((f.x >> 31) ? kPassBit : 0u),
                            C.free_slots[(f.y + c) & C.free_mask]);
        } else {
            r0 = cur.rec0[i];
        }
        const uint32_t flags = r0.z;
        float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
        pid = r0.x;
        depth = r0.z & 0xffffu;
        if (flags & kTerminate) {  // no ray, no BSDF terms: the record is rec0 alone, with the slot where the ray index would be
            slot = r0.y;
        } else if (flags & kFresh) {  // no BSDF terms either: rec0 alone, with the slot where kr would be
            ray_idx = r0.y;
            slot = r0.w;
        } else {
            r1 = cur.rec1[i];
            ray_idx = r0.y;
            slot = __float_as_uint(r1.w);
        }
        pq = (int)((flags >> 20) & 1u);
        // everything the record points to is requested at once, before any of it is looked at: the hit, and the ray itself (a surface hit
        // needs both vectors, a miss the direction: only a path that ends on the depth limit asks for nothing)
        uint4 h = make_uint4(0u, 0u, 0xffffffffu, 0u);
        if (!(flags & kTerminate)) {
            h = cur.hit[ray_idx];
            ro4 = cur.ray_o[ray_idx];
            rd4 = cur.ray_d[ray_idx];
        }
        path_key(C, pid, pq, key, ch);

        if (!(flags & kFresh)) {
            // ---- resolve the pending vertex: Scene.cpp:114-119 (l_dir), 129-149 / 156-176 (l_ind)
            float dl = 0.f;
            if (!(flags & kNoDirect)) {
                if (C.n_dir == 4) {  // one 16-byte request instead of four dwords; same sum, same order (Scene.cpp:76 `l_dir +=`)
                    const float4 c4 = reinterpret_cast<const float4 *>(cur.contrib)[i];
                    dl = (((dl + c4.x) + c4.y) + c4.z) + c4.w;
                } else if ((C.n_dir & 3) == 0) {  // (32 light samples, as the README labels the chess render: 8 requests instead of 32)
                    const float4 *c4 = reinterpret_cast<const float4 *>(cur.contrib) + (size_t)i * (uint32_t)(C.n_dir >> 2);
                    for (int k = 0; k < (C.n_dir >> 2); ++k) {
                        const float4 v = c4[k];
                        dl = (((dl + v.x) + v.y) + v.z) + v.w;
                    }
                } else {
                    for (int k = 0; k < C.n_dir; ++k) dl += cur.contrib[(size_t)i * C.n_dir + k];  // Scene.cpp:76 `l_dir +=`, in order
                }
            }
            const float kr = __uint_as_float(r0.w);
            const float l_dir = (flags & kInside) ? (float)((1. - (double)kr) * (double)dl) : kr * dl;
            if (flags & kTerminate) {
                X = l_dir;  // Scene.cpp:129-131,156-158: returned unclamped
                finished = true;
            } else {
                hit_t = __longlong_as_double((long long)(((unsigned long long)h.y << 32) | h.x));
                hit_prim = (int32_t)h.z;
                hit_mat = h.w;
                const bool surface = hit_prim >= 0 && !(hit_mat >> 31);  // Scene.cpp:135,162
                if (!surface) {
                    const f3 wi = ld3(rd4);
                    const float env = comp(sample_env(S, wi), ch);  // Scene.cpp:145-149,172-176
                    const float l_ind = env * r1.x * C.inv_rr;
                    X = clampf(0, 15, l_dir) + clampf(0, 5, l_ind);
                    finished = true;
                } else if ((int)depth >= C.max_depth) {
                    X = clampf(0, 15, l_dir);  // clamp stack exhausted: drop the indirect term, report it
                    finished = true;
                    overflow = true;
                } else {
                    C.stack[(size_t)depth * C.pool + slot] = make_float4(clampf(0, 15, l_dir), r1.x, r1.y, r1.z);
                    depth += 1;
                    pushed = true;
                    do_shade = true;
                }
            }
        } else {
            hit_t = __longlong_as_double((long long)(((unsigned long long)h.y << 32) | h.x));
            hit_prim = (int32_t)h.z;
            hit_mat = h.w;
            if (hit_prim < 0) {
                X = comp(sample_env(S, ld3(rd4)), ch);  // Scene.cpp:88-95
                finished = true;
            } else {
                do_shade = true;  // the depth-0 emitter test needs the normal; done below
            }
        }
    }

    // ---- vertex set-up for lanes that shade
    f3 ro = mk3(0, 0, 0), rd = mk3(0, 0, 1), p = mk3(0, 0, 0), n = mk3(0, 0, 1), wo = mk3(0, 0, -1);
    f2 uv{0.f, 0.f};
    const int mat_id = (int)(hit_mat & kMatIndexMask);
    if (do_shade) {
        ro = ld3(ro4);
        rd = ld3(rd4);
        if (hit_prim < S.n_tri) {
            const TriShade ts = S.tri_shade[hit_prim];
            n = mk3(ts.n[0], ts.n[1], ts.n[2]);
            p = ro + rd * (float)hit_t;  // Triangle.hpp:245 / Ray.hpp:21
            if (hit_mat & kMatTextured) {  // Triangle.hpp:248: recompute the barycentrics of the recorded hit (the flag travels with the hit)
                double t, u, v;
                const Ray rr = make_ray(ro, rd);
                if (tri_hit(S.tri_geom[hit_prim], rr, t, u, v)) {
                    const float a = (float)(1 - u - v), b = (float)u, c = (float)v;
                    uv.x = a * ts.t0[0] + b * ts.t1[0] + c * ts.t2[0];
                    uv.y = a * ts.t0[1] + b * ts.t1[1] + c * ts.t2[1];
                }
            }
        } else {
            const SphereRec s = S.spheres[hit_prim - S.n_tri];
            p = ro + rd * (float)hit_t;  // Sphere.hpp:40 (t0 is a float)
            n = normalized(p - mk3(s.c[0], s.c[1], s.c[2]));
        }
        wo = -rd;
        if (depth == 0 && (hit_mat >> 31)) {  // Scene.cpp:102-107
            X = clampf(0, 1, S.mats[mat_id].emit[ch] * fabsf(dot(wo, n)));
            finished = true;
            do_shade = false;
        }
    }

    // ---- finish: unwind the clamp stack and publish the path value
    if (finished) {
        X = unwind(C, slot, depth, X);
        (pq ? C.result[1] : C.result[0])[pid] = X;
    }

    // ---- shade the new vertex: Scene.cpp:109-128,150-155
    const MaterialRec m = S.mats[mat_id];
    float u0[4] = {0.f, 0.f, 1.f, 0.f};
    if (do_shade) rng_block(key, depth, 0u, u0);
    const bool has_cont = do_shade && !(u0[2] >= C.rr_rate);  // Scene.cpp:121,129,156

    const f3 q = p + n * kEps;                       // Scene.cpp:114
    const bool inside = dot(wo, n) < 0;              // Scene.cpp:115
    const bool zero_direct = do_shade && direct_is_zero(S, m, q, n, wo, inside, ch);
#ifdef MCPT_CHECK_DIRECT_SKIP
    const bool need_direct = do_shade;  // checking build: evaluate the skipped vertices too (k_direct counts violations)
#else
    const bool need_direct = do_shade && !zero_direct;
#endif

    // one round of block-aggregated atomics: released slots, next-list records, continuation rays, the direct-
    // lighting work list, statistics.  The BSDF sampling below does not need the indices and runs while the
    // atomics are in flight.
    // A vertex where roulette ends the path and whose light samples all contribute zero returns l_dir = kr * 0
    // (Scene.cpp:129-131,156-158) with nothing left to wait for: the path is finished here instead of leaving a
    // record for the next iteration.
    const bool ends_here = do_shade && !has_cont && !need_direct;
    const bool done = finished || ends_here;
    const bool want[9] = {done, do_shade && !ends_here, has_cont, need_direct, pushed, overflow, done && C.track_live && pq == 0,
                          done && C.track_live && pq == 1, ends_here};
    const uint32_t mult[9] = {1u, 1u, 1u, 1u, 1u, 1u, 1u, 1u, 1u};
    uint32_t *const ctr[9] = {&C.counters->free_tail.v, &C.counters->n_paths[next_idx].v, &C.counters->n_rays[next_idx].v,
                              &C.counters->n_direct[next_idx].v, &C.counters->pushes.v, &C.counters->overflow.v,
                              &C.counters->live[0].v, &C.counters->live[1].v, &C.counters->ended.v};
    const bool sub[9] = {false, false, false, false, false, false, true, true, false};
    uint32_t prefix[9], idx[9];
#ifdef MCPT_TRAVERSAL_STATS
    const long long tb0 = clock64();
#endif
    block_alloc_begin<9>(sh, want, mult, ctr, sub, prefix);
#ifdef MCPT_TRAVERSAL_STATS
    const long long tb1 = clock64();
#endif

    float kr = 0.f, ev = 0.f, aw = 0.f, pd = 0.f;
    f3 p2 = mk3(0, 0, 0), wi = mk3(0, 0, 1);
    if (do_shade) {
        const f3 mfn = mat_sample(m, n, u0[0], u0[1]);   // Scene.cpp:109
        kr = mat_fresnel(m, rd, mfn, ch);                // Scene.cpp:110
        const bool isReflect = u0[3] < kr;               // Scene.cpp:123
        if (isReflect) p2 = (dot(wo, mfn) < 0) ? (p - n * kEps) : (p + n * kEps);  // Scene.cpp:124-128
        else p2 = (dot(wo, mfn) < 0) ? (p + n * kEps) : (p - n * kEps);            // Scene.cpp:151-155
        if (has_cont) {
            wi = isReflect ? mat_reflect(wo, mfn) : mat_refract(m, rd, mfn, ch);   // Scene.cpp:132,159
            if (m.isDirac) {
                ev = mat_eval(m, wi, wo, n, ch, uv, isReflect);
                aw = -1.f;
            } else {
                aw = fabsf(dot(wo, n));
                mat_eval_pdf_rough(m, wi, wo, n, ch, uv, isReflect, ev, pd);  // Material::eval and ::pdf sharing h and D
            }
        }
    }

#ifdef MCPT_TRAVERSAL_STATS
    const long long tb2 = clock64();
#endif
    block_alloc_end<9>(sh, mult, prefix, idx);
#ifdef MCPT_TRAVERSAL_STATS
    if (S.dbg && lane_id() == 0) {  // cycles of this wave: in block_alloc_begin (ballots + first barrier), in block_alloc_end (second barrier), since its start
        const long long tb3 = clock64();
        atomicAdd(&S.dbg[20], (unsigned long long)(tb1 - tb0));
        atomicAdd(&S.dbg[21], (unsigned long long)(tb3 - tb2));
        atomicAdd(&S.dbg[22], (unsigned long long)(tb3 - t_wave0));
        atomicAdd(&S.dbg[23], 1ull);
    }
    if (S.dbg) {  // material divergence: how many of the four material types the shading lanes of this wave hold
        int nd = 0;
        for (int t = 0; t < 4; ++t) nd += __ballot(do_shade && m.type == t) != 0ull ? 1 : 0;
        const unsigned long long ms = __ballot(do_shade);
        if (lane_id() == 0) {
            atomicAdd(&S.dbg[24 + nd], 1ull);
            atomicAdd(&S.dbg[29], (unsigned long long)__popcll(ms));
        }
    }
#endif
    if (done) C.free_slots[idx[0] & C.free_mask] = slot;
    if (!do_shade) return;
    if (ends_here) {
        const float l_dir = inside ? (float)((1. - (double)kr) * (double)0.f) : kr * 0.f;  // Scene.cpp:116-119 with l_dir == 0
        (pq ? C.result[1] : C.result[0])[pid] = unwind(C, slot, depth, l_dir);
        return;
    }
    const uint32_t j = idx[1], rj = idx[2], dj = idx[3];

    if (need_direct) {  // work-list entry for k_direct (Scene::directLighting runs there, one lane per light sample)
        Xs.vtx0[dj] = make_float4(q.x, q.y, q.z, uv.x);
        Xs.vtx1[dj] = make_float4(n.x, n.y, n.z, uv.y);
        Xs.vtx2[dj] = make_float4(wo.x, wo.y, wo.z, __uint_as_float((uint32_t)mat_id | ((uint32_t)ch << 16) | (inside ? (1u << 18) : 0u) |
                                                                    (zero_direct ? (1u << 19) : 0u)));
        Xs.vtx_j[dj] = j;
    }
    uint32_t flags = depth | (inside ? kInside : 0u) | (need_direct ? 0u : kNoDirect) | (pq ? kPassBit : 0u);
    if (has_cont) {
        next.ray_o[rj] = make_float4(p2.x, p2.y, p2.z, 0.f);
        next.ray_d[rj] = make_float4(wi.x, wi.y, wi.z, 0.f);
    } else {
        flags |= kTerminate;
    }
    next.rec0[j] = make_uint4(pid, has_cont ? rj : slot, flags, __float_as_uint(kr));
    if (has_cont) next.rec1[j] = make_float4(ev, aw, pd, __uint_as_float(slot));
}

// ------------------------------------------------------------------------------------------------
// k_direct: Scene::directLighting (Scene.cpp:56-82), one lane per (shaded vertex, light sample).
//   c = emit * eval(ws, wo, n) * (ws.n) * (-ws.n_light) / dist^2 / pdf / n_dir_sample          (Scene.cpp:76-79)
// is stored in contrib[]; samples with c != 0 go to the shadow queue.  A sample whose contribution is exactly
// +-0 (Dirac BSDFs away from the mirror direction, back-facing configurations: Material.hpp:338,356,382,397)
// adds nothing to l_dir whether it is visible or not, so it casts no shadow ray; a NaN contribution is not zero
// and is traced.
// ------------------------------------------------------------------------------------------------
// (7 waves per SIMD: 70 VGPRs, nothing spilled.  With the former workgroup-wide queue allocation 8 had been better -- more waves to
// hide its barriers; with the per-wave allocation below, 8 spills 27 registers: frame 4880 against 5000.)
#ifndef MCPT_DIRECT_WAVES
#define MCPT_DIRECT_WAVES 7
#endif
// SMALL: light tables, materials and the triangles / spheres in LDS (small scenes, see SmallGeomLds).
// (Round 3, measured and not kept: for these scenes, where 75-86 % of the light samples do cast a ray, the shadow query run right here in
// the lane that made the sample -- no queue entry, no prefix sums, no second launch.  Correct, and slower: cornell_rc 784^2 spp 256
// 427 against 475 Msamples/s, DEMO 1080p 755 against 857: the fused kernel took 288 ms where k_direct + k_trace_shadow take 112 + 137
// side by side with the other chains.)
template <bool SMALL>
__global__ __launch_bounds__(kBlock, MCPT_DIRECT_WAVES) void k_direct(DevScene S, RenderConst C, Wave next, Scratch Xs, int next_idx) {
    if constexpr (SMALL) {
        __shared__ SmallGeomLds small_geom;
        __shared__ SmallLightLds small_lights;
        stage_small_geom(S, small_geom);
        stage_small_lights(S, small_lights);
        __syncthreads();
    }
    const uint32_t n_dir = (uint32_t)C.n_dir;
    const uint32_t total = C.counters->n_direct[next_idx].v * n_dir;  // the grid is sized from an estimate: stride over the list
    for (uint32_t base = blockIdx.x * kBlock; base < total; base += gridDim.x * kBlock) {  // uniform trip count per block
    const uint32_t g = base + threadIdx.x;
    const bool valid = g < total;
    bool cast = false, window = false;
    f3 q = mk3(0, 0, 0), ws = mk3(0, 0, 1);
    float dist = 0.f;
    uint32_t target = 0;
    if (valid) {
        uint32_t dj, k;
        if (n_dir == 4u) {  // the reference's fixed count (Scene.hpp:28): no division
            dj = g >> 2;
            k = g & 3u;
        } else if ((n_dir & (n_dir - 1u)) == 0u) {  // another power of two (32, the README's count): still no division
            dj = g >> (uint32_t)(__ffs((int)n_dir) - 1);
            k = g & (n_dir - 1u);
        } else {
            dj = g / n_dir;
            k = g - dj * n_dir;
        }
        const float4 v0 = Xs.vtx0[dj], v1 = Xs.vtx1[dj], v2 = Xs.vtx2[dj];
        const uint32_t j = Xs.vtx_j[dj];
        target = j * n_dir + k;
        const uint4 r0 = next.rec0[j];
        const uint32_t bits = __float_as_uint(v2.w);
        const MaterialRec m = S.mats[bits & 0xffffu];
        const bool inside = (bits >> 18) & 1u;
        q = ld3(v0);
        const f3 n = ld3(v1), wo = ld3(v2);
        const f2 uv{v0.w, v1.w};
        RngKey key;
        int ch;
        path_key(C, r0.x, (int)((r0.z >> 20) & 1u), key, ch);
        float u[4];
        rng_block(key, r0.z & 0xffffu, 1u + k, u);
        f3 x_l = mk3(0, 0, 0), n_l = mk3(0, 0, 0), emit3 = mk3(0, 0, 0);
        float pdf = 0.f, c = 0.f;
        int32_t light_prim = -1;
        if (sample_light(S, u, x_l, n_l, emit3, pdf, light_prim)) {
            const float emit = comp(emit3, ch);
            ws = normalized(x_l - q);
            dist = norm(x_l - q);
            c = emit * mat_eval(m, ws, wo, n, ch, uv, !inside) * (dot(ws, n)) * dot(-ws, n_l) / (dist * dist) / pdf / C.n_dir;
        }
        cast = C.enable_shadow && !(c == 0.f);
        if (cast) {
            // Scene.cpp:72-75: the sample counts iff the closest hit of the shadow ray lies within EPSILON of `dist`.  The
            // ray is aimed at a point of primitive light_prim, so that primitive is tested here: a hit closer than
            // dist - EPSILON settles the sample as invisible (no ray at all); a hit inside the window settles the window
            // search, leaving only the occluder search to k_trace_shadow; otherwise the full query runs.
            double t = 0;
            window = true;
            if (prim_hit(S, light_prim, make_ray(q, ws), t)) {
                const double dd = t - (double)dist;
                if (dd <= -(double)kEps) {
                    c = 0.f;
                    cast = false;
                } else if (fabs(dd) < (double)kEps) {
                    window = false;
                }
            }
        }
#ifdef MCPT_CHECK_DIRECT_SKIP
        if (((bits >> 19) & 1u) && S.dbg) {
            atomicAdd(&S.dbg[14], 1ull);
            if (!(c == 0.f)) atomicAdd(&S.dbg[15], 1ull);
        }
#endif
        next.contrib[target] = c;
    }
    // queue entries: one atomic per wave and kind on the counters of the wave's shard (see Counters): no barrier, no LDS
    const unsigned long long mf = __ballot(cast && !window), mw = __ballot(cast && window);
    if (mf | mw) {
        const uint32_t shard = (g >> 6) & (kShadowShards - 1u);
        uint32_t bf = 0, bw = 0;
        if (lane_id() == 0) {
            if (mf) bf = atomicAdd(&C.counters->n_shadow[next_idx][shard].v, (uint32_t)__popcll(mf));
            if (mw) bw = atomicAdd(&C.counters->n_shadow_w[next_idx][shard].v, (uint32_t)__popcll(mw));
        }
        bf = (uint32_t)__shfl((int)bf, 0);
        bw = (uint32_t)__shfl((int)bw, 0);
        if (cast) {
            const uint32_t region = shadow_region((uint32_t)C.pool * n_dir);
            const unsigned long long m = window ? mw : mf;
            const uint32_t off = (window ? bw : bf) + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
            const uint32_t e = window ? (shard + 1u) * region - 1u - off : shard * region + off;
            Xs.shq_o[e] = make_float4(q.x, q.y, q.z, __uint_as_float(target));
            Xs.shq_d[e] = make_float4(ws.x, ws.y, ws.z, dist);
        }
    }
    }
}

// ------------------------------------------------------------------------------------------------
// k_accumulate: framebuffer[m] += Vector3f(R,G,B)/spp, samples in order (Renderer.cpp:80).
// One lane per (owned pixel, channel).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_accumulate(const float *__restrict__ result, const uint32_t *__restrict__ pixel_list,
                                                       uint32_t n_pix, int32_t s_pass, float spp_total, float *__restrict__ fb) {
    const uint32_t g = blockIdx.x * kBlock + threadIdx.x;
    if (g >= n_pix * 3u) return;
    const uint32_t pl = g / 3u, c = g % 3u;
    const uint32_t m = pixel_list ? pixel_list[pl] : pl;
    float acc = fb[(size_t)m * 3 + c];
    const float *src = result + (size_t)pl * s_pass * 3 + c;
    int k = 0;
    for (; k + 8 <= s_pass; k += 8) {  // eight loads in flight; the additions stay in sample order (Renderer.cpp:80)
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)(k + u) * 3];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u] / spp_total;
    }
    for (; k < s_pass; ++k) acc += src[(size_t)k * 3] / spp_total;
    fb[(size_t)m * 3 + c] = acc;
}

__global__ __launch_bounds__(kBlock) void k_mask_unowned(float *fb, int W, int H, int tile, int rank, int nranks) {
    const uint32_t m = blockIdx.x * kBlock + threadIdx.x;
    if (m >= (uint32_t)W * (uint32_t)H) return;
    const int i = (int)(m % (uint32_t)W), j = (int)(m / (uint32_t)W);
    const int tx = (W + tile - 1) / tile;
    if (((j / tile) * tx + i / tile) % nranks == rank) return;  // owned (mcpt_params: tile_size / rank / nranks)
    fb[(size_t)m * 3] = 0.f;
    fb[(size_t)m * 3 + 1] = 0.f;
    fb[(size_t)m * 3 + 2] = 0.f;
}

__global__ __launch_bounds__(kBlock) void k_add_frame(float *__restrict__ a, const float *__restrict__ b, uint32_t n) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) a[i] += b[i];
}

__global__ __launch_bounds__(kBlock) void k_debug_fmath(int kind, uint32_t n, const float *x, const float *y, float *out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
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

// Material functions on arrays (mcpt_debug_material): rows of `in` are {a.xyz, b.xyz, c.xyz, uv.xy, u1, u2}, `sel` = {material, channel,
// is_reflect}; out = 4 floats per row.  kind 0 eval(wi=a, wo=b, n=c), 1 pdf, 2 fresnel(I=a, N=b), 3 sample(n=a, u1, u2), 4 refract(I=a, N=b),
// 5 the fused eval+pdf of k_shade (out = {eval, pdf}), 6 reflect(I=a, N=b).
__global__ __launch_bounds__(kBlock) void k_debug_material(DevScene S, int kind, uint32_t n, const float *__restrict__ in, const int32_t *__restrict__ sel,
                                                           float *__restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const float *r = in + (size_t)i * 13;
    const f3 a = mk3(r[0], r[1], r[2]), b = mk3(r[3], r[4], r[5]), c = mk3(r[6], r[7], r[8]);
    const f2 uv{r[9], r[10]};
    const MaterialRec m = S.mats[sel[3 * i]];
    const int ch = sel[3 * i + 1];
    const bool refl = sel[3 * i + 2] != 0;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    switch (kind) {
    case 0: o[0] = mat_eval(m, a, b, c, ch, uv, refl); break;
    case 1: o[0] = mat_pdf(m, a, b, c, ch, refl); break;
    case 2: o[0] = mat_fresnel(m, a, b, ch); break;
    case 3: {
        const f3 v = mat_sample(m, a, r[11], r[12]);
        o[0] = v.x, o[1] = v.y, o[2] = v.z;
        break;
    }
    case 4: {
        const f3 v = mat_refract(m, a, b, ch);
        o[0] = v.x, o[1] = v.y, o[2] = v.z;
        break;
    }
    case 5: mat_eval_pdf_rough(m, a, b, c, ch, uv, refl, o[0], o[1]); break;
    default: {
        const f3 v = mat_reflect(a, b);
        o[0] = v.x, o[1] = v.y, o[2] = v.z;
        break;
    }
    }
    for (int k = 0; k < 4; ++k) out[(size_t)i * 4 + k] = o[k];
}

// Scene functions on arrays (mcpt_debug_scene): kind 0 Scene::sampleLight for 4 uniforms per row -> {x_l, n_l, emit, pdf} (10 floats; zeros
// when no light was selected), kind 1 Scene::sampleEnv for a direction per row -> rgb.
__global__ __launch_bounds__(kBlock) void k_debug_scene(DevScene S, int kind, uint32_t n, const float *__restrict__ in, float *__restrict__ out) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (kind == 0) {
        const float u[4] = {in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]};
        f3 x = mk3(0, 0, 0), nl = mk3(0, 0, 0), e = mk3(0, 0, 0);
        float pdf = 0.f;
        int32_t prim = -1;
        (void)sample_light(S, u, x, nl, e, pdf, prim);
        float *o = out + (size_t)i * 10;
        o[0] = x.x, o[1] = x.y, o[2] = x.z, o[3] = nl.x, o[4] = nl.y, o[5] = nl.z, o[6] = e.x, o[7] = e.y, o[8] = e.z, o[9] = pdf;
    } else {
        const f3 c = sample_env(S, mk3(in[3 * i], in[3 * i + 1], in[3 * i + 2]));
        out[3 * i] = c.x, out[3 * i + 1] = c.y, out[3 * i + 2] = c.z;
    }
}

// Renderer.cpp:95-103 on the device: one lane per pixel, RGBA8 out (alpha 255).
__global__ __launch_bounds__(kBlock) void k_tonemap(const float *__restrict__ fb, uint32_t n_pix, uchar4 *__restrict__ rgba) {
    const uint32_t i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_pix) return;
    rgba[i] = make_uchar4(mcpt_tonemap_byte(fb[(size_t)i * 3]), mcpt_tonemap_byte(fb[(size_t)i * 3 + 1]), mcpt_tonemap_byte(fb[(size_t)i * 3 + 2]), 255);
}

inline uint32_t blocks(uint32_t n) { return (n + kBlock - 1) / kBlock; }

}  // namespace

// Runs after k_shade(cur -> next).  List `cur` and everything indexed like it (its closest-hit queue, k_direct work
// list and shadow queue, all produced one iteration earlier) are consumed: their lengths go to the totals and are
// cleared, because `cur` is the next iteration's output list.
// In the regular schedule k_primary may already be appending fresh records to list `next` on another stream, so the host
// passes the lengths it read right after k_shade (from_host); in the drain phase they are read here.
__global__ void k_bookkeep(Counters *c, int cur_idx, int from_host, uint32_t n_next, uint32_t n_cont, uint32_t n_direct) {
    const int nxt = cur_idx ^ 1;
    // The 2 x 32 shard counters of the consumed shadow queue: one lane each, summed over the wave.  (Until round 3 lane 0 walked them in a
    // loop: 64 dependent global round trips, 115 us per launch of a kernel that sits between k_shade and k_direct on the main stream --
    // 29 ms of a 650 ms frame.)
    static_assert(2 * kShadowShards == 64, "one lane per shard counter");
    HotCounter *const shard = threadIdx.x < kShadowShards ? &c->n_shadow[cur_idx][threadIdx.x] : &c->n_shadow_w[cur_idx][threadIdx.x - kShadowShards];
    uint32_t n_shadow = shard->v;
    shard->v = 0;
    for (int o = 32; o > 0; o >>= 1) n_shadow += (uint32_t)__shfl_xor((int)n_shadow, o);
    switch (threadIdx.x) {  // one lane per field: the global accesses are independent and overlap
    case 0: {
        c->tot_shadow += n_shadow;
        c->last_shadow = n_shadow;
        break;
    }
    case 1: c->tot_shaded += from_host ? n_next : c->n_paths[nxt].v; break;  // (new samples have no records)
    case 2: c->tot_cont += from_host ? n_cont : c->n_rays[nxt].v; break;
    case 3: c->tot_direct += from_host ? n_direct : c->n_direct[nxt].v; break;
    case 4: c->tot_iterations += 1; break;
    case 8: {  // only k_shade (same stream, already finished) writes these two
        const uint32_t v = c->ended.v;
        c->ended.v = 0;
        c->tot_ended += v;
        break;
    }
    case 9: {
        const uint32_t v = c->pushes.v;
        c->pushes.v = 0;
        c->tot_pushes += v;
        break;
    }
    case 5: c->n_paths[cur_idx].v = 0; break;
    case 6: c->n_rays[cur_idx].v = 0; break;
    case 10: c->n_prays[cur_idx].v = 0; break;
    case 7: c->n_direct[cur_idx].v = 0; break;
    default: break;
    }
}

void launch_bookkeep(Counters *c, int cur_idx, bool from_host, uint32_t n_next, uint32_t n_cont, uint32_t n_direct, hipStream_t s) {
    hipLaunchKernelGGL(k_bookkeep, dim3(1), dim3(64), 0, s, c, cur_idx, from_host ? 1 : 0, n_next, n_cont, n_direct);
}

void launch_init_free(uint32_t *free_slots, Counters *c, uint32_t pool, uint32_t start, uint32_t mask, hipStream_t s) {
    hipLaunchKernelGGL(k_init_free, dim3(blocks(pool)), dim3(kBlock), 0, s, free_slots, c, pool, start, mask);
}

// Stack flavour by tree height (see MCPT_STK_PUSH): <= 17 levels: 16 LDS entries; <= 20: kStkB; <= kPlainMaxHeight (24): 24; deeper: the retry
// flavour (16 LDS entries, retrace list).  (-DMCPT_LDS_ONLY_STACKS: 24 / 32 / 48-entry LDS stacks for deep trees instead, for A/B measurements;
// -DMCPT_FORCE_RETRY, the checking build: the retry flavour for every tree with -DMCPT_STK_RETRY=4 LDS entries, so that most rays are
// traced again.)  RETRACE: the launch of the retrace kernel, issued right behind a retry-flavour kernel on the same stream.
constexpr uint32_t kRetraceGrid = 256;
#ifdef MCPT_LDS_ONLY_STACKS
#define MCPT_STACK_DISPATCH(height, KERNEL, RETRACE, ...)                                          \
    do {                                                                                           \
        if ((height) <= 17) hipLaunchKernelGGL((KERNEL<16, false, false>), __VA_ARGS__);           \
        else if ((height) <= 20) hipLaunchKernelGGL((KERNEL<kStkB, false, false>), __VA_ARGS__);   \
        else if ((height) <= 24) hipLaunchKernelGGL((KERNEL<24, false, false>), __VA_ARGS__);      \
        else if ((height) <= 32) hipLaunchKernelGGL((KERNEL<32, false, false>), __VA_ARGS__);      \
        else hipLaunchKernelGGL((KERNEL<kMaxBvhHeight, false, false>), __VA_ARGS__);               \
    } while (0)
#elif defined(MCPT_FORCE_RETRY)
#define MCPT_STACK_DISPATCH(height, KERNEL, RETRACE, ...)                                          \
    do {                                                                                           \
        hipLaunchKernelGGL((KERNEL<kStkRetry, true, false>), __VA_ARGS__);                         \
        RETRACE;                                                                                   \
    } while (0)
#else
#define MCPT_STACK_DISPATCH(height, KERNEL, RETRACE, ...)                                          \
    do {                                                                                           \
        if (S.small) hipLaunchKernelGGL((KERNEL<kSmallStk, false, true>), __VA_ARGS__); /* scene in LDS */ \
        else if ((height) <= 17) hipLaunchKernelGGL((KERNEL<16, false, false>), __VA_ARGS__);      \
        else if ((height) <= 20) hipLaunchKernelGGL((KERNEL<kStkB, false, false>), __VA_ARGS__);   \
        else if ((height) <= kPlainMaxHeight) hipLaunchKernelGGL((KERNEL<24, false, false>), __VA_ARGS__); \
        else {                                                                                     \
            hipLaunchKernelGGL((KERNEL<kStkRetry, true, false>), __VA_ARGS__);                     \
            RETRACE;                                                                               \
        }                                                                                          \
    } while (0)
#endif

void launch_primary(const DevScene &S, const CameraConst &cam, const RenderConst &C, Wave next, int next_idx, int parity,
                    uint32_t first_sample, uint32_t n_samples, const RetryList &rl, hipStream_t s) {
    if (n_samples == 0) return;
    const dim3 g(blocks(n_samples)), b(kBlock);
    MCPT_STACK_DISPATCH(S.height, k_primary, hipLaunchKernelGGL(k_primary_retrace, dim3(kRetraceGrid), b, 0, s, S, cam, C, next, next_idx, parity, rl), g, b, 0, s, S,
                        cam, C, next, next_idx, parity, first_sample, n_samples, rl);
}

void launch_generate_explicit(const RenderConst &C, Wave next, int next_idx, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_generate_explicit, dim3(blocks(n)), dim3(kBlock), 0, s, next, C.counters, next_idx, n);
}

void launch_camera_rays(const CameraConst &cam, uint32_t seed, uint32_t n, const uint32_t *pixel, const uint32_t *sample,
                        float4 *o, float4 *d, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_camera_rays, dim3(blocks(n)), dim3(kBlock), 0, s, cam, seed, n, pixel, sample, o, d);
}

void launch_trace_closest(const DevScene &S, uint32_t n, const uint32_t *n_dev, const float4 *ray_o, const float4 *ray_d, uint4 *hit,
                          const RetryList &rl, hipStream_t s) {
    if (n == 0) return;
#ifndef MCPT_NO_REFILL
    if (S.qnodes && !S.inst) {  // the common configuration: lanes refill from the wave's chunk of rays
        const dim3 g((n + kBlock * kRaysPerLane - 1) / (kBlock * kRaysPerLane)), b(kBlock);
        MCPT_STACK_DISPATCH(S.height, k_trace_closest_refill, hipLaunchKernelGGL(k_retrace_closest, dim3(kRetraceGrid), b, 0, s, S, ray_o, ray_d, hit, rl), g, b, 0, s, S,
                            n, n_dev, ray_o, ray_d, hit, rl);
        return;
    }
#endif
    const dim3 g(blocks(n)), b(kBlock);
    MCPT_STACK_DISPATCH(S.height, k_trace_closest, hipLaunchKernelGGL(k_retrace_closest, dim3(kRetraceGrid), b, 0, s, S, ray_o, ray_d, hit, rl), g, b, 0, s, S, n, n_dev,
                        ray_o, ray_d, hit, rl);
}

void launch_direct(const DevScene &S, const RenderConst &C, Wave next, Scratch X, int next_idx, uint32_t n_vertices_grid, uint32_t small_per_cu, hipStream_t s) {
    if (n_vertices_grid == 0) return;
    dim3 g(blocks(n_vertices_grid * (uint32_t)C.n_dir)), b(kBlock);
    // (SMALL: every workgroup first copies the scene into LDS; a capped grid lets it stride over several chunks of the list for one copy)
    if (S.small && small_per_cu) g.x = std::min<uint32_t>(g.x, 256u * small_per_cu);
    if (S.small) hipLaunchKernelGGL((k_direct<true>), g, b, 0, s, S, C, next, X, next_idx);
    else hipLaunchKernelGGL((k_direct<false>), g, b, 0, s, S, C, next, X, next_idx);
}

void launch_trace_shadow(const DevScene &S, const Counters *counters, int next_idx, uint32_t n_max, uint32_t cap, Scratch X,
                         float *contrib, uint32_t per_cu, const RetryList &rl, hipStream_t s) {
    if (n_max == 0) return;
    // The queue length is only known on the device.  The grid covers the upper bound, capped at `per_cu` workgroups per CU (Knobs:
    // 128, far more than are resident, so the hardware balances uneven rays dynamically: a persistent 8-per-CU grid
    // was 30 % slower on the Cornell box); workgroups past the end of the queue exit at once, longer queues stride.
    const dim3 g(std::min<uint32_t>(blocks(n_max), 256u * per_cu)), b(kBlock);
    MCPT_STACK_DISPATCH(S.height, k_trace_shadow,
                        hipLaunchKernelGGL(k_retrace_shadow, dim3(kRetraceGrid), b, 0, s, S, counters, next_idx, cap, X.shq_o, X.shq_d, contrib, rl), g, b, 0, s, S,
                        counters, next_idx, cap, X.shq_o, X.shq_d, contrib, rl);
}

void launch_shade(const DevScene &S, const RenderConst &C, Wave cur, Wave next, Scratch X, int cur_idx, uint32_t n_cur_max,
                  hipStream_t s) {
    if (n_cur_max == 0) return;
    hipLaunchKernelGGL(k_shade, dim3((n_cur_max + kShadeBlock - 1) / kShadeBlock), dim3(kShadeBlock), 0, s, S, C, cur, next, X, cur_idx);
}

void launch_tonemap(const float *fb, uint32_t n_pix, unsigned char *rgba, hipStream_t s) {
    if (n_pix == 0) return;
    hipLaunchKernelGGL(k_tonemap, dim3(blocks(n_pix)), dim3(kBlock), 0, s, fb, n_pix, reinterpret_cast<uchar4 *>(rgba));
}

void launch_mask_unowned(float *fb, int width, int height, int tile, int rank, int nranks, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_unowned, dim3(blocks((uint32_t)width * (uint32_t)height)), dim3(kBlock), 0, s, fb, width, height, tile, rank, nranks);
}

void launch_add_frame(float *a, const float *b, uint32_t n, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_add_frame, dim3(blocks(n)), dim3(kBlock), 0, s, a, b, n);
}

void launch_debug_fmath(int kind, uint32_t n, const float *x, const float *y, float *out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_debug_fmath, dim3(blocks(n)), dim3(kBlock), 0, s, kind, n, x, y, out);
}

void launch_debug_scene(const DevScene &S, int kind, uint32_t n, const float *in, float *out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_debug_scene, dim3(blocks(n)), dim3(kBlock), 0, s, S, kind, n, in, out);
}

void launch_debug_material(const DevScene &S, int kind, uint32_t n, const float *in, const int32_t *sel, float *out, hipStream_t s) {
    if (n == 0) return;
    hipLaunchKernelGGL(k_debug_material, dim3(blocks(n)), dim3(kBlock), 0, s, S, kind, n, in, sel, out);
}

void launch_accumulate(const float *result, const uint32_t *pixel_list, uint32_t n_pix, int32_t s_pass, float spp_total,
                       float *fb, hipStream_t s) {
    if (n_pix == 0) return;
    hipLaunchKernelGGL(k_accumulate, dim3(blocks(n_pix * 3u)), dim3(kBlock), 0, s, result, pixel_list, n_pix, s_pass, spp_total, fb);
}

}  // namespace mcpt
