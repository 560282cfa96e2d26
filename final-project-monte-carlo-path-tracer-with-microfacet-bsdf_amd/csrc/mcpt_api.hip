// C ABI of libmcpt_hip.so (include/mcpt.h) and the host-side wavefront loop.
//
// The loop replaces the pixel/spp loops of Renderer::Render (reference src/Renderer.cpp:36-90).  Samples are
// streamed through a fixed pool of path records: every iteration shades all live records, refills the
// pool with new camera samples ("path regeneration") and traces all rays of the iteration in two launches
// (closest-hit queue, shadow queue), so the GPU always works on full, compacted queues.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mcpt_kernels.h"
#include "mcpt_cull.h"
#include "mcpt_lbvh.h"

using namespace mcpt;

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess)                                                                           \
            return fail(e_ == hipErrorOutOfMemory ? MCPT_ERR_OOM : MCPT_ERR_HIP,                        \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                             \
    } while (0)

// Owning device allocation: freed by release() or when it goes out of scope, so an early error return cannot leak it.
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    hipError_t alloc(size_t count) {
        if (count <= n && p) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T));
        if (e == hipSuccess) n = count;
        else p = nullptr;
        return e;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    size_t bytes() const { return n * sizeof(T); }
};

template <typename T>
hipError_t upload(DevBuf<T> &b, const std::vector<T> &v) {
    hipError_t e = b.alloc(v.size());
    if (e != hipSuccess) return e;
    if (v.empty()) return hipSuccess;
    return hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
}

struct WaveBufs {
    DevBuf<uint4> rec0, hit;
    DevBuf<float4> rec1, ray_o, ray_d;
    DevBuf<float> contrib;
    DevBuf<uint2> fresh;
    Wave view() const { return Wave{rec0.p, rec1.p, ray_o.p, ray_d.p, hit.p, contrib.p, fresh.p}; }
    void release() {
        rec0.release(); hit.release(); rec1.release(); ray_o.release(); ray_d.release(); contrib.release(); fresh.release();
    }
};

// The retrace lists of the traversal kernels (RetryList, csrc/mcpt_kernels.h): allocated only for scenes whose tree runs the retry
// flavour of the traversal stack (stack_uses_retry); each list holds as many entries as one launch can have rays.
struct RetryBufs {
    DevBuf<uint32_t> ctl;  // {count, done} x 3, zero between launches
    DevBuf<uint32_t> items[3];
    uint32_t cap[3] = {0, 0, 0};
    hipError_t alloc(const uint32_t want[3]) {
        hipError_t e = ctl.alloc(8);
        if (e != hipSuccess) return e;
        if ((e = hipMemset(ctl.p, 0, 8 * sizeof(uint32_t))) != hipSuccess) return e;
        for (int k = 0; k < 3; ++k) {
            if ((e = items[k].alloc(want[k])) != hipSuccess) return e;
            cap[k] = want[k];
        }
        return hipSuccess;
    }
    RetryList list(int k) const { return ctl.p ? RetryList{ctl.p + 2 * k, ctl.p + 2 * k + 1, items[k].p, cap[k]} : RetryList{nullptr, nullptr, nullptr, 0u}; }
    void release() {
        ctl.release();
        for (int k = 0; k < 3; ++k) items[k].release();
    }
};

struct Workspace {
    uint32_t pool = 0, free_ring = 0, ray_cap = 0;
    RetryBufs retry;  // 0 closest-hit rays, 1 shadow rays, 2 primary samples
    int32_t n_dir = 0, max_depth = 0;
    WaveBufs wave[2];
    DevBuf<float4> vtx0, vtx1, vtx2, shq_o, shq_d;
    DevBuf<uint32_t> vtx_j;
    Scratch scratch() const { return Scratch{vtx0.p, vtx1.p, vtx2.p, vtx_j.p, shq_o.p, shq_d.p}; }
    DevBuf<float4> stack;
    DevBuf<uint32_t> free_slots;
    DevBuf<Counters> counters;
    Counters *h_counters = nullptr;  // pinned
    void release() {
        wave[0].release(); wave[1].release(); stack.release(); free_slots.release();
        vtx0.release(); vtx1.release(); vtx2.release(); vtx_j.release(); shq_o.release(); shq_d.release();
        counters.release();
        retry.release();
        if (h_counters) (void)hipHostFree(h_counters);
        h_counters = nullptr;
        pool = 0;
    }
};

// Buffers shared by the wavefront pools of one scene.
struct SharedBufs {
    DevBuf<float> result;
    DevBuf<uint32_t> pixel_list, key_pixel, key_sample;
    DevBuf<uint32_t> culled_list, cull_count;  // pixel_list partitioned: [may hit | background only] (csrc/mcpt_cull.hip)
    DevBuf<uint8_t> cull_flags, cull_temp;
    DevBuf<int4> cand_tmp, cand_list;  // per pixel: the few primitives its rays can hit (aligned with culled_list)
    DevBuf<int32_t> key_channel;
    int pix_key[5] = {0, 0, 0, 0, 0};  // (W, H, tile, rank, nranks) of the pixel list currently in HBM
    uint32_t n_pix = 0;
    void release() {
        result.release(); pixel_list.release(); key_pixel.release(); key_sample.release(); key_channel.release();
        culled_list.release(); cull_count.release(); cull_flags.release(); cull_temp.release(); cand_tmp.release(); cand_list.release();
    }
};

// Environment knobs, read ONCE per scene in mcpt_scene_create: a render call never calls getenv.
struct Knobs {
    bool overlap = true;        // MCPT_OVERLAP=0: one stream instead of three
    bool queue_ahead = true;    // MCPT_QUEUE_AHEAD=0: wait for the counters before launching the chains
    bool timing = true;         // MCPT_TIMING=0: no per-kernel HIP events
    bool verbose = false;       // MCPT_RENDER_VERBOSE=1: a line per render call on stderr (pass size, pool, time of the allocations)
    int pools = 1;              // MCPT_POOLS=2: two pools on two host threads
    int drain_batch = 4;        // MCPT_DRAIN_BATCH: iterations per host sync in the drain tail
    uint64_t pool_min_work = 1ull << 20;  // MCPT_POOL_MIN_WORK: smallest pass (samples) that uses two pools
    // Grid caps, in workgroups per CU.  Every workgroup of k_trace_shadow computes the prefix sums of the queue's shards first, and the
    // LDS-resident flavours copy the scene into LDS first: with 128 / 64 per CU a workgroup strides over several chunks for one such
    // prologue and the hardware still balances uneven rays (round 3, A/B: cornell_rc 784^2 471 -> 490 Msamples/s, DEMO 1080p 873 -> 924,
    // k_trace_shadow -11 %, k_direct -14 %; chess within noise for 64..1024.  8 per CU, a persistent grid, was 30 % slower in round 1).
    uint32_t shadow_grid_per_cu = 128;    // MCPT_SHADOW_GRID_PER_CU: grid cap of k_trace_shadow
    uint32_t direct_grid_per_cu = 64;     // MCPT_DIRECT_GRID_PER_CU: grid cap of k_direct for LDS-resident scenes (0: none)
    // pure test hooks, compiled only into the checking build (-DMCPT_TEST_HOOKS, libmcpt_hip_check.so)
    uint32_t ring_start = 0;    // MCPT_RING_START: the free ring's counters start here (exercises the 2^32 wrap)
    int host_delay_us = 0;      // MCPT_HOST_DELAY_US: a slow host
    bool sky_cull = true;       // MCPT_SKY_CULL=0: trace the pixels that can only see the background too
    bool small_scene = true;    // MCPT_SMALL_SCENE=0: no LDS-resident flavour for scenes of a few KB
    uint64_t fake_free_mb = 0;  // MCPT_FAKE_FREE_MB: pretend that only this much device memory is free (exercises the pool shrink)
    void read() {
        auto off = [](const char *n) { const char *v = std::getenv(n); return v && v[0] == '0'; };
        overlap = !off("MCPT_OVERLAP");
        queue_ahead = !off("MCPT_QUEUE_AHEAD");
        timing = !off("MCPT_TIMING");
        verbose = std::getenv("MCPT_RENDER_VERBOSE") != nullptr;
        sky_cull = !off("MCPT_SKY_CULL");
        small_scene = !off("MCPT_SMALL_SCENE");
        const char *v;
        if ((v = std::getenv("MCPT_POOLS"))) pools = (v[0] == '2') ? 2 : 1;
        if ((v = std::getenv("MCPT_DRAIN_BATCH"))) drain_batch = std::max(1, std::atoi(v));
        if ((v = std::getenv("MCPT_POOL_MIN_WORK"))) pool_min_work = (uint64_t)std::max(1, std::atoi(v));
        if ((v = std::getenv("MCPT_SHADOW_GRID_PER_CU"))) shadow_grid_per_cu = (uint32_t)std::max(1, std::atoi(v));
        if ((v = std::getenv("MCPT_DIRECT_GRID_PER_CU"))) direct_grid_per_cu = (uint32_t)std::max(0, std::atoi(v));
#ifdef MCPT_TEST_HOOKS
        if ((v = std::getenv("MCPT_RING_START"))) ring_start = (uint32_t)std::strtoul(v, nullptr, 0);
        if ((v = std::getenv("MCPT_HOST_DELAY_US"))) host_delay_us = std::atoi(v);
        if ((v = std::getenv("MCPT_FAKE_FREE_MB"))) fake_free_mb = (uint64_t)std::max(1, std::atoi(v));
#endif
    }
};

enum KClass { K_CLOSEST = 0, K_SHADOW, K_SHADE, K_GENERATE, K_RESOLVE, K_DIRECT, K_NCLASS };

struct Timer {
    // Two banks of events: the host runs one iteration ahead of the GPU, so the events of iteration i are only known to be
    // complete once the read-back of iteration i+1 has arrived; iteration i+1 meanwhile records into the other bank.
    bool enabled = true;
    int bank = 0;
    std::vector<hipEvent_t> pool[2];
    struct Rec { int a, b, cls; };
    std::vector<Rec> recs[2];
    size_t used[2] = {0, 0};
    double ms[K_NCLASS] = {0, 0, 0, 0, 0, 0};
    uint64_t count[K_NCLASS] = {0, 0, 0, 0, 0, 0};
    int get() {
        if (used[bank] == pool[bank].size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return -1;
            pool[bank].push_back(e);
        }
        return (int)used[bank]++;
    }
    int begin(hipStream_t s) {
        if (!enabled) return -1;
        const int a = get();
        if (a >= 0) (void)hipEventRecord(pool[bank][a], s);
        return a;
    }
    void end(int a, int cls, hipStream_t s) {
        count[cls]++;
        if (!enabled || a < 0) return;
        const int b = get();
        if (b < 0) return;
        (void)hipEventRecord(pool[bank][b], s);
        recs[bank].push_back({a, b, cls});
    }
    void collect_bank(int k) {  // every event of bank k must have completed
        for (const Rec &r : recs[k]) {
            float t = 0.f;
            if (hipEventElapsedTime(&t, pool[k][r.a], pool[k][r.b]) == hipSuccess) ms[r.cls] += t;
        }
        recs[k].clear();
        used[k] = 0;
    }
    void collect() {  // call after a full stream sync
        collect_bank(0);
        collect_bank(1);
    }
    void reset() {
        for (int i = 0; i < K_NCLASS; ++i) { ms[i] = 0; count[i] = 0; }
        for (int k = 0; k < 2; ++k) { recs[k].clear(); used[k] = 0; }
        bank = 0;
    }
    void release() {
        for (int k = 0; k < 2; ++k) {
            for (hipEvent_t e : pool[k]) (void)hipEventDestroy(e);
            pool[k].clear();
        }
    }
};

}  // namespace

struct mcpt_scene {
    int device = 0;
    int device_sharers = 1;  // scenes of one group that live on this device (mcpt_group_create with a device listed several times)
    int32_t n_inner = 0;  // inner nodes of the traversal tree (0: the root is a leaf)
    Knobs knobs;
    mcpt_scene_info info{};
    DevBuf<Node> nodes;
    DevBuf<QNode> qnodes;
    DevBuf<TriGeom> tri_geom;
    DevBuf<TriShade> tri_shade;
    DevBuf<SphereRec> spheres;
    DevBuf<MaterialRec> mats;
    DevBuf<LightRec> lights;
    DevBuf<LightNode> light_nodes;
    DevBuf<LightTri> light_tris;
    DevBuf<InstRec> inst;
    DevBuf<float> env;
    DevBuf<unsigned long long> dbg;
    DevScene view{};
    SharedBufs shared;
    // A wavefront pool: its own path lists, queues, clamp stack, counters and streams.  With MCPT_POOLS=2 two pools
    // are driven by two host threads on disjoint halves of each pass, so that one pool's k_shade (and its host
    // round trip) overlaps the other pool's traversal kernels.
    struct PoolCtx {
        Workspace ws;
        Timer timer;
        hipStream_t main = nullptr;  // owned stream (pool 0 uses the caller's stream instead)
        // The three chains of one iteration (direct -> shadow, continuation rays, new primary rays) are
        // independent: they run on separate streams and are joined before the next k_shade.
        hipStream_t side[2] = {nullptr, nullptr};
        hipEvent_t join[2] = {nullptr, nullptr};
        hipEvent_t book = nullptr;  // main -> primary stream: the previous iteration's k_bookkeep has cleared the list counters
        hipEvent_t shaded = nullptr, readback = nullptr;  // k_shade done (-> closest stream); counters are in host memory
        uint64_t pushes = 0, overflow = 0;
        int rc = 0;
        std::string err;
    };
    static constexpr int kMaxPools = 2;
    PoolCtx pools[kMaxPools];
    int n_pools = 2;
    hipEvent_t fork = nullptr;
};
using PoolCtx = mcpt_scene::PoolCtx;

namespace {

hipError_t ensure_workspace(PoolCtx &ctx, uint32_t pool, int32_t n_dir, int32_t max_depth, bool retry_lists) {
    Workspace &w = ctx.ws;
    hipError_t e;
    const size_t n_rays = (size_t)pool + pool / 3 + 64;
    for (int k = 0; k < 2; ++k) {
        WaveBufs &b = w.wave[k];
        if ((e = b.rec0.alloc(pool)) != hipSuccess) return e;
        if ((e = b.rec1.alloc(pool)) != hipSuccess) return e;
        if ((e = b.ray_o.alloc(n_rays)) != hipSuccess) return e;
        if ((e = b.ray_d.alloc(n_rays)) != hipSuccess) return e;
        if ((e = b.hit.alloc(n_rays)) != hipSuccess) return e;
        if ((e = b.contrib.alloc((size_t)pool * n_dir)) != hipSuccess) return e;
        if ((e = b.fresh.alloc(pool / 3 + 64)) != hipSuccess) return e;
    }
    if ((e = w.vtx0.alloc(pool)) != hipSuccess) return e;
    if ((e = w.vtx1.alloc(pool)) != hipSuccess) return e;
    if ((e = w.vtx2.alloc(pool)) != hipSuccess) return e;
    if ((e = w.vtx_j.alloc(pool)) != hipSuccess) return e;
    const size_t shq = (size_t)kShadowShards * shadow_region((uint32_t)std::min<uint64_t>((uint64_t)pool * n_dir, 0xffffffffull));  // (sharded: Counters)
    if ((e = w.shq_o.alloc(shq)) != hipSuccess) return e;
    if ((e = w.shq_d.alloc(shq)) != hipSuccess) return e;
    if ((e = w.stack.alloc((size_t)pool * max_depth)) != hipSuccess) return e;
    uint64_t ring = 1;
    // free-slot ring: a power of two, so that the 32-bit head/tail counters may wrap, and at least twice the pool: the entries
    // k_primary has popped are read one iteration later (by k_shade) and must not be reached by the pushes made meanwhile
    // (free + pushed + popped <= 2 * pool)
    while (ring < 2 * (uint64_t)pool) ring <<= 1;
    if ((e = w.free_slots.alloc(ring)) != hipSuccess) return e;
    w.free_ring = (uint32_t)ring;
    w.ray_cap = (uint32_t)n_rays;
    if (retry_lists) {
        const uint32_t want[3] = {(uint32_t)n_rays, (uint32_t)std::min<uint64_t>((uint64_t)pool * n_dir, 0xffffffffull), pool / 3 + 64};
        if ((e = w.retry.alloc(want)) != hipSuccess) return e;
    }
    if ((e = w.counters.alloc(1)) != hipSuccess) return e;
    if (!w.h_counters && (e = hipHostMalloc((void **)&w.h_counters, sizeof(Counters))) != hipSuccess) return e;
    w.pool = pool;
    w.n_dir = n_dir;
    w.max_depth = max_depth;
    return hipSuccess;
}

int derive_max_depth(const mcpt_params &p) {
    if (p.max_depth > 0) return p.max_depth;
    const double rr = std::min(std::max((double)p.rr_rate, 1e-6), 0.999999);
    const int d = (int)std::ceil(std::log(1e-12) / std::log(rr));
    return std::min(std::max(d, 8), 8192);
}

CameraConst make_camera(const mcpt_camera &c) {
    CameraConst k;
    std::memset(&k, 0, sizeof k);
    k.width = c.width;
    k.height = c.height;
    k.use_dof = c.use_dof;
    // Renderer.cpp:13,25-26: deg2rad(deg) = deg * M_PI(float) / 180.0 returned as float; scale = tan(...)
    const float half = c.fov * 0.5f;
    const float rad = (float)((double)(half * 3.141592653589793f) / 180.0);
    k.scale = (float)std::tan((double)rad);
    k.aspect = c.width / (float)c.height;
    k.focal_distance = c.focal_distance;
    k.aperture_radius = c.aperture_radius;
    for (int i = 0; i < 3; ++i) k.eye[i] = c.position[i];
    for (int i = 0; i < 9; ++i) k.orient[i] = c.orientation[i];
    return k;
}

// Owned pixels in an order that keeps neighbouring list entries neighbouring on screen: tiles in
// row-major order, 8x8 blocks inside a tile.
void build_pixel_list(int W, int H, int tile, int rank, int nranks, std::vector<uint32_t> &out) {
    out.clear();
    if (tile <= 0) tile = 32;
    if (nranks < 1) nranks = 1;
    const int tx = (W + tile - 1) / tile, ty = (H + tile - 1) / tile;
    for (int tj = 0; tj < ty; ++tj)
        for (int ti = 0; ti < tx; ++ti) {
            if (((tj * tx + ti) % nranks) != rank) continue;
            const int x0 = ti * tile, y0 = tj * tile, x1 = std::min(W, x0 + tile), y1 = std::min(H, y0 + tile);
            for (int by = y0; by < y1; by += 8)
                for (int bx = x0; bx < x1; bx += 8)
                    for (int y = by; y < std::min(y1, by + 8); ++y)
                        for (int x = bx; x < std::min(x1, bx + 8); ++x) out.push_back((uint32_t)(y * W + x));
        }
}

struct LoopTotals {
    uint64_t iterations = 0, shaded = 0, closest = 0, shadow = 0, direct = 0;
};

// One pass of a render call: `n_work` camera samples whose results live in one half of the result buffer.
struct PassPlan {
    uint32_t first_work;  // first sample slot of the pass handled by this pool
    uint32_t n_work;      // sample slots handled by this pool
    int32_t s_pass, sample_offset;
};

// Where finished passes go.  acc == nullptr: the caller accumulates (single pass only).
struct AccumPlan {
    float *fb;
    float spp_total;
    uint32_t n_pix;
    const uint32_t *pixel_list;
    float *result[2];
};

// Runs the wavefront loop over a schedule of passes (mode 0) or over `plan[0].n_work` explicit paths (mode 1).
// Up to two passes are in flight: as soon as a pass has no samples left to issue, the next one starts filling the pool,
// so the drain tail of a pass overlaps useful work.  A pass is complete when its live-path counter is 0 (and all its
// samples were issued at least one iteration ago); passes are accumulated into the framebuffer strictly in order.
int run_wavefront(mcpt_scene *sc, PoolCtx &ctx, const RenderConst &C0, const CameraConst *cam, const std::vector<PassPlan> &plan,
                  const AccumPlan *acc, hipStream_t st, LoopTotals &tot) {
    Workspace &w = ctx.ws;
    Timer &T = ctx.timer;
    RenderConst C = C0;
    C.pool = w.pool;
    C.stack = w.stack.p;
    C.free_slots = w.free_slots.p;
    C.free_mask = w.free_ring - 1u;
    C.ray_cap = w.ray_cap;
    C.counters = w.counters.p;
    const uint32_t pool = w.pool;
    const int n_dir = C.n_dir;
    const int P = (int)plan.size();
    C.track_live = (acc && P > 1) ? 1 : 0;
    const Knobs &K = sc->knobs;
    launch_init_free(w.free_slots.p, w.counters.p, pool, K.ring_start, w.free_ring - 1u, st);
    if (ctx.side[1]) {  // fork: the side streams start after everything queued on `st` so far (counters, pixel list, framebuffer)
        HIP_TRY(hipEventRecord(ctx.book, st));
        for (int k = 0; k < 2; ++k) HIP_TRY(hipStreamWaitEvent(ctx.side[k], ctx.book, 0));
    }
    int cur = 0;
    uint32_t n_cur_max = 0;  // upper bound of the record count of wave[cur] (the exact count lives on the device)
    int issue_pass = 0, accum_next = 0;
    uint32_t issued = 0;
    uint32_t free_known = 0;  // free slots according to the last read-back (a lower bound of what k_primary may pop)
    bool have_counters = false;  // w.h_counters holds a read-back of THIS call
    // continuation rays / direct-lighting vertices per record, as observed in the last iteration: they size the grids of the
    // kernels that are queued before the host knows the true lengths (grid-stride kernels: any grid is correct)
    double cont_ratio = 1.0, direct_ratio = 1.0;
    double shadow_ratio = 1.0;      // shadow rays per light sample, as observed (sizes the grid of k_trace_shadow; any grid is correct)
    uint32_t n_direct_prev = 0, n_direct_prev2 = 0;  // lengths of the k_direct work lists of the two previous iterations
    bool n_cur_exact = false;  // n_cur_max is the true list length (false right after the prologue: an upper bound)
    long it = 0;
    std::vector<long> issue_done_iter(P, -1);
    hipStream_t s_close = ctx.side[0] ? ctx.side[0] : st, s_prim = ctx.side[1] ? ctx.side[1] : st;

    auto set_pass_consts = [&](int pi) {  // kernel constants of the pass occupying parity pi & 1
        const int32_t sp = plan[pi].s_pass;
        C.s_pass[pi & 1] = sp;
        C.sample_offset[pi & 1] = plan[pi].sample_offset;
        int sh = -1;
        if (sp > 0 && (sp & (sp - 1)) == 0)
            for (sh = 0; (1 << sh) < sp; ++sh) {}
        C.s_pass_shift[pi & 1] = sh;
    };
    // issues up to `room` new samples from the passes that may be in flight; returns how many
    auto issue = [&](Wave nx, int nxt, uint32_t room) -> uint32_t {
        uint32_t total = 0;
        while (room > 0 && issue_pass < P && (!acc || issue_pass < accum_next + 2)) {
            if (issued == 0) set_pass_consts(issue_pass);
            const uint32_t g = std::min<uint32_t>(room, plan[issue_pass].n_work - issued);
            if (g > 0) {
                int ev = T.begin(s_prim);
                launch_primary(sc->view, *cam, C, nx, nxt, issue_pass & 1, plan[issue_pass].first_work + issued, g, w.retry.list(2), s_prim);
                T.end(ev, K_GENERATE, s_prim);
                tot.closest += g;
            }
            issued += g;
            room -= g;
            total += g;
            if (issued == plan[issue_pass].n_work) {
                issue_done_iter[issue_pass] = it;
                issue_pass++;
                issued = 0;
            } else {
                break;
            }
        }
        return total;
    };
    // accumulates, in order, every pass that is complete according to the counters just read back
    auto accumulate_done = [&](hipStream_t s) {
        while (acc && have_counters && accum_next < issue_pass && issue_done_iter[accum_next] < it &&
               (C.track_live ? w.h_counters->live[accum_next & 1].v == 0 : (n_cur_max == 0 && issue_pass >= P))) {
            int ev = T.begin(s);
            launch_accumulate(acc->result[accum_next & 1], acc->pixel_list, acc->n_pix, plan[accum_next].s_pass, acc->spp_total, acc->fb, s);
            T.end(ev, K_RESOLVE, s);
            accum_next++;
        }
    };

    // prologue: fill the pool
    {
        Wave nx = w.wave[cur].view();
        if (C.mode == 0) {
            n_cur_max = 3 * issue(nx, cur, pool / 3);
        } else {
            // explicit rays were uploaded into wave[cur].ray_o/ray_d by the caller
            const uint32_t n_work = plan[0].n_work;
            launch_generate_explicit(C, nx, cur, n_work, st);
            issue_pass = P;
            n_cur_max = n_work;
            int ev = T.begin(st);
            launch_trace_closest(sc->view, n_work, nullptr, nx.ray_o, nx.ray_d, nx.hit, w.retry.list(0), st);
            T.end(ev, K_CLOSEST, st);
            tot.closest += n_work;
        }
        for (int k = 0; k < 2; ++k) {  // join the side streams before the first k_shade
            if (!ctx.side[k]) continue;
            HIP_TRY(hipEventRecord(ctx.join[k], ctx.side[k]));
            HIP_TRY(hipStreamWaitEvent(st, ctx.join[k], 0));
        }
    }

    const bool queue_ahead = K.queue_ahead;
    const int host_delay_us = K.host_delay_us;
    const int drain_batch = K.drain_batch;  // iterations queued per host sync once no samples are left to issue
    while (n_cur_max > 0 || issue_pass < P || (acc && accum_next < P)) {
        ++it;
        // (big lists keep the three-stream schedule with exact launch sizes: over-sized grids only pay off when small)
        const bool draining = issue_pass >= P && n_cur_max > 0 && n_cur_max <= (2u << 20);
        if (draining && drain_batch > 1) {
            // Drain phase: no regeneration, so list lengths only shrink.  Several iterations are queued back to back
            // on one stream with the last known length as the grid bound (every kernel reads the true lengths on the
            // device); the host looks at the counters once per batch.
            for (int k = 0; k < drain_batch; ++k) {
                const int nxt = cur ^ 1;
                Wave cw = w.wave[cur].view(), nx = w.wave[nxt].view();
                int ev = T.begin(st);
                launch_shade(sc->view, C, cw, nx, w.scratch(), cur, n_cur_max, st);
                T.end(ev, K_SHADE, st);
                launch_bookkeep(w.counters.p, cur, false, 0, 0, 0, st);
                ev = T.begin(st);
                launch_direct(sc->view, C, nx, w.scratch(), nxt, n_cur_max, K.direct_grid_per_cu, st);
                T.end(ev, K_DIRECT, st);
                if (C.enable_shadow) {
                    ev = T.begin(st);
                    launch_trace_shadow(sc->view, w.counters.p, nxt, n_cur_max * (uint32_t)n_dir, (uint32_t)C.pool * (uint32_t)n_dir, w.scratch(), nx.contrib, K.shadow_grid_per_cu, w.retry.list(1), st);
                    T.end(ev, K_SHADOW, st);
                }
                ev = T.begin(st);
                launch_trace_closest(sc->view, n_cur_max, &w.counters.p->n_rays[nxt].v, nx.ray_o, nx.ray_d, nx.hit, w.retry.list(0), st);
                T.end(ev, K_CLOSEST, st);
                cur = nxt;
            }
            HIP_TRY(hipMemcpyAsync(w.h_counters, w.counters.p, kCountersHeadBytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            T.collect();
            have_counters = true;
            n_cur_max = w.h_counters->n_paths[cur].v + 3u * w.h_counters->n_prays[cur].v;
            n_cur_exact = true;
            accumulate_done(st);
            continue;
        }
        const int nxt = cur ^ 1;
        Wave cw = w.wave[cur].view(), nx = w.wave[nxt].view();
        // (the counters indexed `nxt` were cleared by the previous iteration's k_bookkeep, or by k_init_free)
        // New samples are generated CONCURRENTLY with k_shade (memory-latency-bound; the primary kernel is issue-bound):
        // both append to list `nxt`.  How many fit is decided from the previous read-back: k_shade only ever adds free
        // slots and never lengthens the list, so the slots and the list room known then are still there.
        accumulate_done(s_prim);  // frees the result half that the pass after next needs; ordered before its k_primary
        if (C.mode == 0 && n_cur_max < pool && free_known >= 3u) {
            if (ctx.side[1]) HIP_TRY(hipStreamWaitEvent(s_prim, ctx.book, 0));
            issue(nx, nxt, std::min<uint32_t>((pool - n_cur_max) / 3, free_known / 3));
        }
        int ev = -1;
        if (n_cur_max > 0) {
            ev = T.begin(st);
            launch_shade(sc->view, C, cw, nx, w.scratch(), cur, n_cur_max, st);
            T.end(ev, K_SHADE, st);
        }
        if (ctx.side[0]) HIP_TRY(hipEventRecord(ctx.shaded, st));
        if (ctx.side[1]) {  // the read-back waits for the new samples (and for a k_accumulate issued above) as well
            HIP_TRY(hipEventRecord(ctx.join[1], s_prim));
            HIP_TRY(hipStreamWaitEvent(st, ctx.join[1], 0));
        }
        HIP_TRY(hipMemcpyAsync(w.h_counters, w.counters.p, kCountersHeadBytes, hipMemcpyDeviceToHost, st));  // (not the sharded counters)
        HIP_TRY(hipEventRecord(ctx.readback, st));

        // queue_ahead: the rest of the iteration is queued BEFORE the host looks at the counters.  Every kernel reads the
        // true queue lengths on the device and strides over its queue, so the grids only need estimates (the ratios seen in
        // the previous iteration).  The GPU then never waits for the host round trip: while the host sizes the next
        // iteration, the three chains below are running.  Otherwise the host waits first and launches exact grids.
        uint32_t n_cont = 0, n_direct = 0;
        auto wait_counters = [&]() -> int {
            HIP_TRY(hipEventSynchronize(ctx.readback));
            if (host_delay_us > 0) usleep((useconds_t)host_delay_us);  // test hook: a slow host
            T.collect_bank(T.bank ^ 1);  // the previous iteration's kernels all finished before this iteration's k_shade
            have_counters = true;
            n_cont = w.h_counters->n_rays[nxt].v;
            n_direct = w.h_counters->n_direct[nxt].v;
            return MCPT_OK;
        };
        if (!queue_ahead) {
            const int rc = wait_counters();
            if (rc != MCPT_OK) return rc;
        }
        // (+25 %, and never fewer than 2048 workgroups' worth of lanes: a grid that is too small still works, but loses balance)
        const uint32_t grid_floor = std::min<uint32_t>(n_cur_max, 2048u * 256u);
        const uint32_t grid_cont = !queue_ahead ? n_cont : std::max<uint32_t>(grid_floor, std::min<uint32_t>(n_cur_max, (uint32_t)(1.25 * cont_ratio * n_cur_max) + 4096u));
        const uint32_t grid_direct = !queue_ahead ? n_direct : std::max<uint32_t>(grid_floor, std::min<uint32_t>(n_cur_max, (uint32_t)(1.25 * direct_ratio * n_cur_max) + 4096u));
        if (grid_cont > 0) {
            if (ctx.side[0]) HIP_TRY(hipStreamWaitEvent(s_close, ctx.shaded, 0));
            ev = T.begin(s_close);
            launch_trace_closest(sc->view, grid_cont, queue_ahead ? &w.counters.p->n_rays[nxt].v : nullptr, nx.ray_o, nx.ray_d, nx.hit, w.retry.list(0), s_close);
            T.end(ev, K_CLOSEST, s_close);
        }
        launch_bookkeep(w.counters.p, cur, false, 0, 0, 0, st);  // totals += lengths; list `cur` is consumed
        if (ctx.side[1]) HIP_TRY(hipEventRecord(ctx.book, st));
        if (grid_direct > 0) {
            ev = T.begin(st);
            launch_direct(sc->view, C, nx, w.scratch(), nxt, grid_direct, K.direct_grid_per_cu, st);
            T.end(ev, K_DIRECT, st);
            if (C.enable_shadow) {
                ev = T.begin(st);
                // (every workgroup of k_trace_shadow pays for the prefix sums of the queue's shards before it knows whether it has work:
                // the grid follows the observed number of shadow rays per light sample instead of covering every light sample)
                const uint32_t n_samples_max = grid_direct * (uint32_t)n_dir;
                const uint32_t grid_shadow = std::max<uint32_t>(std::min<uint32_t>(n_samples_max, 1024u * 256u),
                                                                std::min<uint32_t>(n_samples_max, (uint32_t)(1.5 * shadow_ratio * n_samples_max) + 4096u));
                launch_trace_shadow(sc->view, w.counters.p, nxt, grid_shadow, (uint32_t)C.pool * (uint32_t)n_dir, w.scratch(), nx.contrib, K.shadow_grid_per_cu, w.retry.list(1), st);
                T.end(ev, K_SHADOW, st);
            }
        }
        // join
        if (ctx.side[0]) {
            HIP_TRY(hipEventRecord(ctx.join[0], ctx.side[0]));
            HIP_TRY(hipStreamWaitEvent(st, ctx.join[0], 0));
        }
        if (queue_ahead) {
            const int rc = wait_counters();
            if (rc != MCPT_OK) return rc;
        }
        T.bank ^= 1;
        const uint32_t n_next = w.h_counters->n_paths[nxt].v + 3u * w.h_counters->n_prays[nxt].v;  // records + three lanes per new sample
        free_known = w.h_counters->free_tail.v - w.h_counters->free_head.v;
        if (n_cur_max > 0) {
            if (n_cur_exact) {  // (an upper bound in the denominator would under-size the next grids)
                cont_ratio = (double)n_cont / n_cur_max;
                direct_ratio = (double)n_direct / n_cur_max;
            }
            // (last_shadow: the queue k_bookkeep cleared before this read-back, i.e. the one k_direct filled two iterations ago)
            if (n_direct_prev2 > 0) shadow_ratio = std::min(1.0, (double)w.h_counters->last_shadow / ((double)n_direct_prev2 * n_dir));
        }
        n_direct_prev2 = n_direct_prev;
        n_direct_prev = n_direct;
        n_cur_max = n_next;
        n_cur_exact = true;
        cur = nxt;
    }
    // the last shadow queue was consumed after the last k_bookkeep: fold it into the totals
    HIP_TRY(hipMemcpyAsync(w.h_counters, w.counters.p, sizeof(Counters), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    T.collect();
    const Counters &hc = *w.h_counters;
    ctx.pushes += hc.tot_pushes + hc.pushes.v;
    ctx.overflow += hc.overflow.v;
    tot.iterations += hc.tot_iterations;
    tot.shaded += hc.tot_shaded + hc.tot_ended + hc.ended.v;
    tot.direct += hc.tot_direct;
    tot.closest += hc.tot_cont;
    tot.shadow += hc.tot_shadow;
    for (uint32_t k = 0; k < kShadowShards; ++k) tot.shadow += hc.n_shadow[0][k].v + hc.n_shadow[1][k].v + hc.n_shadow_w[0][k].v + hc.n_shadow_w[1][k].v;
    return MCPT_OK;
}

// After a failed run_wavefront the side streams may still hold kernels that use the workspace: wait for them before the
// caller sees the error (and possibly frees buffers).  The error text of the failure is kept.
int drained(int rc) {
    if (rc != MCPT_OK) {
        const std::string keep = g_err;
        (void)hipDeviceSynchronize();
        (void)hipGetLastError();
        g_err = keep;
    }
    return rc;
}

// Device bytes one pool path costs in ensure_workspace (two waves + scratch + clamp stack + free ring).
uint64_t bytes_per_pool_path(int n_dir, int max_depth) {
    const double rays = 1.0 + 1.0 / 3.0;
    const double wave = 16 + 16 + rays * 48 + 4.0 * n_dir + 8.0 / 3.0;
    const double scratch = 3 * 16 + 4 + 32.0 * n_dir;
    const double retrace = 4.0 * (rays + n_dir + 1.0 / 3.0);  // (lists of the retry flavour; counted whether or not the tree needs them)
    return (uint64_t)(2 * wave + scratch + retrace + 16.0 * max_depth + 16.0);
}

int render_impl(mcpt_scene *sc, const mcpt_camera *cam, const mcpt_params *pp, float *fb_dev, hipStream_t st,
                mcpt_stats *stats) {
    if (!sc || !cam || !pp || !fb_dev) return fail(MCPT_ERR_ARG, "mcpt_render: null argument");
    const mcpt_params &p = *pp;
    if (cam->width <= 0 || cam->height <= 0 || p.spp <= 0 || p.n_dir_sample <= 0 || !(p.rr_rate > 0.f))
        return fail(MCPT_ERR_ARG, "mcpt_render: width/height/spp/n_dir_sample/rr_rate must be positive");
    if ((uint64_t)cam->width * cam->height > 0x7fffffffull) return fail(MCPT_ERR_ARG, "mcpt_render: frame too large");
    HIP_TRY(hipSetDevice(sc->device));
    (void)hipGetLastError();  // an earlier, already reported failure of this thread must not be taken for one of this call
    const auto t0 = std::chrono::steady_clock::now();
    const int W = cam->width, H = cam->height;

    // owned pixels: rebuilt and uploaded only when the partition changes (progressive calls reuse it)
    SharedBufs &sh = sc->shared;
    const int pk[5] = {W, H, p.tile_size, p.nranks > 1 ? p.rank : 0, p.nranks > 1 ? p.nranks : 1};
    if (std::memcmp(pk, sh.pix_key, sizeof pk) != 0 || !sh.pixel_list.p) {
        std::vector<uint32_t> pix;
        build_pixel_list(W, H, pk[2], pk[3], pk[4], pix);
        sh.n_pix = (uint32_t)pix.size();
        HIP_TRY(sh.pixel_list.alloc(std::max<size_t>(pix.size(), 1)));
        if (!pix.empty()) HIP_TRY(hipMemcpy(sh.pixel_list.p, pix.data(), pix.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
        std::memcpy(sh.pix_key, pk, sizeof pk);
    }
    const uint32_t n_pix_owned = sh.n_pix;
    const CameraConst cc = make_camera(*cam);
    const float spp_total = (float)(p.spp_total > 0 ? p.spp_total : p.spp);
    if (!p.accumulate) HIP_TRY(hipMemsetAsync(fb_dev, 0, (size_t)W * H * 3 * sizeof(float), st));

    // Pixels that can only see the background (no environment map: every sample returns the same constant) are finished here,
    // without a ray; the wavefront loop below runs over the others.  csrc/mcpt_cull.hip has the conservative bound.
    uint32_t n_pix = n_pix_owned;
    const uint32_t *pixel_list = sh.pixel_list.p;
    const int4 *pixel_cand = nullptr;
    if (sc->knobs.sky_cull && sc->view.env_w <= 0 && n_pix_owned > 0) {
        const size_t tb = cull_temp_bytes(n_pix_owned);
        HIP_TRY(sh.culled_list.alloc(n_pix_owned));
        HIP_TRY(sh.cull_flags.alloc(n_pix_owned));
        HIP_TRY(sh.cull_temp.alloc(tb));
        HIP_TRY(sh.cull_count.alloc(1));
        HIP_TRY(sh.cand_tmp.alloc(n_pix_owned));
        HIP_TRY(sh.cand_list.alloc(n_pix_owned));
        uint32_t n_trace = n_pix_owned + 1;  // (left untouched when the camera is outside what the bound covers)
        HIP_TRY(cull_sky_pixels(sc->view, cc, sh.pixel_list.p, n_pix_owned, sh.culled_list.p, sh.cull_flags.p, sh.cand_tmp.p, sh.cand_list.p, sh.cull_temp.p,
                                tb, sh.cull_count.p, &n_trace, st));
        if (n_trace <= n_pix_owned) {  // classified: the traced pixels come first, in their original order, with their candidate lists
            launch_sky_fill(sh.culled_list.p + n_trace, n_pix_owned - n_trace, sc->view.background, p.spp, spp_total, fb_dev, st);
            n_pix = n_trace;
            pixel_list = sh.culled_list.p;
            // (candidate lists skip the float box tests of a primitive's ancestors: a ray that grazes a box face within rounding is a hit
            // through the list and a miss through the tree.  With the reference's own topology the kernels promise the reference's box
            // semantics exactly, so primary rays walk the tree there; the sky cull itself stays.)
            pixel_cand = sc->info.builder == 1 ? nullptr : sh.cand_list.p;
        }
    }

    const int max_depth = derive_max_depth(p);
    // default: the smallest pool within 1 % of the best rate.  Measured on the chess frame (round 3, A/B on one box): 40 Mi paths 5031-5045,
    // 48 Mi 5061, 60 Mi 5052-5066 Msamples/s (round 2: 28 Mi 4467, 40 Mi 4557, 60 Mi 4617, 80 Mi 4605); 40 Mi paths are 38 GB of workspace
    uint64_t pool64 = p.pool_paths > 0 ? (uint64_t)p.pool_paths : (40ull << 20);
    pool64 = std::max<uint64_t>(pool64, 3 * 256);
    // The pass size: what the caller asks for, or (spp_per_pass 0) one chosen here.  Only two passes are in flight (one result half each), and
    // the tail of a pass -- a few long paths -- holds its half for some twenty iterations: a pass has to carry many pools' worth of samples
    // or the pool runs half empty between passes.  Measured (tools/pass_size.py, chess 1080p spp 2048; the pool holds 13.4 M samples): 1.2 M
    // traced pixels x 32 spp (3x the pool) 4263 Msamples/s, x 64 4684, x 128 4907, x 256 4997, x 512 5010, x 1024 / 2048 4880 (the result buffer
    // grows with the pass); one rank of eight (0.15 M pixels): 32 spp 2317, 256 4282, 512 4610, 1024 4732, 2048 4769.  Chosen: the power
    // of two that makes a pass at least 16 pools' worth of samples (256 spp for the 1080p chess frame: 5.5 GB of result buffers; 2048 for an eighth
    // of it), between 32 spp and the call's own spp, within a quarter of the free memory.
    // (device memory this call may use: what is free now plus what this scene's workspace and result buffer already hold; other tenants
    // of the GPU, a second scene, replicas of one group that share the device -- a rehearsal on a one-GPU box -- each take their share)
    uint64_t have = 0;
    bool have_mem = false;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            if (sc->knobs.fake_free_mb) free_b = std::min<size_t>(free_b, (size_t)sc->knobs.fake_free_mb << 20);  // (test hook)
            uint64_t held = 0;
            for (int k = 0; k < mcpt_scene::kMaxPools; ++k) held += (uint64_t)sc->pools[k].ws.pool * bytes_per_pool_path(sc->pools[k].ws.n_dir, sc->pools[k].ws.max_depth);
            have = ((uint64_t)free_b + held + sc->shared.result.bytes()) / (uint64_t)std::max(1, sc->device_sharers);
            have_mem = true;
        }
    }
    int s_pass_req = p.spp_per_pass;  // what the result buffer is sized for
    if (s_pass_req <= 0) {
        const uint64_t want = 16ull * (pool64 / 3) / std::max<uint32_t>(n_pix, 1u) + 1ull;
        int cap = 32;
        while (cap < p.spp && cap < (1 << 20)) cap *= 2;  // (no larger than the call needs: the buffer of a short call stays small)
        s_pass_req = 32;
        while ((uint64_t)s_pass_req < want && s_pass_req < cap) s_pass_req *= 2;
        if (have_mem) while (s_pass_req > 32 && (uint64_t)n_pix * s_pass_req * 3ull * 4ull * 2ull > have / 4) s_pass_req /= 2;
    }
    while ((uint64_t)n_pix * s_pass_req * 3ull > 0xfffffff0ull && s_pass_req > 1) s_pass_req /= 2;
    const int s_pass = std::min(s_pass_req, p.spp);
    // keep the clamp stack within 48 GiB
    while (pool64 * (uint64_t)max_depth * 16ull > (48ull << 30) && pool64 > 3 * 4096) pool64 /= 2;
    // light-sample indices (record * n_dir + k) are 32-bit
    while (pool64 * (uint64_t)p.n_dir_sample > (1ull << 31) && pool64 > 3 * 4096) pool64 /= 2;
    pool64 = std::min<uint64_t>(pool64, std::max<uint64_t>(3ull * n_pix * (uint64_t)s_pass, 3 * 256));
    // ... and the workspace within 80 % of that memory: a smaller pool is slower, never wrong
    if (have_mem) {
        const uint64_t result_b = (uint64_t)n_pix * s_pass_req * 3ull * 4ull * 2ull;
        const uint64_t budget = have * 8 / 10 > result_b ? have * 8 / 10 - result_b : 0;
        const uint64_t per = bytes_per_pool_path(p.n_dir_sample, max_depth);
        while (pool64 * per > budget && pool64 > 3 * 4096) pool64 /= 2;
    }
    // two pools (each half the paths) once a pass is big enough to keep both busy
    int n_pools = sc->n_pools;
    const uint64_t min_work = sc->knobs.pool_min_work;  // samples per pass below which one pool is used
    if ((uint64_t)n_pix * s_pass < min_work || pool64 / 2 < 3 * 256) n_pools = 1;
    const uint32_t pool = (uint32_t)(pool64 / n_pools / 3 * 3);

    if (n_pix_owned == 0) {
        HIP_TRY(hipStreamSynchronize(st));
        if (stats) std::memset(stats, 0, sizeof *stats);
        return MCPT_OK;
    }
    // (n_pix == 0 with owned pixels: every one of them was culled; the loop below then has no samples to issue and falls through)
    const auto t_alloc0 = std::chrono::steady_clock::now();
    for (int k = 0; k < n_pools; ++k) HIP_TRY(ensure_workspace(sc->pools[k], pool, p.n_dir_sample, max_depth, stack_uses_retry(sc->view.height)));
    // two halves: a pass accumulates from one while the next pass fills the other (one half with a single pass)
    const size_t half_floats = (size_t)n_pix * s_pass * 3;
    const bool two_halves = n_pools == 1 && p.spp > s_pass;
    // both halves are allocated, for the REQUESTED pass size, even when this call needs less: a later call with more or longer passes
    // (a warm-up followed by the real frame) must not pay a multi-GB hipFree + hipMalloc
    HIP_TRY(sh.result.alloc((size_t)n_pix * s_pass_req * 3 * (n_pools == 1 ? 2 : 1)));
    if (sc->knobs.verbose)
        std::fprintf(stderr, "[mcpt render] %u traced pixels, pass %d spp, pool %u paths; set-up before the allocations %.1f ms, workspace + result buffers %.1f ms\n", n_pix, s_pass,
                     pool, std::chrono::duration<double, std::milli>(t_alloc0 - t0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc0).count());

    RenderConst C;
    std::memset(&C, 0, sizeof C);
    C.rr_rate = p.rr_rate;
    C.inv_rr = 1 / p.rr_rate;  // Scene.hpp:112
    C.n_dir = p.n_dir_sample;
    C.enable_shadow = p.enable_shadow;
    C.seed = p.seed;
    C.mode = 0;
    C.pixel_list = pixel_list;
    C.pixel_cand = pixel_cand;
    C.max_depth = max_depth;
    C.result[0] = sh.result.p;
    C.result[1] = two_halves ? sh.result.p + half_floats : sh.result.p;
    for (int k = 0; k < n_pools; ++k) {
        sc->pools[k].timer.reset();
        sc->pools[k].timer.enabled = sc->knobs.timing;
        sc->pools[k].pushes = sc->pools[k].overflow = 0;
    }
    LoopTotals tot[mcpt_scene::kMaxPools];
    if (n_pools == 1) {
        // one pool: all passes of the call go through one pipelined schedule
        std::vector<PassPlan> plan;
        for (int k0 = 0; k0 < p.spp; k0 += s_pass) {
            const int s_now = std::min(s_pass, p.spp - k0);
            plan.push_back(PassPlan{0u, n_pix * (uint32_t)s_now, s_now, p.sample_offset + k0});
        }
        AccumPlan acc{fb_dev, spp_total, n_pix, pixel_list, {C.result[0], C.result[1]}};
        const int rc = drained(run_wavefront(sc, sc->pools[0], C, &cc, plan, &acc, st, tot[0]));
        if (rc != MCPT_OK) return rc;
    } else {
        for (int k0 = 0; k0 < p.spp; k0 += s_pass) {
            const int s_now = std::min(s_pass, p.spp - k0);
            const uint32_t n_work = n_pix * (uint32_t)s_now;
            // pool 1 (own stream, own host thread) takes the second half of the pass; it starts after the
            // framebuffer clear / pixel-list upload / previous accumulate queued on the caller's stream
            const uint32_t half = n_work / 2;
            HIP_TRY(hipEventRecord(sc->fork, st));
            PoolCtx &c1 = sc->pools[1];
            HIP_TRY(hipStreamWaitEvent(c1.main, sc->fork, 0));
            c1.rc = MCPT_OK;
            const std::vector<PassPlan> plan0{PassPlan{0u, half, s_now, p.sample_offset + k0}};
            const std::vector<PassPlan> plan1{PassPlan{half, n_work - half, s_now, p.sample_offset + k0}};
            std::thread worker([&]() {
                if (hipSetDevice(sc->device) != hipSuccess) {
                    c1.rc = MCPT_ERR_HIP;
                    c1.err = "hipSetDevice failed in the pool thread";
                    return;
                }
                c1.rc = run_wavefront(sc, c1, C, &cc, plan1, nullptr, c1.main, tot[1]);
                if (c1.rc != MCPT_OK) c1.err = g_err;
            });
            const int rc0 = run_wavefront(sc, sc->pools[0], C, &cc, plan0, nullptr, st, tot[0]);
            worker.join();  // run_wavefront ends with a stream synchronise: both halves are complete here
            if (rc0 != MCPT_OK) return drained(rc0);
            if (c1.rc != MCPT_OK) return drained(fail(c1.rc, c1.err));
            Timer &T0 = sc->pools[0].timer;
            int ev = T0.begin(st);
            launch_accumulate(C.result[0], pixel_list, n_pix, s_now, spp_total, fb_dev, st);
            T0.end(ev, K_RESOLVE, st);
        }
    }
    HIP_TRY(hipStreamSynchronize(st));
    sc->pools[0].timer.collect();
    HIP_TRY(hipGetLastError());
    uint64_t pushes = 0, overflow = 0;
    LoopTotals sum;
    double ms[K_NCLASS] = {0, 0, 0, 0, 0, 0};
    uint64_t cnt[K_NCLASS] = {0, 0, 0, 0, 0, 0};
    for (int k = 0; k < n_pools; ++k) {
        pushes += sc->pools[k].pushes;
        overflow += sc->pools[k].overflow;
        sum.iterations += tot[k].iterations;
        sum.shaded += tot[k].shaded;
        sum.closest += tot[k].closest;
        sum.shadow += tot[k].shadow;
        sum.direct += tot[k].direct;
        for (int c = 0; c < K_NCLASS; ++c) {
            ms[c] += sc->pools[k].timer.ms[c];
            cnt[c] += sc->pools[k].timer.count[c];
        }
    }
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        // (culled pixels count like traced ones: the reference runs one camera ray and three castRay invocations, each with one
        // Scene::intersect, for every sample of them too)
        stats->samples = (uint64_t)n_pix_owned * p.spp;
        stats->paths = 3 * stats->samples;
        stats->vertices = stats->paths + pushes;
        stats->shaded = sum.shaded;
        stats->closest_rays = sum.closest;
        stats->shadow_rays = sum.shadow;
        stats->direct_vertices = sum.direct;
        // Scene::intersect calls of the reference: one per castRay invocation (Scene.cpp:87), n_dir per shaded
        // vertex (Scene.cpp:73), one look-ahead per vertex that survives roulette (Scene.cpp:134,161).
        const uint64_t cont = sum.closest - (uint64_t)n_pix * p.spp;  // closest-hit rays beyond the primary rays actually traced
        stats->ref_scene_rays = stats->vertices + (uint64_t)p.n_dir_sample * sum.shaded + cont;
        stats->iterations = sum.iterations;
        stats->overflow_paths = overflow;
        stats->ms_trace_closest = ms[K_CLOSEST];
        stats->ms_trace_shadow = ms[K_SHADOW];
        stats->ms_shade = ms[K_SHADE];
        stats->ms_generate = ms[K_GENERATE];
        stats->ms_resolve = ms[K_RESOLVE];
        stats->ms_direct = ms[K_DIRECT];
        stats->n_direct = cnt[K_DIRECT];
        stats->n_trace_closest = cnt[K_CLOSEST];
        stats->n_trace_shadow = cnt[K_SHADOW];
        stats->n_shade = cnt[K_SHADE];
        stats->n_generate = cnt[K_GENERATE];
        stats->n_resolve = cnt[K_RESOLVE];
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (overflow) return fail(MCPT_ERR_OVERFLOW, "some paths outran the clamp stack (raise params.max_depth)");
    return MCPT_OK;
}

}  // namespace

extern "C" {

const char *mcpt_last_error(void) { return g_err.c_str(); }
const char *mcpt_version(void) {
#if defined(MCPT_TEST_HOOKS) || defined(MCPT_CHECK_DIRECT_SKIP) || defined(MCPT_TRAVERSAL_STATS)
    return "mcpt-hip 0.2 (gfx950) checking build";
#else
    return "mcpt-hip 0.2 (gfx950)";
#endif
}

int mcpt_scene_create(const mcpt_scene_desc *desc, int device, mcpt_scene **out) { return mcpt_scene_create_ex(desc, device, nullptr, out); }

}  // extern "C"

namespace mcpt {

// First use of a device by this process: context creation and the load of this library's code objects cost 100-150 ms (round 2's
// `upload_ms` of 129-155 ms for a 6.8 KB scene was exactly this, not the copies).  It does not depend on the scene, so it runs on a
// helper thread while the calling thread flattens the scene and builds its tree, and it is reported on its own (mcpt_scene_info).
double warm_up_device(int device) {
    const auto t0 = std::chrono::steady_clock::now();
    if (hipSetDevice(device) != hipSuccess) return 0.0;
    uint32_t *p = nullptr;
    if (hipMalloc((void **)&p, 256) == hipSuccess) {
        launch_add_frame(reinterpret_cast<float *>(p), reinterpret_cast<float *>(p), 0u, nullptr);  // (n = 0: no launch; keeps the symbol referenced)
        (void)hipMemset(p, 0, 256);
        launch_mask_unowned(reinterpret_cast<float *>(p), 1, 1, 1, 0, 1, nullptr);  // one tiny kernel of this library: forces its code objects in
        (void)hipDeviceSynchronize();
        (void)hipFree(p);
    }
    (void)hipGetLastError();
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
}

int build_scene_host(const mcpt_scene_desc *desc, const mcpt_build_options *options, HostBuild &hb) {
    if (!desc) return fail(MCPT_ERR_ARG, "mcpt_scene_create: null argument");
    const char *err = "";
    const auto t_build = std::chrono::steady_clock::now();
    hb.choice = resolve_build_choice(options);
    // (a single primitive has no inner node: nothing for the device builder to do)
    if ((hb.choice.builder == MCPT_BUILD_GPU_LBVH || hb.choice.builder == MCPT_BUILD_GPU_PLOC) && desc->objects) {
        int64_t n_prim = desc->n_triangles;
        for (int i = 0; i < desc->n_objects; ++i) n_prim += desc->objects[i].kind == MCPT_OBJ_SPHERE ? 1 : 0;
        if (n_prim < 2) hb.choice.builder = MCPT_BUILD_SAH;
    }
    const int rc = build_host_scene(*desc, hb.hs, &err, hb.choice);
    if (rc != MCPT_OK) return fail(rc, std::string("mcpt_scene_create: ") + err);
    hb.build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_build).count();
    return MCPT_OK;
}

}  // namespace mcpt

extern "C" {

int mcpt_scene_create_ex(const mcpt_scene_desc *desc, int device, const mcpt_build_options *options, mcpt_scene **out) {
    if (!desc || !out) return fail(MCPT_ERR_ARG, "mcpt_scene_create: null argument");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MCPT_ERR_HIP, "mcpt_scene_create: no HIP device available (this library has no CPU fallback)");
    if (device < 0) {
        if (hipGetDevice(&device) != hipSuccess) device = 0;
    }
    if (device >= ndev) return fail(MCPT_ERR_ARG, "mcpt_scene_create: device index out of range");
    double init_ms = 0.0;
    std::thread warm([&]() { init_ms = warm_up_device(device); });  // beside the host build
    HostBuild hb;
    const int rc = build_scene_host(desc, options, hb);
    warm.join();
    if (rc != MCPT_OK) return rc;
    hb.init_ms = init_ms;
    return upload_scene(desc, hb, device, out);
}

}  // extern "C"

namespace mcpt {

// The device half of mcpt_scene_create: copies a flattened scene to `device` (and, for MCPT_BUILD_GPU_LBVH, builds the tree there).
// mcpt_group_create builds the host scene ONCE and calls this from one thread per device.
int upload_scene(const mcpt_scene_desc *desc, HostBuild &hb, int device, mcpt_scene **out) {
    HostScene &hs = hb.hs;
    const BuildChoice &choice = hb.choice;
    HIP_TRY(hipSetDevice(device));
    const auto t_upload = std::chrono::steady_clock::now();

    mcpt_scene *sc = new (std::nothrow) mcpt_scene();
    if (!sc) return fail(MCPT_ERR_OOM, "mcpt_scene_create: host allocation failed");
    sc->device = device;
    hipError_t e = hipSuccess;
    sc->knobs.read();
    // One pool by default: two pools measured +1..2 % at equal total size (2772 vs 2748 Msamples/s), within noise.
    sc->n_pools = sc->knobs.pools;
    for (int q = 0; q < sc->n_pools && e == hipSuccess; ++q) {
        PoolCtx &c = sc->pools[q];
        if (q > 0) e = hipStreamCreateWithFlags(&c.main, hipStreamNonBlocking);
        if (sc->knobs.overlap) {
            for (int k = 0; k < 2 && e == hipSuccess; ++k) {
                e = hipStreamCreateWithFlags(&c.side[k], hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&c.join[k], hipEventDisableTiming);
            }
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c.book, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&c.shaded, hipEventDisableTiming);
        }
    }
    for (int q = 0; q < sc->n_pools && e == hipSuccess; ++q) e = hipEventCreateWithFlags(&sc->pools[q].readback, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&sc->fork, hipEventDisableTiming);
    auto up = [&](auto &buf, const auto &vec) {
        if (e == hipSuccess) e = upload(buf, vec);
    };
    if (hs.builder < 2) {
        up(sc->nodes, hs.nodes);
        if (!hs.qnodes.empty()) up(sc->qnodes, hs.qnodes);
    }
    up(sc->tri_geom, hs.tri_geom);
    up(sc->tri_shade, hs.tri_shade);
    up(sc->spheres, hs.spheres);
    up(sc->mats, hs.materials);
    up(sc->lights, hs.lights);
    up(sc->light_nodes, hs.light_nodes);
    up(sc->light_tris, hs.light_tris);
    up(sc->env, hs.env);
    if (!hs.instances.empty()) up(sc->inst, hs.instances);
    if (e != hipSuccess) {
        mcpt_scene_destroy(sc);
        return fail(e == hipErrorOutOfMemory ? MCPT_ERR_OOM : MCPT_ERR_HIP, std::string("scene upload: ") + hipGetErrorString(e));
    }
    double gpu_build_ms = 0.0;
    if (hs.builder >= 2) {  // the traversal tree is built on the device from the caller's triangles (csrc/mcpt_lbvh.hip)
        const auto tb = std::chrono::steady_clock::now();
        const int n_sph = (int)hs.sphere_objects.size();
        const int n_prim = hs.n_triangles + n_sph;
        DevBuf<mcpt_triangle> d_tris;
        DevBuf<int32_t> d_sph;
        e = d_tris.alloc((size_t)hs.n_triangles);
        if (e == hipSuccess && hs.n_triangles > 0) e = hipMemcpy(d_tris.p, desc->triangles, (size_t)hs.n_triangles * sizeof(mcpt_triangle), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = upload(d_sph, hs.sphere_objects);
        if (e == hipSuccess) e = sc->nodes.alloc((size_t)n_prim - 1);
        if (e == hipSuccess) e = sc->qnodes.alloc((size_t)n_prim - 1);
        LbvhResult R;
        if (e == hipSuccess) e = build_lbvh_device(d_tris.p, hs.n_triangles, d_sph.p, sc->spheres.p, n_sph, choice.quantise, hs.builder == 3 ? 1 : 0, choice.ploc_radius, choice.ploc_top,
                                                 sc->nodes.p, sc->qnodes.p, &R, nullptr);
        if (e != hipSuccess) {
            mcpt_scene_destroy(sc);
            return fail(e == hipErrorOutOfMemory ? MCPT_ERR_OOM : MCPT_ERR_HIP, std::string("GPU BVH build: ") + hipGetErrorString(e));
        }
        if (R.height > kMaxBvhHeight) {
            mcpt_scene_destroy(sc);
            return fail(MCPT_ERR_LIMIT, "the GPU-built BVH is deeper than the traversal stack (kMaxBvhHeight); use MCPT_BUILD_SAH");
        }
        hs.root = R.root;
        hs.height = R.height;
        for (int k = 0; k < 3; ++k) {
            hs.root_min[k] = R.root_min[k];
            hs.root_max[k] = R.root_max[k];
            hs.q_origin[k] = R.q_origin[k];
            hs.q_cell[k] = R.q_cell[k];
        }
        if (!R.quantised) sc->qnodes.release();
        sc->n_inner = R.n_nodes;
        gpu_build_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
    } else {
        sc->n_inner = hs.root < 0 ? 0 : (int32_t)hs.nodes.size();
    }
    if (traversal_stack_entries(hs.height) < hs.height - 1) {  // whatever built the tree: a push beyond the stack (one entry per inner ancestor) would be dropped
        mcpt_scene_destroy(sc);
        return fail(MCPT_ERR_LIMIT, "the BVH is deeper than the traversal stack (kMaxBvhHeight)");
    }
    DevScene &v = sc->view;
    v.nodes = sc->nodes.p;
    v.light_area_sum = hs.light_area_sum;
    v.qnodes = sc->qnodes.p;  // nullptr: the float nodes are traversed
    v.pnodes = nullptr;       // (set by the SMALL kernels to their LDS copy)
    for (int k = 0; k < 3; ++k) {
        v.q_origin[k] = hs.q_origin[k];
        v.q_cell[k] = hs.q_cell[k];
    }
    v.tri_geom = sc->tri_geom.p;
    v.tri_shade = sc->tri_shade.p;
    v.spheres = sc->spheres.p;
    v.mats = sc->mats.p;
    v.lights = sc->lights.p;
    v.light_nodes = sc->light_nodes.p;
    v.light_tris = sc->light_tris.p;
    v.inst = hs.instances.empty() ? nullptr : sc->inst.p;
    v.n_leaf_prims = hs.n_leaf_prims;
    v.env = sc->env.p;
    for (int k = 0; k < 3; ++k) {
        v.root_min[k] = hs.root_min[k];
        v.root_max[k] = hs.root_max[k];
        v.background[k] = hs.background[k];
    }
    v.root = hs.root;
    v.n_tri = hs.n_triangles;
    v.n_lights = (int32_t)hs.lights.size();
    v.env_w = hs.env_w;
    v.env_h = hs.env_h;
    v.height = hs.height;
    for (int k = 0; k < 3; ++k) v.light_center[k] = hs.light_center[k];
    v.light_radius = hs.light_radius;
    v.n_inner = sc->n_inner;
    v.n_sphere_slots = (int32_t)hs.spheres.size();
    v.n_mats = (int32_t)hs.materials.size();
    v.n_light_nodes = (int32_t)hs.light_nodes.size();
    v.n_light_tris = (int32_t)hs.light_tris.size();
    // the LDS-resident flavour (SMALL kernels): everything the traversal and light sampling read fits the kSmall* limits
    v.small = 0;
#if !defined(MCPT_FORCE_RETRY) && !defined(MCPT_LDS_ONLY_STACKS)
    if (sc->knobs.small_scene && !v.inst && v.root >= 0 && v.n_inner <= kSmallNodes && v.n_tri <= kSmallTris && v.n_sphere_slots <= kSmallSphereSlots &&
        v.n_mats <= kSmallMats && v.n_lights <= kSmallLights && v.n_light_nodes <= kSmallLightNodes && v.n_light_tris <= kSmallLightTris &&
        v.height - 1 <= kSmallStk)
        v.small = 1;
#endif
    v.dbg = nullptr;
#if defined(MCPT_TRAVERSAL_STATS) || defined(MCPT_CHECK_DIRECT_SKIP)
    if (sc->dbg.alloc(32) == hipSuccess) {  // (16 reported by mcpt_debug_counters; the statistics build prints the rest at destruction)
        (void)hipMemset(sc->dbg.p, 0, 32 * sizeof(unsigned long long));
        v.dbg = sc->dbg.p;
    }
#endif
    sc->info.build_ms = hb.build_ms + gpu_build_ms;
    sc->info.init_ms = hb.init_ms;
    sc->info.upload_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_upload).count() - gpu_build_ms;
    sc->info.builder = hs.builder;
    sc->info.quantised = sc->qnodes.p ? 1 : 0;
    sc->info.n_instances = (int32_t)hs.instances.size();
    sc->info.lds_resident = v.small;
    sc->info.n_nodes = sc->n_inner;
    sc->info.bvh_height = hs.height;
    sc->info.n_lights = v.n_lights;
    sc->info.n_prims = hs.n_triangles + hs.n_objects;
    sc->info.scene_bytes = sc->nodes.bytes() + sc->qnodes.bytes() + sc->tri_geom.bytes() + sc->tri_shade.bytes() + sc->spheres.bytes() +
                           sc->mats.bytes() + sc->lights.bytes() + sc->light_nodes.bytes() + sc->light_tris.bytes() + sc->inst.bytes() +
                           sc->env.bytes();
    *out = sc;
    return MCPT_OK;
}

}  // namespace mcpt

extern "C" {

void mcpt_scene_destroy(mcpt_scene *sc) {
    if (!sc) return;
    (void)hipSetDevice(sc->device);
#if defined(MCPT_TRAVERSAL_STATS) || defined(MCPT_CHECK_DIRECT_SKIP)
    if (sc->dbg.p) {
        unsigned long long h[32];
        if (hipMemcpy(h, sc->dbg.p, sizeof h, hipMemcpyDeviceToHost) == hipSuccess) {
            if (h[23])
                std::fprintf(stderr, "[mcpt k_shade stats] %llu waves: %.0f cycles from start to the end of the allocation, of which %.0f in its first half (ballots, first barrier) and %.0f in the second barrier\n",
                             h[23], (double)h[22] / h[23], (double)h[20] / h[23], (double)h[21] / h[23]);
            if (h[24] + h[25] + h[26] + h[27] + h[28])
                std::fprintf(stderr, "[mcpt k_shade stats] waves by the number of material types among their shading lanes: 0: %llu, 1: %llu, 2: %llu, 3: %llu, 4: %llu; shading lanes %llu\n",
                             h[24], h[25], h[26], h[27], h[28], h[29]);
            if (h[14]) std::fprintf(stderr, "[mcpt direct-skip check] light samples at skipped vertices: %llu, non-zero contributions among them: %llu\n", h[14], h[15]);
            for (int k = 0; k < 2; ++k) {
                const unsigned long long *d = h + 8 * k;
                if (!d[0]) continue;
                std::fprintf(stderr, "[mcpt traversal stats] %s: rays %llu, node visits/ray %.2f, prim tests/ray %.2f, %s %.3f, found %.3f, SIMD efficiency %.3f\n",
                             k ? "shadow" : "closest", d[0], (double)d[1] / d[0], (double)d[2] / d[0], k ? "occluded" : "hit",
                             (double)d[3] / d[0], (double)d[5] / d[0], (double)(d[1] + d[2]) / (double)d[4]);
            }
        }
        sc->dbg.release();
    }
#endif
    for (PoolCtx &c : sc->pools) {
        c.ws.release();
        c.timer.release();
        for (int k = 0; k < 2; ++k) {
            if (c.join[k]) (void)hipEventDestroy(c.join[k]);
            if (k == 0 && c.book) (void)hipEventDestroy(c.book);
            if (k == 0 && c.shaded) (void)hipEventDestroy(c.shaded);
            if (k == 0 && c.readback) (void)hipEventDestroy(c.readback);
            if (c.side[k]) (void)hipStreamDestroy(c.side[k]);
        }
        if (c.main) (void)hipStreamDestroy(c.main);
    }
    sc->shared.release();
    if (sc->fork) (void)hipEventDestroy(sc->fork);
    sc->nodes.release(); sc->qnodes.release(); sc->tri_geom.release(); sc->tri_shade.release(); sc->spheres.release(); sc->mats.release();
    sc->lights.release(); sc->light_nodes.release(); sc->light_tris.release(); sc->env.release(); sc->inst.release();
    delete sc;
}

int mcpt_bvh_dump(const mcpt_scene_desc *desc, mcpt_bvh_info *info, float *boxes, int32_t *children, uint16_t *qboxes,
                  float *inst_shift, int32_t *inst_root_first) {
    if (!desc || !info) return fail(MCPT_ERR_ARG, "mcpt_bvh_dump: null argument");
    HostScene hs;
    const char *err = "";
    const BuildChoice choice = resolve_build_choice(nullptr);
    if (choice.builder == MCPT_BUILD_GPU_LBVH || choice.builder == MCPT_BUILD_GPU_PLOC)
        return fail(MCPT_ERR_ARG, "mcpt_bvh_dump: the tree is built on the device (MCPT_BVH=lbvh / ploc): use mcpt_scene_dump_bvh");
    const int rc = build_host_scene(*desc, hs, &err, choice);
    if (rc != MCPT_OK) return fail(rc, std::string("mcpt_bvh_dump: ") + err);
    std::memset(info, 0, sizeof *info);
    const bool placeholder = hs.root < 0;  // a single primitive: no inner node (the array holds one unused record)
    info->n_nodes = placeholder ? 0 : (int32_t)hs.nodes.size();
    info->root = hs.root;
    info->stack_entries = hs.height;
    info->quantised = hs.qnodes.empty() ? 0 : 1;
    info->n_instances = (int32_t)hs.instances.size();
    info->n_leaf_prims = hs.n_leaf_prims;
    for (int k = 0; k < 3; ++k) {
        info->root_min[k] = hs.root_min[k];
        info->root_max[k] = hs.root_max[k];
        info->q_origin[k] = hs.q_origin[k];
        info->q_cell[k] = hs.q_cell[k];
    }
    for (size_t k = 0; k < hs.instances.size() && inst_shift && inst_root_first; ++k) {
        for (int c = 0; c < 3; ++c) inst_shift[3 * k + c] = hs.instances[k].shift[c];
        inst_root_first[2 * k] = hs.instances[k].root;
        inst_root_first[2 * k + 1] = hs.instances[k].first_tri;
    }
    if (!boxes || !children) return MCPT_OK;
    for (int32_t i = 0; i < info->n_nodes; ++i) {
        const Node &N = hs.nodes[i];
        float *b = boxes + (size_t)i * 12;
        for (int k = 0; k < 3; ++k) {
            b[k] = N.lmin[k];
            b[3 + k] = N.lmax[k];
            b[6 + k] = N.rmin[k];
            b[9 + k] = N.rmax[k];
        }
        children[2 * i] = N.left;
        children[2 * i + 1] = N.right;
        if (qboxes && info->quantised) {
            const QNode &Q = hs.qnodes[i];
            uint16_t *q = qboxes + (size_t)i * 12;
            for (int w = 0; w < 6; ++w) {
                q[2 * w] = (uint16_t)(Q.w[w] & 0xffffu);
                q[2 * w + 1] = (uint16_t)(Q.w[w] >> 16);
            }
        }
    }
    return MCPT_OK;
}

int mcpt_scene_dump_bvh(mcpt_scene *sc, mcpt_bvh_info *info, float *boxes, int32_t *children, uint16_t *qboxes, float *inst_shift,
                        int32_t *inst_root_first) {
    if (!sc || !info) return fail(MCPT_ERR_ARG, "mcpt_scene_dump_bvh: null argument");
    HIP_TRY(hipSetDevice(sc->device));
    std::memset(info, 0, sizeof *info);
    const DevScene &v = sc->view;
    info->n_nodes = sc->n_inner;
    info->root = v.root;
    info->stack_entries = v.height;
    info->quantised = v.qnodes ? 1 : 0;
    info->n_instances = sc->info.n_instances;
    info->n_leaf_prims = v.n_leaf_prims;
    if (info->n_instances > 0 && inst_shift && inst_root_first) {
        std::vector<InstRec> I((size_t)info->n_instances);
        HIP_TRY(hipMemcpy(I.data(), sc->inst.p, I.size() * sizeof(InstRec), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < I.size(); ++k) {
            for (int c = 0; c < 3; ++c) inst_shift[3 * k + c] = I[k].shift[c];
            inst_root_first[2 * k] = I[k].root;
            inst_root_first[2 * k + 1] = I[k].first_tri;
        }
    }
    for (int k = 0; k < 3; ++k) {
        info->root_min[k] = v.root_min[k];
        info->root_max[k] = v.root_max[k];
        info->q_origin[k] = v.q_origin[k];
        info->q_cell[k] = v.q_cell[k];
    }
    if (!boxes || !children || info->n_nodes == 0) return MCPT_OK;
    std::vector<Node> nodes((size_t)info->n_nodes);
    HIP_TRY(hipMemcpy(nodes.data(), sc->nodes.p, nodes.size() * sizeof(Node), hipMemcpyDeviceToHost));
    std::vector<QNode> qn;
    if (qboxes && info->quantised) {
        qn.resize(nodes.size());
        HIP_TRY(hipMemcpy(qn.data(), sc->qnodes.p, qn.size() * sizeof(QNode), hipMemcpyDeviceToHost));
    }
    for (int32_t i = 0; i < info->n_nodes; ++i) {
        const Node &N = nodes[i];
        float *b = boxes + (size_t)i * 12;
        for (int k = 0; k < 3; ++k) {
            b[k] = N.lmin[k];
            b[3 + k] = N.lmax[k];
            b[6 + k] = N.rmin[k];
            b[9 + k] = N.rmax[k];
        }
        children[2 * i] = N.left;
        children[2 * i + 1] = N.right;
        if (!qn.empty()) {
            uint16_t *q = qboxes + (size_t)i * 12;
            for (int w = 0; w < 6; ++w) {
                q[2 * w] = (uint16_t)(qn[i].w[w] & 0xffffu);
                q[2 * w + 1] = (uint16_t)(qn[i].w[w] >> 16);
            }
        }
    }
    return MCPT_OK;
}

}  // extern "C"
namespace mcpt {
void set_device_sharers(mcpt_scene *sc, int n) {
    if (sc) sc->device_sharers = n > 0 ? n : 1;
}
}  // namespace mcpt
extern "C" {

int mcpt_scene_get_info(const mcpt_scene *sc, mcpt_scene_info *info) {
    if (!sc || !info) return fail(MCPT_ERR_ARG, "mcpt_scene_get_info: null argument");
    *info = sc->info;
    return MCPT_OK;
}

int mcpt_render_device(mcpt_scene *sc, const mcpt_camera *cam, const mcpt_params *p, float *fb_device, void *hip_stream,
                       mcpt_stats *stats) {
    return render_impl(sc, cam, p, fb_device, (hipStream_t)hip_stream, stats);
}

int mcpt_render(mcpt_scene *sc, const mcpt_camera *cam, const mcpt_params *p, float *fb_host, mcpt_stats *stats) {
    if (!sc || !cam || !p || !fb_host) return fail(MCPT_ERR_ARG, "mcpt_render: null argument");
    HIP_TRY(hipSetDevice(sc->device));
    const size_t n = (size_t)cam->width * cam->height * 3;
    DevBuf<float> fb;
    HIP_TRY(fb.alloc(n));
    if (p->accumulate) HIP_TRY(hipMemcpy(fb.p, fb_host, n * sizeof(float), hipMemcpyHostToDevice));
    const int rc = render_impl(sc, cam, p, fb.p, nullptr, stats);
    if (rc == MCPT_OK || rc == MCPT_ERR_OVERFLOW) {
        const hipError_t e = hipMemcpy(fb_host, fb.p, n * sizeof(float), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("framebuffer download: ") + hipGetErrorString(e));
    }
    return rc;
}

int mcpt_intersect(mcpt_scene *sc, int64_t n, const float *origins, const float *dirs, double *out_t, int32_t *out_prim) {
    if (!sc || n < 0 || (n > 0 && (!origins || !dirs || !out_t || !out_prim))) return fail(MCPT_ERR_ARG, "mcpt_intersect: bad argument");
    if (n == 0) return MCPT_OK;
    if (n > 0x7fffffff) return fail(MCPT_ERR_ARG, "mcpt_intersect: too many rays for one call");
    HIP_TRY(hipSetDevice(sc->device));
    std::vector<float4> o(n), d(n);
    for (int64_t i = 0; i < n; ++i) {
        o[i] = make_float4(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2], 0.f);
        d[i] = make_float4(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], 0.f);
    }
    DevBuf<float4> dO, dD;
    DevBuf<uint4> dH;
    HIP_TRY(upload(dO, o));
    HIP_TRY(upload(dD, d));
    HIP_TRY(dH.alloc(n));
    RetryBufs retry;
    if (stack_uses_retry(sc->view.height)) {
        const uint32_t want[3] = {(uint32_t)n, 1u, 1u};
        HIP_TRY(retry.alloc(want));
    }
    launch_trace_closest(sc->view, (uint32_t)n, nullptr, dO.p, dD.p, dH.p, retry.list(0), nullptr);
    std::vector<uint4> h(n);
    const hipError_t e = hipMemcpy(h.data(), dH.p, n * sizeof(uint4), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("mcpt_intersect: ") + hipGetErrorString(e));
    for (int64_t i = 0; i < n; ++i) {
        const unsigned long long b = ((unsigned long long)h[i].y << 32) | h[i].x;
        double t;
        std::memcpy(&t, &b, sizeof t);
        out_t[i] = t;
        out_prim[i] = (int32_t)h[i].z;
    }
    return MCPT_OK;
}

int mcpt_cast_rays(mcpt_scene *sc, const mcpt_params *pp, int64_t n, const float *origins, const float *dirs,
                   const uint32_t *pixel, const uint32_t *sample, const int32_t *channel, float *out) {
    if (!sc || !pp || n < 0 || (n > 0 && (!origins || !dirs || !pixel || !sample || !channel || !out)))
        return fail(MCPT_ERR_ARG, "mcpt_cast_rays: bad argument");
    if (n == 0) return MCPT_OK;
    const mcpt_params &p = *pp;
    if (p.n_dir_sample <= 0 || !(p.rr_rate > 0.f)) return fail(MCPT_ERR_ARG, "mcpt_cast_rays: n_dir_sample/rr_rate must be positive");
    for (int64_t i = 0; i < n; ++i)
        if (channel[i] < 0 || channel[i] > 2) return fail(MCPT_ERR_ARG, "mcpt_cast_rays: channel must be 0..2");
    HIP_TRY(hipSetDevice(sc->device));
    PoolCtx &ctx = sc->pools[0];
    Workspace &w = ctx.ws;
    SharedBufs &sh = sc->shared;
    const int max_depth = derive_max_depth(p);
    const int64_t chunk_max = 1 << 20;
    ctx.timer.reset();
    ctx.timer.enabled = false;
    for (int64_t base = 0; base < n; base += chunk_max) {
        const uint32_t m = (uint32_t)std::min<int64_t>(chunk_max, n - base);
        const uint32_t pool = std::max<uint32_t>((m + 2) / 3 * 3, 3 * 256);
        HIP_TRY(ensure_workspace(ctx, std::max(pool, w.pool), p.n_dir_sample, std::max(max_depth, w.max_depth), stack_uses_retry(sc->view.height)));
        HIP_TRY(sh.result.alloc(m));
        HIP_TRY(sh.key_pixel.alloc(m));
        HIP_TRY(sh.key_sample.alloc(m));
        HIP_TRY(sh.key_channel.alloc(m));
        std::vector<float4> o(m), d(m);
        for (uint32_t i = 0; i < m; ++i) {
            const int64_t q = base + i;
            o[i] = make_float4(origins[3 * q], origins[3 * q + 1], origins[3 * q + 2], 0.f);
            d[i] = make_float4(dirs[3 * q], dirs[3 * q + 1], dirs[3 * q + 2], 0.f);
        }
        HIP_TRY(hipMemcpy(w.wave[0].ray_o.p, o.data(), m * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(w.wave[0].ray_d.p, d.data(), m * sizeof(float4), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sh.key_pixel.p, pixel + base, m * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sh.key_sample.p, sample + base, m * sizeof(uint32_t), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(sh.key_channel.p, channel + base, m * sizeof(int32_t), hipMemcpyHostToDevice));
        RenderConst C;
        std::memset(&C, 0, sizeof C);
        C.rr_rate = p.rr_rate;
        C.inv_rr = 1 / p.rr_rate;
        C.n_dir = p.n_dir_sample;
        C.enable_shadow = p.enable_shadow;
        C.seed = p.seed;
        C.mode = 1;
        C.key_pixel = sh.key_pixel.p;
        C.key_sample = sh.key_sample.p;
        C.key_channel = sh.key_channel.p;
        C.max_depth = w.max_depth;
        C.result[0] = C.result[1] = sh.result.p;
        LoopTotals tot;
        const std::vector<PassPlan> plan{PassPlan{0u, m, 1, 0}};
        const int rc = drained(run_wavefront(sc, ctx, C, nullptr, plan, nullptr, nullptr, tot));
        if (rc != MCPT_OK) return rc;
        HIP_TRY(hipMemcpy(out + base, sh.result.p, m * sizeof(float), hipMemcpyDeviceToHost));
    }
    return MCPT_OK;
}

int mcpt_camera_rays(mcpt_scene *sc, const mcpt_camera *cam, uint32_t seed, int64_t n, const uint32_t *pixel,
                     const uint32_t *sample, float *origins, float *dirs) {
    if (!sc || !cam || n < 0 || (n > 0 && (!pixel || !sample || !origins || !dirs))) return fail(MCPT_ERR_ARG, "mcpt_camera_rays: bad argument");
    if (n == 0) return MCPT_OK;
    if (n > 0x7fffffff) return fail(MCPT_ERR_ARG, "mcpt_camera_rays: too many rays for one call");
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf<uint32_t> dP, dS;
    DevBuf<float4> dO, dD;
    HIP_TRY(dP.alloc(n));
    HIP_TRY(dS.alloc(n));
    HIP_TRY(dO.alloc(n));
    HIP_TRY(dD.alloc(n));
    HIP_TRY(hipMemcpy(dP.p, pixel, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dS.p, sample, n * sizeof(uint32_t), hipMemcpyHostToDevice));
    launch_camera_rays(make_camera(*cam), seed, (uint32_t)n, dP.p, dS.p, dO.p, dD.p, nullptr);
    std::vector<float4> o(n), d(n);
    hipError_t e = hipMemcpy(o.data(), dO.p, n * sizeof(float4), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(d.data(), dD.p, n * sizeof(float4), hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(MCPT_ERR_HIP, std::string("mcpt_camera_rays: ") + hipGetErrorString(e));
    for (int64_t i = 0; i < n; ++i) {
        origins[3 * i] = o[i].x; origins[3 * i + 1] = o[i].y; origins[3 * i + 2] = o[i].z;
        dirs[3 * i] = d[i].x; dirs[3 * i + 1] = d[i].y; dirs[3 * i + 2] = d[i].z;
    }
    return MCPT_OK;
}

int mcpt_tonemap_device(mcpt_scene *sc, const float *fb_device, int64_t n_pixels, uint8_t *rgba_device, void *hip_stream) {
    if (!sc || n_pixels < 0 || (n_pixels > 0 && (!fb_device || !rgba_device))) return fail(MCPT_ERR_ARG, "mcpt_tonemap_device: bad argument");
    if (n_pixels > 0x7fffffff) return fail(MCPT_ERR_ARG, "mcpt_tonemap_device: frame too large");
    HIP_TRY(hipSetDevice(sc->device));
    launch_tonemap(fb_device, (uint32_t)n_pixels, rgba_device, (hipStream_t)hip_stream);
    HIP_TRY(hipGetLastError());
    return MCPT_OK;
}

int mcpt_tonemap(mcpt_scene *sc, const float *fb_host, int64_t n_pixels, uint8_t *rgba_host) {
    if (!sc || n_pixels < 0 || (n_pixels > 0 && (!fb_host || !rgba_host))) return fail(MCPT_ERR_ARG, "mcpt_tonemap: bad argument");
    if (n_pixels == 0) return MCPT_OK;
    if (n_pixels > 0x7fffffff) return fail(MCPT_ERR_ARG, "mcpt_tonemap: frame too large");
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf<float> fb;
    DevBuf<uint8_t> out;
    HIP_TRY(fb.alloc((size_t)n_pixels * 3));
    HIP_TRY(out.alloc((size_t)n_pixels * 4));
    HIP_TRY(hipMemcpy(fb.p, fb_host, (size_t)n_pixels * 3 * sizeof(float), hipMemcpyHostToDevice));
    launch_tonemap(fb.p, (uint32_t)n_pixels, out.p, nullptr);
    HIP_TRY(hipMemcpy(rgba_host, out.p, (size_t)n_pixels * 4, hipMemcpyDeviceToHost));
    return MCPT_OK;
}

int mcpt_debug_counters(mcpt_scene *sc, uint64_t out[16]) {
    if (!sc || !out) return fail(MCPT_ERR_ARG, "mcpt_debug_counters: null argument");
    std::memset(out, 0, 16 * sizeof(uint64_t));
    if (!sc->dbg.p) return MCPT_OK;  // a product build: nothing is counted
    HIP_TRY(hipSetDevice(sc->device));
    HIP_TRY(hipDeviceSynchronize());
    unsigned long long h[16];
    HIP_TRY(hipMemcpy(h, sc->dbg.p, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < 16; ++k) out[k] = h[k];
    return MCPT_OK;
}

int mcpt_debug_fmath(mcpt_scene *sc, int kind, int64_t n, const float *x, const float *y, float *out) {
    if (!sc || kind < 0 || kind > 5 || n < 0 || (n > 0 && (!x || !out || ((kind == 2 || kind == 4) && !y)))) return fail(MCPT_ERR_ARG, "mcpt_debug_fmath: bad argument");
    if (n == 0) return MCPT_OK;
    if (n > 0x7fffffff) return fail(MCPT_ERR_ARG, "mcpt_debug_fmath: too many values for one call");
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf<float> dX, dY, dO;
    HIP_TRY(dX.alloc(n));
    HIP_TRY(dY.alloc(n));
    HIP_TRY(dO.alloc(n));
    HIP_TRY(hipMemcpy(dX.p, x, n * sizeof(float), hipMemcpyHostToDevice));
    if (y) HIP_TRY(hipMemcpy(dY.p, y, n * sizeof(float), hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(dY.p, 0, n * sizeof(float)));
    launch_debug_fmath(kind, (uint32_t)n, dX.p, dY.p, dO.p, nullptr);
    HIP_TRY(hipMemcpy(out, dO.p, n * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

int mcpt_debug_material(mcpt_scene *sc, int kind, int64_t n, const float *in, const int32_t *sel, float *out) {
    if (!sc || kind < 0 || kind > 6 || n < 0 || (n > 0 && (!in || !sel || !out))) return fail(MCPT_ERR_ARG, "mcpt_debug_material: bad argument");
    if (n == 0) return MCPT_OK;
    if (n > 0x0fffffff) return fail(MCPT_ERR_ARG, "mcpt_debug_material: too many rows for one call");
    for (int64_t i = 0; i < n; ++i)
        if (sel[3 * i] < 0 || (size_t)sel[3 * i] >= sc->mats.bytes() / sizeof(MaterialRec) || sel[3 * i + 1] < 0 || sel[3 * i + 1] > 2) return fail(MCPT_ERR_ARG, "mcpt_debug_material: material or channel out of range");
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf<float> dI, dO;
    DevBuf<int32_t> dS;
    HIP_TRY(dI.alloc((size_t)n * 13));
    HIP_TRY(dS.alloc((size_t)n * 3));
    HIP_TRY(dO.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpy(dI.p, in, (size_t)n * 13 * sizeof(float), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(dS.p, sel, (size_t)n * 3 * sizeof(int32_t), hipMemcpyHostToDevice));
    launch_debug_material(sc->view, kind, (uint32_t)n, dI.p, dS.p, dO.p, nullptr);
    HIP_TRY(hipMemcpy(out, dO.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

int mcpt_debug_scene(mcpt_scene *sc, int kind, int64_t n, const float *in, float *out) {
    if (!sc || kind < 0 || kind > 1 || n < 0 || (n > 0 && (!in || !out))) return fail(MCPT_ERR_ARG, "mcpt_debug_scene: bad argument");
    if (n == 0) return MCPT_OK;
    if (n > 0x0fffffff) return fail(MCPT_ERR_ARG, "mcpt_debug_scene: too many rows for one call");
    const size_t n_in = kind == 0 ? 4 : 3, n_out = kind == 0 ? 10 : 3;
    HIP_TRY(hipSetDevice(sc->device));
    DevBuf<float> dI, dO;
    HIP_TRY(dI.alloc((size_t)n * n_in));
    HIP_TRY(dO.alloc((size_t)n * n_out));
    HIP_TRY(hipMemcpy(dI.p, in, (size_t)n * n_in * sizeof(float), hipMemcpyHostToDevice));
    launch_debug_scene(sc->view, kind, (uint32_t)n, dI.p, dO.p, nullptr);
    HIP_TRY(hipMemcpy(out, dO.p, (size_t)n * n_out * sizeof(float), hipMemcpyDeviceToHost));
    return MCPT_OK;
}

}  // extern "C"
