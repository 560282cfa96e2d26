// Multi-GPU inside the boundary: mcpt_group_* keep the caller single-threaded like the reference's
// main() (Renderer::Render blocks, main.cpp:333) while every GPU of the node renders its share of the frame.
//
//   * one replica of the scene per device (the scene is a few MB; replication is the cheap part),
//   * the frame is partitioned into interleaved 32x32 tiles round-robin over the devices (mcpt_params.tile_size/rank/nranks:
//     disjoint pixels, the same Philox keys, so the merged frame is bit-identical to the one-GPU frame),
//   * one host thread per device drives that device's wavefront loop (mcpt_render_device on its own stream),
//   * when all of them have finished, the per-device frames (zero outside the owned tiles) are summed into device 0's frame with ONE ncclReduce over xGMI
//     (RCCL, loaded lazily with dlopen the first time a group with distinct devices renders), the path's only exchange step,
//   * device 0's frame is copied to the caller's host buffer.
// A group whose entries all name the same device is a rehearsal of this schedule on a one-GPU box: threads, partition and merge are
// the same, the sum is done by a kernel on that device instead of RCCL.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "mcpt_kernels.h"

using namespace mcpt;

namespace {

thread_local std::string g_group_err;

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string &err) {
        if (handle) return true;
        handle = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!handle) handle = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!handle) {
            err = std::string("cannot load RCCL: ") + dlerror();
            return false;
        }
        CommInitAll = (decltype(CommInitAll))dlsym(handle, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(handle, "ncclCommDestroy");
        Reduce = (decltype(Reduce))dlsym(handle, "ncclReduce");
        GroupStart = (decltype(GroupStart))dlsym(handle, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(handle, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(handle, "ncclGetErrorString");
        if (!CommInitAll || !CommDestroy || !Reduce || !GroupStart || !GroupEnd || !GetErrorString) {
            err = "RCCL is missing a required symbol";
            return false;
        }
        return true;
    }
};

}  // namespace

struct mcpt_group {
    std::vector<int> devices;
    std::vector<mcpt_scene *> scenes;
    std::vector<float *> fb;       // one device framebuffer per entry (W*H*3 floats), grown on demand
    std::vector<size_t> fb_floats;
    std::vector<hipStream_t> streams;
    bool same_device = true;
    RcclApi rccl;
    std::vector<ncclComm_t> comms;
    double build_ms = 0.0, setup_ms = 0.0;  // host build (once); all of mcpt_group_create
};

namespace {

int gfail(int code, const std::string &msg) {
    g_group_err = msg;
    return code;
}

}  // namespace

extern "C" {

const char *mcpt_group_last_error(void) { return g_group_err.c_str(); }

void mcpt_group_destroy(mcpt_group *g) {
    if (!g) return;
    for (size_t i = 0; i < g->comms.size(); ++i)
        if (g->comms[i] && g->rccl.CommDestroy) (void)g->rccl.CommDestroy(g->comms[i]);
    for (size_t i = 0; i < g->scenes.size(); ++i) {
        if (!g->scenes[i]) continue;  // (never created: its device index may not even exist)
        (void)hipSetDevice(g->devices[i]);
        if (i < g->fb.size() && g->fb[i]) (void)hipFree(g->fb[i]);
        if (i < g->streams.size() && g->streams[i]) (void)hipStreamDestroy(g->streams[i]);
        if (g->scenes[i]) mcpt_scene_destroy(g->scenes[i]);
    }
    // (the RCCL handle stays loaded: unloading a library that owns device state is not worth the risk)
    delete g;
}

int mcpt_group_create(const mcpt_scene_desc *desc, int n_devices, const int *devices, mcpt_group **out) {
    if (!desc || !out || n_devices <= 0 || n_devices > 64 || !devices) return gfail(MCPT_ERR_ARG, "mcpt_group_create: bad argument");
    *out = nullptr;
    mcpt_group *g = new (std::nothrow) mcpt_group();
    if (!g) return gfail(MCPT_ERR_OOM, "mcpt_group_create: host allocation failed");
    g->devices.assign(devices, devices + n_devices);
    g->scenes.assign(n_devices, nullptr);
    g->fb.assign(n_devices, nullptr);
    g->fb_floats.assign(n_devices, 0);
    g->streams.assign(n_devices, nullptr);
    g->same_device = true;
    for (int i = 1; i < n_devices; ++i) g->same_device = g->same_device && devices[i] == devices[0];
#ifdef MCPT_TEST_HOOKS
    // Test hook of the checking build only (libmcpt_hip_check.so): MCPT_GROUP_FORCE_RCCL=1 sends a ONE-device group through the RCCL merge
    // as well (a communicator of one rank), so that the library loading, communicator set-up and the ncclReduce call can be exercised on
    // a one-GPU box.
    const char *force = std::getenv("MCPT_GROUP_FORCE_RCCL");
    if (n_devices == 1 && force && force[0] == '1') g->same_device = false;
#endif
    if (!g->same_device)
        for (int i = 0; i < n_devices; ++i)
            for (int j = 0; j < i; ++j)
                if (devices[i] == devices[j]) {
                    mcpt_group_destroy(g);
                    return gfail(MCPT_ERR_ARG, "mcpt_group_create: a device may appear once (or every entry names the same device: rehearsal)");
                }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        mcpt_group_destroy(g);
        return gfail(MCPT_ERR_HIP, "mcpt_group_create: no HIP device available (this library has no CPU fallback)");
    }
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) {
            mcpt_group_destroy(g);
            return gfail(MCPT_ERR_ARG, "mcpt_group_create: device " + std::to_string(devices[i]) + ": device index out of range");
        }
    // The scene is flattened and its tree built ONCE, on the calling thread, while one helper thread per distinct device brings that
    // device up (context, code objects); then one thread per entry copies the flattened scene to its device.  (Round 2 rebuilt the tree
    // once per device, serially: 8 x (41 + ~150) ms for the chess scene.)
    const auto t0 = std::chrono::steady_clock::now();
    HostBuild hb;
    {
        std::vector<std::thread> warm;
        std::vector<double> warm_ms((size_t)n_devices, 0.0);
        for (int i = 0; i < n_devices; ++i) {
            bool first = true;
            for (int j = 0; j < i; ++j) first = first && devices[j] != devices[i];
            if (first) warm.emplace_back([&warm_ms, devices, i]() { warm_ms[(size_t)i] = warm_up_device(devices[i]); });
        }
        const int rc = build_scene_host(desc, nullptr, hb);
        for (std::thread &t : warm) t.join();
        if (rc != MCPT_OK) {
            const std::string e = mcpt_last_error();
            mcpt_group_destroy(g);
            return gfail(rc, "mcpt_group_create: " + e);
        }
        for (double v : warm_ms) hb.init_ms = std::max(hb.init_ms, v);
    }
    g->build_ms = hb.build_ms;
    std::vector<int> rc((size_t)n_devices, MCPT_OK);
    std::vector<std::string> err((size_t)n_devices);
    auto up = [&](int i) {
        HostBuild mine = hb;  // (upload_scene fills in what a device-side tree build returns: every thread works on its own copy)
        rc[(size_t)i] = upload_scene(desc, mine, devices[i], &g->scenes[(size_t)i]);
        if (rc[(size_t)i] != MCPT_OK) {
            err[(size_t)i] = mcpt_last_error();
            return;
        }
        if (hipSetDevice(devices[i]) != hipSuccess || hipStreamCreateWithFlags(&g->streams[(size_t)i], hipStreamNonBlocking) != hipSuccess) {
            rc[(size_t)i] = MCPT_ERR_HIP;
            err[(size_t)i] = "cannot create a stream";
        }
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < n_devices; ++i) th.emplace_back(up, i);
        up(0);
        for (std::thread &t : th) t.join();
    }
    for (int i = 0; i < n_devices; ++i)
        if (rc[(size_t)i] != MCPT_OK) {
            const int code = rc[(size_t)i];
            const std::string e = "mcpt_group_create: device " + std::to_string(devices[i]) + ": " + err[(size_t)i];
            mcpt_group_destroy(g);
            return gfail(code, e);
        }
    for (int i = 0; i < n_devices; ++i) {
        int sharers = 0;
        for (int j = 0; j < n_devices; ++j) sharers += devices[j] == devices[i] ? 1 : 0;
        set_device_sharers(g->scenes[(size_t)i], sharers);
    }
    g->setup_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *out = g;
    return MCPT_OK;
}

int mcpt_group_size(const mcpt_group *g) { return g ? (int)g->scenes.size() : 0; }

mcpt_scene *mcpt_group_scene(mcpt_group *g, int index) { return (g && index >= 0 && index < (int)g->scenes.size()) ? g->scenes[(size_t)index] : nullptr; }

int mcpt_group_get_info(const mcpt_group *g, mcpt_group_info *info) {
    if (!g || !info) return gfail(MCPT_ERR_ARG, "mcpt_group_get_info: null argument");
    std::memset(info, 0, sizeof *info);
    info->n_devices = (int32_t)g->scenes.size();
    info->uses_rccl = g->same_device ? 0 : 1;
    info->build_ms = g->build_ms;
    info->setup_ms = g->setup_ms;
    for (mcpt_scene *sc : g->scenes) {
        mcpt_scene_info si;
        if (sc && mcpt_scene_get_info(sc, &si) == MCPT_OK) {
            info->upload_ms_max = std::max(info->upload_ms_max, si.upload_ms);
            info->init_ms_max = std::max(info->init_ms_max, si.init_ms);
        }
    }
    return MCPT_OK;
}

int mcpt_group_render(mcpt_group *g, const mcpt_camera *cam, const mcpt_params *pp, float *fb_host, mcpt_stats *stats) {
    if (!g || !cam || !pp || !fb_host) return gfail(MCPT_ERR_ARG, "mcpt_group_render: null argument");
    if (cam->width <= 0 || cam->height <= 0) return gfail(MCPT_ERR_ARG, "mcpt_group_render: bad frame size");
    const auto t0 = std::chrono::steady_clock::now();
    const int N = (int)g->scenes.size();
    const size_t n = (size_t)cam->width * cam->height * 3;
    const int tile = pp->tile_size > 0 ? pp->tile_size : 32;

    if (!g->same_device && g->comms.empty()) {  // lazily: the communicator costs a second or so and one-GPU callers never need it
        std::string err;
        if (!g->rccl.load(err)) return gfail(MCPT_ERR_HIP, "mcpt_group_render: " + err);
        g->comms.assign(N, nullptr);
        const ncclResult_t r = g->rccl.CommInitAll(g->comms.data(), N, g->devices.data());
        if (r != ncclSuccess) {
            g->comms.clear();
            return gfail(MCPT_ERR_HIP, std::string("mcpt_group_render: ncclCommInitAll: ") + g->rccl.GetErrorString(r));
        }
    }

    std::vector<int> rc(N, MCPT_OK);
    std::vector<std::string> err(N);
    std::vector<mcpt_stats> st(N);
    auto work = [&](int i) {
        auto hip_ok = [&](hipError_t e, const char *what) {
            if (e == hipSuccess) return true;
            rc[i] = (e == hipErrorOutOfMemory) ? MCPT_ERR_OOM : MCPT_ERR_HIP;
            err[i] = std::string(what) + ": " + hipGetErrorString(e);
            return false;
        };
        if (!hip_ok(hipSetDevice(g->devices[i]), "hipSetDevice")) return;
        if (g->fb_floats[i] < n) {
            if (g->fb[i]) (void)hipFree(g->fb[i]);
            g->fb[i] = nullptr;
            g->fb_floats[i] = 0;
            if (!hip_ok(hipMalloc((void **)&g->fb[i], n * sizeof(float)), "hipMalloc(framebuffer)")) return;
            g->fb_floats[i] = n;
        }
        mcpt_params p = *pp;
        p.tile_size = tile;
        p.rank = i;
        p.nranks = N;
        hipStream_t s = g->streams[i];
        if (p.accumulate) {  // every replica continues from the caller's frame on ITS pixels; the others are zeroed before the merge
            if (!hip_ok(hipMemcpyAsync(g->fb[i], fb_host, n * sizeof(float), hipMemcpyHostToDevice, s), "framebuffer upload")) return;
        }
        rc[i] = mcpt_render_device(g->scenes[i], cam, &p, g->fb[i], (void *)s, &st[i]);
        if (rc[i] != MCPT_OK && rc[i] != MCPT_ERR_OVERFLOW) {
            err[i] = mcpt_last_error();
            return;
        }
        if (rc[i] == MCPT_ERR_OVERFLOW) err[i] = mcpt_last_error();
        if (p.accumulate) launch_mask_unowned(g->fb[i], cam->width, cam->height, tile, i, N, s);
        (void)hip_ok(hipStreamSynchronize(s), "hipStreamSynchronize");
    };
    {
        std::vector<std::thread> th;
        for (int i = 1; i < N; ++i) th.emplace_back(work, i);
        work(0);
        for (std::thread &t : th) t.join();
    }
    int worst = MCPT_OK;
    for (int i = 0; i < N; ++i)
        if (rc[i] != MCPT_OK && rc[i] != MCPT_ERR_OVERFLOW) return gfail(rc[i], "mcpt_group_render: device " + std::to_string(g->devices[i]) + ": " + err[i]);
        else if (rc[i] == MCPT_ERR_OVERFLOW) worst = MCPT_ERR_OVERFLOW;

    // The merge starts only when every replica has rendered without error: a rank that failed must not leave the others
    // waiting inside the collective.
    if (!g->same_device) {
        // frames are zero outside the owned tiles: the sum is exact, whatever order RCCL adds in.  One thread, one group call
        // over all communicators (the single-process multi-GPU pattern), then every stream is waited for.
        ncclResult_t r = g->rccl.GroupStart();
        for (int i = 0; i < N && r == ncclSuccess; ++i)
            r = g->rccl.Reduce(g->fb[i], g->fb[i], n, ncclFloat, ncclSum, 0, g->comms[i], g->streams[i]);
        const ncclResult_t r2 = g->rccl.GroupEnd();
        if (r == ncclSuccess) r = r2;
        if (r != ncclSuccess) return gfail(MCPT_ERR_HIP, std::string("mcpt_group_render: ncclReduce: ") + g->rccl.GetErrorString(r));
        for (int i = 0; i < N; ++i) {
            if (hipSetDevice(g->devices[i]) != hipSuccess || hipStreamSynchronize(g->streams[i]) != hipSuccess)
                return gfail(MCPT_ERR_HIP, "mcpt_group_render: the framebuffer reduce failed on device " + std::to_string(g->devices[i]));
        }
    }
    if (hipSetDevice(g->devices[0]) != hipSuccess) return gfail(MCPT_ERR_HIP, "mcpt_group_render: hipSetDevice");
    if (g->same_device) {  // rehearsal: every frame lives on this device
        for (int i = 1; i < N; ++i) launch_add_frame(g->fb[0], g->fb[i], (uint32_t)n, g->streams[0]);
    }
    hipError_t e = hipMemcpyAsync(fb_host, g->fb[0], n * sizeof(float), hipMemcpyDeviceToHost, g->streams[0]);
    if (e == hipSuccess) e = hipStreamSynchronize(g->streams[0]);
    if (e != hipSuccess) return gfail(MCPT_ERR_HIP, std::string("mcpt_group_render: framebuffer download: ") + hipGetErrorString(e));

    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        for (int i = 0; i < N; ++i) {
            const mcpt_stats &a = st[i];
            stats->samples += a.samples; stats->paths += a.paths; stats->vertices += a.vertices; stats->shaded += a.shaded;
            stats->closest_rays += a.closest_rays; stats->shadow_rays += a.shadow_rays; stats->ref_scene_rays += a.ref_scene_rays;
            stats->iterations += a.iterations; stats->overflow_paths += a.overflow_paths; stats->direct_vertices += a.direct_vertices;
            stats->ms_trace_closest += a.ms_trace_closest; stats->ms_trace_shadow += a.ms_trace_shadow; stats->ms_shade += a.ms_shade;
            stats->ms_generate += a.ms_generate; stats->ms_resolve += a.ms_resolve; stats->ms_direct += a.ms_direct;
            stats->n_trace_closest += a.n_trace_closest; stats->n_trace_shadow += a.n_trace_shadow; stats->n_shade += a.n_shade;
            stats->n_generate += a.n_generate; stats->n_resolve += a.n_resolve; stats->n_direct += a.n_direct;
        }
        stats->ms_total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
    if (worst == MCPT_ERR_OVERFLOW) return gfail(worst, "some paths outran the clamp stack (raise params.max_depth)");
    return MCPT_OK;
}

}  // extern "C"
