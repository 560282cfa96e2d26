#!/usr/bin/env python3
"""Benchmark of the path-tracing hot path on MI355X.

Metric (BASELINE.json): Msamples/s = width*height*spp / seconds / 1e6 on the chess scene at 1920x1080
(BASELINE config 4: conf.json as shipped, DoF on, RR 0.4, constant sky colour because sky.png is missing,
n_dir_sample = 4 as the reference executes; pass --n-dir 32 for the README's label).

A "step" is one pass of --spp-per-step samples per pixel over the whole frame, accumulated into the
device framebuffer (progressive rendering: K steps = K*spp_per_step spp of the same frame).  The defaults
(8 steps x 256 spp) are the full 1920x1080, spp = 2048 frame the metric is quoted on.  The K timed steps are
issued as ONE mcpt_render_device call with spp = K*spp_per_step and spp_per_pass = spp_per_step * N ranks (a rank owns
1/N of the pixels, so a pass is the same amount of work -- and the same per-pass result buffer -- at any N): the library
keeps two passes in flight, so the drain tail of a pass overlaps the start of the next one (--per-step-calls issues
one call per step instead).  The scene
is resident in HBM before the timed region.  With N ranks the frame is partitioned into interleaved
32x32 pixel tiles (strong scaling: the frame is fixed), every rank renders its tiles, and the timed
region ends with one RCCL reduce of the framebuffer to rank 0 (torch.distributed, backend nccl).

Launch: `python bench.py --gpus N` is enough.  With N > 1 and no WORLD_SIZE in the environment this process -- before it
imports torch or touches the GPU -- starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same
arguments>` as a CHILD process (never an exec), relays its output and exits with its code; started under torch.distributed.run
directly (WORLD_SIZE set) it is a rank.  Every rank reports the device it runs on (index, gcnArchName, PCI bus id, uuid); the
JSON line carries them under "ranks", and fewer distinct devices than ranks is an error unless --share-device (rehearsal).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BYTES_PER_RAY = 96     # ray record 32 B written + 32 B read, hit record 16 B written + 16 B read
BYTES_PER_VERTEX = 96  # path state 48 B read + 48 B written
BYTES_PER_SAMPLE = 12  # framebuffer contribution


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--n-dir", type=int, default=4)
    ap.add_argument("--scene", default="chess", choices=["chess", "chess_high", "cornell_demo", "cornell_rc"],
                    help="chess_high: conf.json with model_quality high honoured (296 k triangles; the shipped executable cannot reach it)")
    ap.add_argument("--pool-paths", type=int, default=0)
    ap.add_argument("--pass-steps", type=int, default=1,
                    help="the library renders in passes of this many steps' samples times the rank count (its per-pass result buffer; tools/pass_size.py: 256 spp per rank-share of the frame is within 1 % of the best)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=8, help="spp of the bounded CPU-baseline sample (full frame) on all cores; the 8-thread leg uses half")
    ap.add_argument("--serialized", action="store_true",
                    help="MCPT_OVERLAP=0: the library issues every kernel on one stream, so per-kernel durations are not inflated by "
                         "each other (profiling runs; the default overlaps three streams)")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--save-png", default="")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo + --share-device rehearses the N>1 path on a one-GPU box (the reduce goes through host memory)")
    ap.add_argument("--share-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--per-step-calls", action="store_true", help="one library call per step (drains between steps)")
    return ap.parse_args()


def make_scene(pkg, args):
    if args.scene == "chess":
        return pkg.scenes.chess_scene(width=args.width, height=args.height, spp=args.spp_per_step)
    if args.scene == "chess_high":
        return pkg.scenes.chess_high(width=args.width, height=args.height, spp=args.spp_per_step)
    if args.scene == "cornell_demo":
        return pkg.scenes.cornell_demo(args.width, args.height, args.spp_per_step)
    return pkg.scenes.cornell_rc(args.width, args.height, args.spp_per_step)


def host_cores():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (the GPU box
    exposes every host core but grants a share of them)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def load_traffic():
    """profiles/traffic.json: what the committed rocprofv3 passes of the serialised runs measured (tools/make_profile_summary.py), one
    entry per profiled configuration: per kernel the HBM bytes per launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes, gfx950
    correction applied), the serialised average launch duration and the VALU-busy fraction; per job the HBM bytes per sample.  None if absent."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return None


def launch_ranks(args):
    """`bench.py --gpus N` outside a launcher: run the N ranks under torch.distributed.run in a child process.  Called before torch
    is imported: this process never initialises HIP (a GPU-initialised process must not exec or fork into another GPU program)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)  # stdout / stderr are inherited: rank 0's JSON line is this process's output


def device_identity(torch, index):
    """What tells two devices apart: arch name, PCI domain:bus:device, uuid (attributes probed: they vary with the torch version)."""
    pr = torch.cuda.get_device_properties(index)
    pci = None
    if hasattr(pr, "pci_bus_id"):
        pci = "%04x:%02x:%02x" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, getattr(pr, "pci_device_id", 0))
    return {"device_index": index, "name": pr.name, "gcnArchName": getattr(pr, "gcnArchName", None), "pci_bus_id": pci,
            "uuid": str(getattr(pr, "uuid", "")) or None}


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    import torch
    import mcpt_loader
    pkg = mcpt_loader.load()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (under a launcher even ONE rank goes through the process group, the RCCL reduce and the device gathering: what a one-GPU box can
    # exercise of the nccl leg)
    distributed = "WORLD_SIZE" in os.environ
    if distributed and world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.share_device:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: local rank %d but only %d GPU(s) visible (--share-device rehearses N ranks on one GPU)"
                         % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend="gloo")
    # which device every rank really runs on: gathered on all ranks so that all of them stop together when devices are shared
    me = dict(device_identity(torch, local_rank), rank=rank, local_rank=local_rank, pid=os.getpid())
    ranks_seen = [me]
    if distributed:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, me)
        distinct = len({(r["pci_bus_id"], r["uuid"], r["device_index"]) for r in ranks_seen})
        if distinct < world and not args.share_device:
            raise SystemExit("bench.py: %d ranks but only %d distinct device(s): %s" % (world, distinct, ranks_seen))

    def reduce_to_rank0(t):
        if args.backend == "nccl":
            dist.reduce(t, dst=0)
        else:  # rehearsal path
            h = t.cpu()
            dist.reduce(h, dst=0)
            t.copy_(h)

    def allreduce(t, op):
        if args.backend == "nccl":
            dist.all_reduce(t, op=op)
            return t
        h = t.cpu()
        dist.all_reduce(h, op=op)
        return h

    sd = make_scene(pkg, args)
    W, H = args.width, args.height
    if args.serialized:
        os.environ["MCPT_OVERLAP"] = "0"  # read once, when the scene is created
    hs = pkg.HipScene(sd, device=local_rank)  # scene -> HBM, outside the timed region
    scene_info = hs.info()
    fb = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    spp_step = args.spp_per_step
    total_spp = spp_step * args.steps

    def step(k, spp_total, accumulate, n_steps=1):
        return hs.render_device(fb.data_ptr(), stream.cuda_stream, spp=spp_step * n_steps, spp_total=spp_total,
                                sample_offset=k * spp_step, accumulate=accumulate, seed=1, tile_size=32,
                                rank=rank, nranks=world, n_dir_sample=args.n_dir, spp_per_pass=args.pass_steps * spp_step * world,
                                pool_paths=args.pool_paths)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if args.warmup:
        step(0, max(1, args.warmup * spp_step), 0, n_steps=args.warmup)
    if distributed and args.warmup:
        reduce_to_rank0(fb)
    fb.zero_()

    barrier()
    t0 = time.perf_counter()
    agg = None
    if args.per_step_calls:
        for k in range(args.steps):
            d = step(k, total_spp, 1).as_dict()
            agg = d if agg is None else {key: agg[key] + d[key] for key in d}
    else:
        agg = step(0, total_spp, 1, n_steps=args.steps).as_dict()
    if distributed:
        reduce_to_rank0(fb)  # RCCL framebuffer merge over xGMI, inside the timed region
    barrier()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = allreduce(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX)
        dt = float(tmax.item())
        cnt = allreduce(torch.tensor([agg["samples"], agg["vertices"], agg["ref_scene_rays"], agg["closest_rays"], agg["shadow_rays"]],
                                     dtype=torch.float64, device=dev), dist.ReduceOp.SUM)
        tot_samples, tot_vertices, tot_ref_rays, tot_closest, tot_shadow = [float(x) for x in cnt.tolist()]
    else:
        tot_samples, tot_vertices, tot_ref_rays = float(agg["samples"]), float(agg["vertices"]), float(agg["ref_scene_rays"])
        tot_closest, tot_shadow = float(agg["closest_rays"]), float(agg["shadow_rays"])

    if rank != 0:
        if distributed:
            dist.destroy_process_group()
        return

    value = tot_samples / dt / 1e6
    # PCIe-inclusive variant (SURVEY 8(d) defines t_render with the framebuffer download; `value` leaves the frame in HBM)
    t1 = time.perf_counter()
    fb_host = fb.cpu()
    d2h_s = time.perf_counter() - t1

    # ---- roofline of the dominant kernel.  Durations: HIP events recorded by the library on the stream each kernel is
    # launched on, summed over the timed region (rocprofv3 --kernel-trace --stats of this command gives the same averages).
    # In the default run three streams overlap, which inflates every kernel's duration; the dominant kernel is therefore
    # picked by its share of the SERIALISED run (profiles/traffic.json, `bench.py --serialized`), whose figures are
    # reported beside the live ones.
    # units: rays traced per launch (k_trace, k_primary) / path vertices shaded (k_shade, k_direct)
    # continuation rays: Scene::intersect calls of the reference = castRay invocations + n_dir per shaded vertex + one look-ahead per
    # continuing vertex (mcpt_stats.ref_scene_rays), so the look-aheads -- the rays k_trace_closest traces -- follow from the counters;
    # the primary rays actually traced are the rest of closest_rays (pixels that can only see the background are never traced)
    n_cont = agg["ref_scene_rays"] - agg["vertices"] - args.n_dir * agg["shaded"]
    kern = {"k_trace<shadow>": (agg["ms_trace_shadow"], agg["n_trace_shadow"], agg["shadow_rays"]),
            "k_trace<closest>": (agg["ms_trace_closest"], agg["n_trace_closest"], n_cont),
            "k_primary": (agg["ms_generate"], agg["n_generate"], agg["closest_rays"] - n_cont),
            "k_direct": (agg["ms_direct"], agg["n_direct"], agg["direct_vertices"]),
            "k_shade": (agg["ms_shade"], agg["n_shade"], agg["shaded"])}
    # Profile-derived fields (limiter, PMC traffic, serialised figures, the dominant kernel of the SERIALISED step) come from
    # profiles/traffic.json and describe one build rendering one configuration: they are used only when this run is that
    # configuration on that build (source tag of csrc/); otherwise the dominant kernel is the live one and those fields are null.
    prof_all = load_traffic() or {}
    build_tag = pkg.build.source_tag()
    prof = {}
    for entry in prof_all.get("configs", {}).values():
        c = entry.get("config", {})
        if (c.get("scene"), c.get("width"), c.get("height"), c.get("n_dir")) == (args.scene, W, H, args.n_dir):
            prof = entry
    pc = prof.get("config", {})
    prof_ok = bool(prof) and prof_all.get("_build_tag") == build_tag
    ser = prof.get("serialized", {}) if prof_ok else {}
    live_dom = max(kern, key=lambda k: kern[k][0])
    dom = max(ser, key=lambda k: ser[k].get("ms_per_step", 0.0)) if ser and not args.serialized else live_dom
    if dom not in kern:
        dom = live_dom
    ms, n_launch, units = kern[dom]
    per_unit = BYTES_PER_VERTEX if dom in ("k_shade", "k_direct") else BYTES_PER_RAY
    roofline = None
    if ms > 0 and n_launch > 0:
        avg_ms = ms / n_launch
        bytes_per_launch = per_unit * units / n_launch
        achieved = bytes_per_launch / (avg_ms * 1e-3) / 1e9
        # PMC bytes per launch were measured on the serialised profile's launches; scale them to this run's launch size (bytes per
        # ray / vertex are what carries over)
        traffic = prof.get("kernels", {}).get(dom) if prof_ok else None
        ser_units = ser.get(dom, {}).get("units_per_launch")
        if traffic and ser_units:
            traffic = int(traffic * (units / n_launch) / ser_units)
        vb = ser.get(dom, {}).get("valu_busy_frac")
        roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                    "avg_launch_ms": round(avg_ms, 4), "launches": int(n_launch), "units_per_launch": round(units / n_launch, 1),
                    "algorithmic_bytes_per_unit": per_unit,
                    "live_dom": live_dom,  # the kernel with the largest summed HIP-event time in THIS run (streams overlap)
                    "timing": "HIP events on the launch streams, timed region, %s" % ("one stream (serialised)" if args.serialized else "three overlapping streams"),
                    "traffic_GBps": None if not traffic else round(traffic / (avg_ms * 1e-3) / 1e9, 1),
                    # what actually limits this kernel (PMC pass of the serialised run): the share of its duration the SIMDs
                    # spend issuing vector instructions -- an L2-resident scene is VALU-issue bound long before HBM-bound
                    "limiter": None if vb is None else ("valu_issue" if vb > 0.6 else "hbm/latency"),
                    "valu_busy_frac": vb,
                    "serialized": ser.get(dom),
                    "profile": {"applies": prof_ok, "build_tag": build_tag, "profile_build_tag": prof_all.get("_build_tag"), "profile_config": pc or None},
                    "kernel_ms": {k: round(v[0], 2) for k, v in kern.items()},
                    "kernel_launches": {k: int(v[1]) for k, v in kern.items()},
                    "kernel_units": {k: int(v[2]) for k, v in kern.items()}}
    ref_bytes_per_sample = (BYTES_PER_RAY * tot_ref_rays + BYTES_PER_VERTEX * tot_vertices) / tot_samples + BYTES_PER_SAMPLE
    ref_gbs = value * 1e6 * ref_bytes_per_sample / 1e9
    traced_bytes_per_sample = (BYTES_PER_RAY * (tot_closest + tot_shadow) + BYTES_PER_VERTEX * tot_vertices) / tot_samples + BYTES_PER_SAMPLE
    jt = prof.get("job", {})
    hbm_bps = jt.get("hbm_bytes_per_sample") if prof_ok else None

    # ---- parity vs the CPU oracle, same Philox seed, on a reduced configuration of the same scene
    parity = None
    if not args.no_psnr:
        import numpy as np
        from oracle import oracle as orc
        small = make_scene(pkg, argparse.Namespace(**{**vars(args), "width": 240, "height": 136}))
        a, _ = orc.OracleScene(small).render(spp=8, seed=1, n_dir_sample=args.n_dir)
        b, _ = pkg.HipScene(small, device=local_rank).render(spp=8, seed=1, n_dir_sample=args.n_dir)
        psnr = pkg.pngio.psnr_u8(pkg.pngio.tonemap_u8(a), pkg.pngio.tonemap_u8(b))
        differing = int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())
        parity = {"config": "%s 240x136 spp 8, same Philox seed, vs the CPU oracle" % args.scene,
                  "psnr_db": round(min(psnr, 200.0), 2),  # 200 = the two 8-bit images are identical
                  "differing_framebuffer_values": differing, "framebuffer_values": int(a.size)}

    # ---- CPU baseline: the oracle (a port of the reference path) on this box's host cores, bounded samples:
    # all cores the cgroup grants, and 8 threads (the reference's fixed PARALLELISM, Renderer.cpp:16)
    cpu, cpu8 = None, None
    if not args.no_cpu_baseline and world == 1:  # (rank 0 at N = 1 only)
        from oracle import oracle as orc
        ncores = host_cores()
        osc = orc.OracleScene(sd)

        def cpu_leg(threads, spp):
            _, cst = osc.render(spp=spp, seed=1, n_threads=threads, n_dir_sample=args.n_dir)
            return {"value": round(cst.samples / cst.seconds / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
                    "sample": "%s %dx%d spp %d, n_dir %d, OpenMP schedule(dynamic,8), oracle built -O3, %.1f s" %
                              (args.scene, W, H, spp, args.n_dir, cst.seconds)}
        cpu = cpu_leg(ncores, args.cpu_spp)
        cpu8 = cpu_leg(min(8, ncores), max(1, args.cpu_spp // 2))

    if args.save_png:
        img = pkg.pngio.tonemap_u8(fb_host.numpy().reshape(H, W, 3))
        pkg.pngio.write_png(args.save_png, img)

    out = {
        "metric": "Msamples/s (pixels x spp), %s scene %dx%d, PSNR vs CPU" % (args.scene, W, H),  # (the default run: chess scene 1920x1080, BASELINE's metric)
        "value": round(value, 3), "unit": "Msamples/s", "n_gpus": world, "ranks": ranks_seen, "backend": args.backend if distributed else None, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt * 1e3 / args.steps, 3), "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "%s %dx%d, %d spp per step, n_dir_sample %d, RR %.2f, DoF %s, constant sky colour"
                               % (args.scene, W, H, spp_step, args.n_dir, sd.rr_rate, "on" if int(sd.camera["use_dof"].reshape(-1)[0]) else "off"),
                   "partition": "interleaved 32x32 tiles over %d rank(s), RCCL reduce of the framebuffer" % world,
                   "streams": "one (serialised)" if args.serialized else "three"},
        "wall_clock_1920x1080_spp2048_s": round(1920 * 1080 * 2048 / (value * 1e6), 2),
        # the frame stays in HBM inside the timed region; with its download to host memory (mcpt_render's contract):
        "value_incl_d2h": round(tot_samples / (dt + d2h_s) / 1e6, 3), "d2h_ms": round(d2h_s * 1e3, 3),
        "psnr_vs_cpu_db": None if parity is None else parity["psnr_db"],
        "parity": parity,
        "roofline": roofline,
        "job": {
            # SURVEY 8(d)'s normalised figure: the bytes the REFERENCE's ray and vertex counts would stream (it traces every
            # continuation ray twice and one shadow ray per light sample); a throughput normalisation, not HBM traffic
            "reference_equivalent": {"bytes_per_sample": round(ref_bytes_per_sample, 1), "GBps": round(ref_gbs, 2),
                                     "frac_of_hbm_peak": round(ref_gbs / HBM_PEAK_GBS, 5),
                                     "ref_rays_per_sample": round(tot_ref_rays / tot_samples, 3)},
            # the same 96 B/ray + 96 B/vertex + 12 B/sample charged on the rays actually traced
            "algorithmic": {"bytes_per_sample": round(traced_bytes_per_sample, 1), "GBps": round(value * 1e6 * traced_bytes_per_sample / 1e9, 2),
                            "frac_of_hbm_peak": round(value * 1e6 * traced_bytes_per_sample / 1e9 / HBM_PEAK_GBS, 5)},
            # measured HBM traffic: bytes/sample from the PMC passes of the serialised run (profiles/traffic.json) x this run's rate
            "hbm_traffic": None if not hbm_bps else {"bytes_per_sample": round(hbm_bps, 1), "GBps": round(value * 1e6 * hbm_bps / 1e9, 1),
                                                     "frac_of_hbm_peak": round(value * 1e6 * hbm_bps / 1e9 / HBM_PEAK_GBS, 5),
                                                     "source": jt.get("source")},
            "vertices_per_sample": round(tot_vertices / tot_samples, 3),
            "traced_rays_per_sample": round((tot_closest + tot_shadow) / tot_samples, 3),
            "Mrays_per_s_traced": round((tot_closest + tot_shadow) / dt / 1e6, 1),
            "wavefront_iterations": int(agg["iterations"])},
        "scene": scene_info,
        "cpu_baseline": cpu,
        "cpu_baseline_8threads": cpu8,
    }
    print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
