"""Multi-GPU on a one-GPU box.

Inside the boundary: mcpt_group_* with every entry naming device 0 rehearses the schedule the library runs on N GPUs -- one
replica and one host thread per entry, interleaved-tile partition, merge into the first frame -- and must reproduce the
one-GPU frame bit for bit (with distinct devices the merge is one RCCL ncclReduce; that leg needs an N-GPU node).
Outside: bench.py's one-process-per-GPU path (torch.distributed) is run as plain `python bench.py --gpus 2` (bench.py starts its
ranks itself) with both ranks sharing the GPU over gloo, and must write the PNG the 1-rank run writes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd", "host")
MODELS = os.path.join(ROOT, "assets", "models")


def test_group_rehearsal_is_bit_identical_to_one_gpu(pkg, hip):
    sd = pkg.scenes.chess_scene(width=200, height=120, spp=6)
    one, st1 = hip.HipScene(sd).render(spp=6, seed=4)
    for n in (2, 3):
        g = hip.HipGroup(sd, [0] * n)
        fb, st = g.render(spp=6, seed=4)
        assert np.array_equal(one, fb, equal_nan=True), n
        assert (st.samples, st.vertices, st.shaded, st.closest_rays, st.shadow_rays) == (st1.samples, st1.vertices, st1.shaded, st1.closest_rays, st1.shadow_rays)
        # progressive accumulation through the group: two calls of 3 spp == one call of 6 spp
        fb2, _ = g.render(spp=3, spp_total=6, sample_offset=0, seed=4)
        fb2, _ = g.render(fb=fb2, spp=3, spp_total=6, sample_offset=3, accumulate=1, seed=4)
        assert np.array_equal(one, fb2, equal_nan=True), n
        g.close()


def test_rccl_merge_path_with_a_communicator_of_one(pkg, hip, hip_check, monkeypatch):
    """The RCCL leg of mcpt_group_render (dlopen of librccl, ncclCommInitAll, ncclGroupStart / ncclReduce / ncclGroupEnd on the group's
    stream) needs distinct devices.  The CHECKING build has a test hook for it, MCPT_GROUP_FORCE_RCCL=1: a group of ONE device goes through
    a one-rank communicator, which is what a one-GPU box can exercise of that leg.  The frame must be the plain one; the product library
    has no such hook (it ignores the variable)."""
    sd = pkg.scenes.cornell_demo(64, 48, 4)
    ref, _ = hip.HipScene(sd).render(spp=4, seed=2)
    monkeypatch.setenv("MCPT_GROUP_FORCE_RCCL", "1")
    g = hip.HipGroup(sd, [0], library=hip_check)
    assert g.info()["uses_rccl"] == 1
    fb, st = g.render(spp=4, seed=2)
    assert np.array_equal(ref, fb, equal_nan=True) and st.samples == 64 * 48 * 4
    fb2, _ = g.render(fb=fb.copy(), spp=4, spp_total=8, sample_offset=4, accumulate=1, seed=2)  # the communicator is reused
    assert np.isfinite(fb2).all()
    g.close()
    g = hip.HipGroup(sd, [0])  # the product build
    assert g.info()["uses_rccl"] == 0
    g.close()


def test_group_builds_the_tree_once(pkg, hip):
    """mcpt_group_create flattens the scene and builds its tree once, then uploads from one thread per device: the set-up of four replicas
    costs about one build, not four (round 2: a build per device, serially)."""
    sd = pkg.scenes.chess_scene(width=64, height=36, spp=1)
    one = hip.HipGroup(sd, [0]).info()
    four = hip.HipGroup(sd, [0, 0, 0, 0]).info()
    print("\n[group] set-up of 1 replica: %.1f ms (build %.1f, upload %.1f, device init %.1f); of 4 replicas: %.1f ms (build %.1f, slowest upload %.1f)"
          % (one["setup_ms"], one["build_ms"], one["upload_ms_max"], one["init_ms_max"], four["setup_ms"], four["build_ms"], four["upload_ms_max"]))
    assert four["n_devices"] == 4 and four["build_ms"] < 2.5 * max(one["build_ms"], 20.0)
    assert four["setup_ms"] < 2.0 * four["build_ms"] + 4 * max(four["upload_ms_max"], 5.0) + 100.0


def test_group_argument_errors(pkg, hip):
    sd = pkg.scenes.cornell_rc(32, 32, 1)
    with pytest.raises(hip.McptError):
        hip.HipGroup(sd, [0, 0, 1])  # neither all-equal nor all-distinct
    with pytest.raises(hip.McptError):
        hip.HipGroup(sd, [0, 99])  # no such device
    g = hip.HipGroup(sd, [0])  # a group of one is just the scene
    fb, _ = g.render(spp=2, seed=1)
    ref, _ = hip.HipScene(sd).render(spp=2, seed=1)
    assert np.array_equal(fb, ref)


def test_cpp_executable_on_two_replicas(pkg, hip, tmp_path):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    exe = os.path.join(HOST, "RayTracingDemo")
    outs = []
    for extra, name in (([], "one.png"), (["--devices", "0,0"], "two.png")):
        out = str(tmp_path / name)
        p = subprocess.run([exe, "--models", MODELS, "--width", "96", "--height", "64", "--spp", "6", "--output", out] + extra,
                           cwd=str(tmp_path), capture_output=True, text=True)
        assert p.returncode == 0 and "Rendering finished in" in p.stdout, p.stderr
        outs.append(open(out, "rb").read())
    assert "2 GPU replicas" in p.stdout
    assert outs[0] == outs[1]


def test_bench_two_ranks_write_the_one_rank_png(tmp_path):
    """Plain `python bench.py --gpus 2` -- no launcher: bench.py starts its two ranks itself (a child torch.distributed.run, before the
    parent touches the GPU) -- both ranks on cuda:0, reduce through gloo; the JSON line says n_gpus 2 and names both ranks' device, and the
    PNG of the reduced frame is byte-identical to the 1-rank run's."""
    common = ["--steps", "2", "--warmup", "1", "--spp-per-step", "3", "--width", "320", "--height", "180", "--no-cpu-baseline", "--no-psnr"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    one = str(tmp_path / "one.png")
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--save-png", one] + common,
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    two = str(tmp_path / "two.png")
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--save-png", two] + common,
                        capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    assert p2.returncode == 0, p2.stderr[-2000:]
    j1 = json.loads([l for l in p1.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in p2.stdout.splitlines() if l.startswith("{")][-1])
    assert j1["n_gpus"] == 1 and j2["n_gpus"] == 2 and j2["value"] > 0
    assert [r["rank"] for r in j2["ranks"]] == [0, 1] and len({r["pid"] for r in j2["ranks"]}) == 2
    assert all(r["gcnArchName"] and r["gcnArchName"].startswith("gfx950") for r in j2["ranks"])
    assert j2["job"]["vertices_per_sample"] == j1["job"]["vertices_per_sample"]  # the same work, split over two ranks
    assert open(one, "rb").read() == open(two, "rb").read()


def test_bench_refuses_ranks_without_their_own_device(tmp_path):
    """Two ranks on a one-GPU box without --share-device: every rank must stop with a message, and bench.py with a non-zero code,
    instead of quietly rendering on one GPU and printing n_gpus 1."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a one-GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0", "--spp-per-step", "1",
                        "--width", "64", "--height", "64", "--no-cpu-baseline", "--no-psnr"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p.returncode != 0
    assert "GPU(s) visible" in p.stderr and not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_bench_one_rank_through_rccl(tmp_path):
    """bench.py as a rank of a one-rank job (`torch.distributed.run --nproc-per-node 1`), backend nccl: process-group set-up on RCCL, the
    gathering of the ranks' devices, `dist.reduce` of the device framebuffer and the max / sum all-reduces all execute -- with one rank,
    which is what a one-GPU box can run of that leg (RCCL refuses two ranks on one device).  The PNG equals the plain run's."""
    import socket
    common = ["--steps", "1", "--warmup", "1", "--spp-per-step", "4", "--width", "256", "--height", "144", "--no-cpu-baseline", "--no-psnr"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    one = str(tmp_path / "plain.png")
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--save-png", one] + common, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
    assert p1.returncode == 0, p1.stderr[-2000:]
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    two = str(tmp_path / "rccl.png")
    p2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                         os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl", "--save-png", two] + common,
                        capture_output=True, text=True, env=dict(env, MASTER_ADDR="127.0.0.1"), cwd=ROOT, timeout=900)
    assert p2.returncode == 0, p2.stderr[-3000:]
    j = json.loads([l for l in p2.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 1 and j["backend"] == "nccl" and len(j["ranks"]) == 1 and j["ranks"][0]["gcnArchName"].startswith("gfx950")
    assert open(one, "rb").read() == open(two, "rb").read()
