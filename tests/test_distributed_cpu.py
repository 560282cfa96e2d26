"""The N>1 path on CPU: world_size-2 gloo.  Each rank renders its interleaved-tile share of the frame
(the partition rule of include/mcpt.h: mcpt_params.tile_size/rank/nranks) and the frames are summed
to rank 0 with one reduce -- exactly what bench.py does with RCCL.  The renderer here is the CPU oracle (tests
may use it); the GPU side of the same rule is covered by test_gpu_parity.py::test_tile_partition_*."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    import mcpt_loader
    from oracle import oracle
    pkg = mcpt_loader.load()
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    sd = pkg.scenes.cornell_rc(64, 48, 2)
    fb = np.zeros((48, 64, 3), dtype=np.float32)
    oracle.OracleScene(sd).render(fb=fb, spp=2, seed=5, n_threads=2, tile_size=16, rank=rank, nranks=world)
    t = torch.from_numpy(fb)
    owned = int((fb != 0).any(axis=2).sum())
    dist.reduce(t, dst=0)
    counts = torch.tensor([owned], dtype=torch.int64)
    dist.all_reduce(counts)
    if rank == 0:
        np.save(out_path, np.concatenate([t.numpy().ravel(), [float(counts.item())]]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_tile_partition_and_reduce(tmp_path, pkg, oracle):
    import torch.multiprocessing as mp
    out = str(tmp_path / "fb.npy")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    fb2, owned = got[:-1].reshape(48, 64, 3).astype(np.float32), int(got[-1])
    sd = pkg.scenes.cornell_rc(64, 48, 2)
    fb1, _ = oracle.OracleScene(sd).render(spp=2, seed=5)
    assert np.array_equal(fb1, fb2)  # disjoint tiles + identical Philox keys: bit-identical to the 1-rank frame
    assert 0 < owned <= 64 * 48  # every lit pixel was produced by exactly one rank


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher and no WORLD_SIZE: the parent must start two ranks under torch.distributed.run (as a
    child process, before importing torch) instead of running one rank and printing n_gpus 1.  Without a GPU every rank stops at the
    "needs a GPU" check (the product has no CPU fallback), so the run fails -- with both ranks' messages and no JSON line."""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("CPU-only check (the GPU version is tests/test_gpu_multi.py)")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert p.returncode != 0
    assert "starting 2 ranks" in p.stderr and "--nproc-per-node 2" in p.stderr
    assert "bench.py needs a GPU" in p.stderr  # (from at least one rank: the launcher stops the others as soon as the first one fails)
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    # one rank needs no launcher and says the same
    p1 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
    assert p1.returncode != 0 and "starting" not in p1.stderr and "bench.py needs a GPU" in p1.stderr
