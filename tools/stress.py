"""Randomised consistency run (not part of the test suite): random frame sizes, spp, light samples, roulette rates, pass sizes, pools,
builders (host SAH / reference topology / GPU LBVH / GPU PLOC), instancing, culling, the LDS-resident small-scene flavour on and off, rank
counts -- every variant must render the frame of the plain configuration of the same scene and parameters.  python tools/stress.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, mcpt_loader
pkg = mcpt_loader.load()


def run(budget, seed, verbose=True):
    """Renders random configurations for `budget` seconds; returns (variant renders, mismatches)."""
    rng = np.random.default_rng(seed)
    t0, n, bad = time.time(), 0, 0
    t_report = t0
    while time.time() - t0 < budget:
        scene = str(rng.choice(["cornell_demo", "cornell_rc", "chess", "chess"]))
        w, h, spp = int(rng.integers(8, 400)), int(rng.integers(8, 260)), int(rng.integers(1, 24))
        sd = pkg.scenes.chess_scene(width=w, height=h, spp=spp) if scene == "chess" else getattr(pkg.scenes, scene)(w, h, spp)
        sd.rr_rate = float(rng.choice([0.2, 0.4, 0.7, 0.9]))
        if rng.random() < 0.3:
            sd.camera["use_dof"] = int(rng.integers(0, 2))
        kw = dict(spp=spp, seed=int(rng.integers(0, 1000)), n_dir_sample=int(rng.choice([1, 3, 4, 8])), spp_per_pass=int(rng.integers(0, spp + 1)))  # (0: the library's choice)
        os.environ.pop("MCPT_SKY_CULL", None)
        os.environ.pop("MCPT_SMALL_SCENE", None)
        ref, st0 = pkg.HipScene(sd, builder="sah", instancing=False).render(**kw)
        variants = [dict(builder="lbvh"), dict(builder="ploc"), dict(builder="sah", quantise=0), dict(builder="sah", instancing=True), dict(builder="reference"),
                    dict(builder="ploc", quantise=0)]
        for v in variants:
            extra = dict(pool_paths=int(rng.choice([0, 3 * 256, 3 * 4096, 3 * 65536])))
            for knob in ("MCPT_SKY_CULL", "MCPT_SMALL_SCENE"):
                if rng.random() < 0.5:
                    os.environ[knob] = "0"
                else:
                    os.environ.pop(knob, None)
            hs = pkg.HipScene(sd, **v)
            nr = int(rng.choice([1, 1, 2, 3, 5]))
            if nr == 1:
                fb, st = hs.render(**kw, **extra)
            else:
                ts = int(rng.choice([4, 16, 32]))
                parts = [hs.render(**kw, **extra, rank=r, nranks=nr, tile_size=ts)[0] for r in range(nr)]
                fb = sum(parts[1:], parts[0])
            differing = int((~((fb == ref) | (np.isnan(fb) & np.isnan(ref)))).sum())
            n += 1
            if differing > 3:
                bad += 1
                if verbose:
                    print("MISMATCH", scene, w, h, kw, v, extra, nr, differing, flush=True)
            hs.close()
        if verbose and time.time() - t_report > 60:  # (a sign of life: the GPU pool's watchdog ends a silent run after seven minutes)
            t_report = time.time()
            print("stress: %d variant renders so far, %d mismatches" % (n, bad), flush=True)
    os.environ.pop("MCPT_SKY_CULL", None)
    os.environ.pop("MCPT_SMALL_SCENE", None)
    return n, bad


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    t0 = time.time()
    n, bad = run(budget, int(time.time()) & 0xffff)
    print("stress: %d variant renders in %.0f s, %d mismatches" % (n, time.time() - t0, bad))
