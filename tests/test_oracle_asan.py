"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (scene construction, recursive castRay, both BVH levels, light
sampling, the environment lookup, degenerate rays, destruction): the checker itself must be memory-clean."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan(pkg, tmp_path):
    exe = str(tmp_path / "oracle_asan")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           os.path.join(ROOT, "tests", "native", "oracle_driver.c"), os.path.join(ROOT, "oracle", "mcpt_oracle.c"), "-o", exe, "-lm"])
    rng = np.random.default_rng(0)
    env = rng.random((8, 16, 3)).astype(np.float32)
    scenes = [("cornell_demo", pkg.scenes.cornell_demo(24, 16, 3), None), ("chess", pkg.scenes.chess_scene(width=24, height=16, spp=3), None),
              ("chess_env", pkg.scenes.chess_scene(width=20, height=12, spp=3), env)]
    for name, sd, e in scenes:
        path = str(tmp_path / (name + ".bin"))
        with open(path, "wb") as fh:
            eh, ew = (e.shape[0], e.shape[1]) if e is not None else (0, 0)
            fh.write(np.array([len(sd.triangles), len(sd.materials), len(sd.objects), ew, eh], np.int32).tobytes())
            fh.write(np.ascontiguousarray(sd.triangles).tobytes())
            fh.write(np.ascontiguousarray(sd.materials).tobytes())
            fh.write(np.ascontiguousarray(sd.objects).tobytes())
            fh.write(np.ascontiguousarray(sd.camera).tobytes())
            fh.write(np.float32(sd.rr_rate).tobytes())
            if e is not None:
                fh.write(e.tobytes())
        p = subprocess.run([exe, path], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1", OMP_NUM_THREADS="2"))
        assert p.returncode == 0 and "samples" in p.stdout, (name, p.stdout[-500:], p.stderr[-4000:])
