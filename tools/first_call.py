"""First call of mcpt_render against the following ones (workspace allocation, first touch): python tools/first_call.py"""
import sys, time
sys.path.insert(0, '.')
import numpy as np, mcpt_loader
pkg = mcpt_loader.load()
sd = pkg.scenes.chess_scene(width=1920, height=1080, spp=2048)
t = time.perf_counter(); hs = pkg.HipScene(sd, device=0); print("scene create %.1f ms" % ((time.perf_counter() - t) * 1e3))
for k in range(3):
    t = time.perf_counter(); fb, st = hs.render(spp=2048, seed=1); dt = (time.perf_counter() - t) * 1e3
    print("render call %d: %.1f ms wall (host framebuffer), kernels' own ms: %s" % (k, dt, {n: round(getattr(st, n), 1) for n in dir(st) if n.startswith('ms_')}), flush=True)
