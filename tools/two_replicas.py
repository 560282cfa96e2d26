"""Does driving one GPU from two independent wavefront pipelines pay?  mcpt_group with the device listed twice = two host threads, each
with its own scene replica, pool and streams, on interleaved tiles of the same frame; against one pipeline."""
import sys, os, time; sys.path.insert(0, os.getcwd())
import numpy as np, mcpt_loader
pkg = mcpt_loader.load()
sd = pkg.scenes.chess_scene(width=1920, height=1080, spp=256)
def timed(obj, **kw):
    obj.render(spp=256, seed=1, spp_per_pass=256, **kw)
    t = time.perf_counter(); fb, st = obj.render(spp=1024, seed=1, spp_per_pass=256, **kw); dt = time.perf_counter() - t
    return st.samples / dt / 1e6, fb
one, f1 = timed(pkg.HipScene(sd))
for n in (2, 3):
    g = pkg.HipGroup(sd, [0] * n)
    v, f2 = timed(g)
    print("one pipeline %.0f Msamples/s (incl. download); %d replicas on the same GPU %.0f; identical %s" % (one, n, v, np.array_equal(f1, f2)))
    g.close()
