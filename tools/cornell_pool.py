import sys, time
sys.path.insert(0, '.')
import mcpt_loader
pkg = mcpt_loader.load()
sd = pkg.scenes.cornell_rc(784, 784, 256)
hs = pkg.HipScene(sd, device=0)
for kw in (dict(), dict(max_depth=64), dict(pool_paths=10 << 20), dict(max_depth=64, pool_paths=30 << 20)):
    hs.render(spp=256, seed=1, **kw)
    t = time.perf_counter(); fb, st = hs.render(spp=256, seed=1, **kw); dt = (time.perf_counter() - t) * 1e3
    print(kw, "%.1f ms, %d iterations, %.1f Msamples/s, overflow %d" % (dt, st.iterations, 784 * 784 * 256 / dt / 1e3, st.overflow_paths), flush=True)
