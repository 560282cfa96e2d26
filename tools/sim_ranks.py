import sys, time
sys.path.insert(0, '.')
import torch, mcpt_loader
pkg = mcpt_loader.load()
sd = pkg.scenes.chess_scene(width=1920, height=1080, spp=256)
hs = pkg.HipScene(sd, device=0)
fb = torch.zeros(1920*1080*3, dtype=torch.float32, device='cuda')
st = torch.cuda.current_stream()
PER_STEP = "--per-step-calls" in sys.argv
SCALE_PASS = "--scale-pass" in sys.argv  # spp per pass grows with the rank count (constant work per pass)
POOL = 0  # --pool-mi N: pool of N Mi paths instead of the library's default
for k, a in enumerate(sys.argv):
    if a == "--pool-mi":
        POOL = int(sys.argv[k + 1]) << 20
KW = dict(pool_paths=POOL) if POOL else {}
def run(nranks, steps=8):
    hs.render_device(fb.data_ptr(), st.cuda_stream, spp=256, spp_total=2048, accumulate=0, rank=0, nranks=nranks, spp_per_pass=256, **KW)
    torch.cuda.synchronize(); t=time.perf_counter(); its=0
    if PER_STEP:
        for k in range(steps):
            s = hs.render_device(fb.data_ptr(), st.cuda_stream, spp=256, spp_total=2048, sample_offset=k*256, accumulate=1, rank=0, nranks=nranks, spp_per_pass=256, **KW)
            its += s.iterations
    else:  # one call, passes pipelined inside the library (what bench.py does)
        its = hs.render_device(fb.data_ptr(), st.cuda_stream, spp=256*steps, spp_total=2048, accumulate=1, rank=0, nranks=nranks, spp_per_pass=(256*nranks if SCALE_PASS else 256), **KW).iterations
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/steps
    return dt*1e3, its/steps
base,_ = run(1)
for n in (1,2,4,8):
    ms, its = run(n)
    print("nranks=%d: %.1f ms/step (ideal %.1f), efficiency %.2f, iterations/step %.0f" % (n, ms, base/n, base/n/ms, its))
