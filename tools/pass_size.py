"""Frame time against the pass size (mcpt_params::spp_per_pass; 0 = the library's default) on one GPU and for one of eight ranks:
python tools/pass_size.py"""
import sys, time
sys.path.insert(0, '.')
import torch, mcpt_loader
pkg = mcpt_loader.load()
sd = pkg.scenes.chess_scene(width=1920, height=1080, spp=256)
hs = pkg.HipScene(sd, device=0)
fb = torch.zeros(1920 * 1080 * 3, dtype=torch.float32, device='cuda')
st = torch.cuda.current_stream()
def call(nranks, spp, sp):
    torch.cuda.synchronize(); t = time.perf_counter()
    s = hs.render_device(fb.data_ptr(), st.cuda_stream, spp=spp, spp_total=spp, accumulate=0, rank=0, nranks=nranks, spp_per_pass=sp)
    torch.cuda.synchronize(); return (time.perf_counter() - t) * 1e3, s.iterations
for n in (1, 8):
    for sp in (0, 32, 64, 128, 256, 512, 1024, 2048):
        call(n, 2048, sp)
        ms, it = min(call(n, 2048, sp) for _ in range(2))
        print("nranks=%d spp 2048, spp_per_pass=%4d: %8.2f ms, %4.0f iterations, %7.1f Msamples/s" % (n, sp, ms, it, 1920 * 1080 * 2048 / n / ms / 1e3), flush=True)
