import sys, os, time; sys.path.insert(0, os.getcwd())
import torch, numpy as np, mcpt_loader
pkg = mcpt_loader.load()
sd = pkg.scenes.chess_scene(width=1920, height=1080, spp=256)
hs = pkg.HipScene(sd, device=0)
fb = torch.zeros(1920*1080*3, dtype=torch.float32, device='cuda'); st = torch.cuda.current_stream()
def run(cam, spp=512):
    hs.render_device(fb.data_ptr(), st.cuda_stream, camera=cam, spp=256, spp_per_pass=256)
    torch.cuda.synchronize(); t = time.perf_counter()
    s = hs.render_device(fb.data_ptr(), st.cuda_stream, camera=cam, spp=spp, spp_per_pass=256)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    return dt * 1e3 / (spp / 64), s
base, s0 = run(sd.camera)
sky = pkg.scenes.make_camera(1920, 1080, 70, (278, 150, -2550), (278, 5000, 0), (0, 0, 1), True, 3036.98, 10)
skyt, s1 = run(sky)
f = fb.cpu().numpy().reshape(1080, 1920, 3)
print("normal: %.2f ms per 64 spp; all-sky camera: %.2f ms per 64 spp (closest rays %d of %d samples, shaded %d)" % (base, skyt, s1.closest_rays, s1.samples, s1.shaded))
# fraction of pixels of the normal frame whose every sample missed: estimate with 64 samples per pixel via cast... use the frame: pixels equal to the background colour exactly
hs.render_device(fb.data_ptr(), st.cuda_stream, spp=64, spp_per_pass=64)
f = fb.cpu().numpy().reshape(1080, 1920, 3)
bg = np.asarray(sd.background, np.float32)
acc = np.zeros(3, np.float32)
for k in range(64): acc += bg / np.float32(64)
print("all-sky pixels (64 spp): %.3f" % (np.all(f == acc, axis=2).mean()))
