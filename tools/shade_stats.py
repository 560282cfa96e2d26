import sys, os; sys.path.insert(0, os.getcwd())
import mcpt_loader; pkg = mcpt_loader.load()
lib = os.path.join(os.getcwd(), "final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd", "libmcpt_hip_stats.so")
for name, sd in (("chess", pkg.scenes.chess_scene(width=1920, height=1080, spp=64)), ("cornell_rc", pkg.scenes.cornell_rc(784, 784, 64))):
    hs = pkg.HipScene(sd, library=lib)
    hs.render(spp=64, seed=1, spp_per_pass=64)
    print(name, flush=True)
    hs.close()
