import sys, os; sys.path.insert(0, os.getcwd())
import mcpt_loader; pkg = mcpt_loader.load()
import numpy as np
lib = os.path.join(os.getcwd(), "final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd", "libmcpt_hip_stats.so")
for name, sd in (("chess", pkg.scenes.chess_scene(width=480, height=270, spp=16)), ("chess_high", pkg.scenes.chess_high(480, 270, 16))):
    for builder in ("sah", "lbvh", "ploc"):
        hs = pkg.HipScene(sd, library=lib, builder=builder)
        hs.render(spp=16, seed=1)
        c = hs.debug_counters().astype(np.float64)
        print(name, builder, "height %d | closest: rays %d, node visits/ray %.2f, prim tests/ray %.2f, hit %.3f, lane utilisation %.3f, deepest stack %d | shadow: rays %d, visits %.2f, tests %.2f, util %.3f, deepest stack %d"
              % (hs.info()["bvh_height"], c[0], c[1] / c[0], c[2] / c[0], c[3] / c[0], (c[1] + c[2]) / c[4], c[6], c[8], c[9] / max(c[8], 1), c[10] / max(c[8], 1), (c[9] + c[10]) / max(c[12], 1), c[7]))
        if c[15] > 0:
            print("    primary rays: %.1f different primitives hit per wave of 64 rays (waves with a hit: %d)" % (c[14] / c[15], c[15]))
        hs.close()
