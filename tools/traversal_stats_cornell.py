import sys, os; sys.path.insert(0, os.getcwd())
import mcpt_loader; pkg = mcpt_loader.load()
import numpy as np
lib = os.path.join(os.getcwd(), "final-project-monte-carlo-path-tracer-with-microfacet-bsdf_amd", "libmcpt_hip_stats.so")
for name, sd in (("cornell_rc", pkg.scenes.cornell_rc(392, 392, 16)), ("cornell_demo", pkg.scenes.cornell_demo(480, 270, 16))):
    for small in ("1", "0"):
        os.environ["MCPT_SMALL_SCENE"] = small
        hs = pkg.HipScene(sd, library=lib)
        hs.render(spp=16, seed=1)
        c = hs.debug_counters().astype(np.float64)
        print(name, "lds_resident", hs.info()["lds_resident"], "height %d | closest: rays %d, node visits/ray %.2f, prim tests/ray %.2f, hit %.3f, lane utilisation %.3f, deepest stack %d | shadow: rays %d, visits %.2f, tests %.2f, util %.3f, deepest stack %d"
              % (hs.info()["bvh_height"], c[0], c[1] / c[0], c[2] / c[0], c[3] / c[0], (c[1] + c[2]) / c[4], c[6], c[8], c[9] / max(c[8], 1), c[10] / max(c[8], 1), (c[9] + c[10]) / max(c[12], 1), c[7]))
        hs.close()
