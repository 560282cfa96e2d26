"""The checking build (libmcpt_hip_check.so, -DMCPT_CHECK_DIRECT_SKIP): every vertex whose direct lighting the product skips
as "provably zero" (direct_is_zero in csrc/mcpt_kernels.hip: no emitter / conductor seen from inside / Dirac BSDF whose mirror or
Snell direction misses the cone of the emitters' bounding sphere) is evaluated anyway, and a non-zero light sample among them
is counted.  The count must be 0 on the three shipped scenes and on a scene built to sit on the rule's edges
(Material.hpp:379-403: a Dirac eval is non-zero only within acos(1 - 1e-4) of the mirror / Snell direction)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _quad(pkg, a, b, c, d):
    t = np.zeros(2, pkg.scenes.TRI_DTYPE)
    t["v0"], t["v1"], t["v2"] = [a, a], [b, c], [c, d]
    return t


def adversarial_scene(pkg, w=96, h=64):
    """A 1 x 1 area light (bounding-sphere radius R = 0.708) with Dirac surfaces from just outside the rule's distance gate
    (D > 1.01 R) outwards: a mirror 0.73 below the light centre (D/R from 1.03 at its middle), a tilted mirror whose reflections
    of the camera rays sweep across the light's rim, a glass sphere almost touching the light, a glass slab under it (refraction
    from inside, the 0.15 rad rule), and a rough floor that sends paths everywhere else."""
    s = pkg.scenes
    P = s.material_presets()
    b = s._Builder()
    light = s._mat(s.ROUGH_CONDUCTOR, emission=(40, 35, 30))
    y = 2.0
    b.add_mesh(_quad(pkg, (-0.5, y, -0.5), (0.5, y, -0.5), (0.5, y, 0.5), (-0.5, y, 0.5)), b.material("light", light))
    b.add_mesh(_quad(pkg, (-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6)), b.material("rough_white_conductor", P["rough_white_conductor"]))
    ym = y - 0.73
    b.add_mesh(_quad(pkg, (-0.4, ym, -0.4), (-0.4, ym, 0.4), (0.4, ym, 0.4), (0.4, ym, -0.4)), b.material("silver_mirror", P["silver_mirror"]))
    b.add_mesh(_quad(pkg, (1.0, 0.2, -1.0), (1.0, 0.2, 1.0), (2.2, 1.6, 1.0), (2.2, 1.6, -1.0)), b.material("gold_conductor", P["gold_conductor"]))
    b.add_sphere((-1.05, 1.9, 0.0), 0.5, b.material("smooth_glass", P["smooth_glass"]))
    b.add_sphere((0.0, 0.45, 1.2), 0.45, b.material("smooth_glass_gem", P["smooth_glass_gem"]))
    g0, g1 = 0.9, 1.1  # glass slab (two faces) between the floor and the light
    b.add_mesh(np.concatenate([_quad(pkg, (-0.8, g1, -0.8), (-0.8, g1, 0.8), (0.8, g1, 0.8), (0.8, g1, -0.8)),
                               _quad(pkg, (-0.8, g0, -0.8), (0.8, g0, -0.8), (0.8, g0, 0.8), (-0.8, g0, 0.8))]),
               b.material("smooth_glass", P["smooth_glass"]))
    cam = s.make_camera(w, h, 55, (0.3, 1.3, -4.5), (0.2, 1.1, 0.0))
    return s.SceneData(triangles=np.concatenate(b.tris).astype(s.TRI_DTYPE), materials=np.stack(b.mats).astype(s.MAT_DTYPE),
                       objects=np.stack(b.objs).astype(s.OBJ_DTYPE), background=np.float32([0.05, 0.05, 0.08]), camera=cam,
                       rr_rate=0.8, spp=16, name="adversarial")


@pytest.mark.parametrize("name", ["cornell_demo", "cornell_rc", "chess", "adversarial"])
def test_skipped_direct_lighting_is_exactly_zero(pkg, hip, hip_check, oracle, name):
    sd = {"cornell_demo": lambda: pkg.scenes.cornell_demo(128, 128, 16), "cornell_rc": lambda: pkg.scenes.cornell_rc(96, 96, 8),
          "chess": lambda: pkg.scenes.chess_scene(width=320, height=180, spp=16), "adversarial": lambda: adversarial_scene(pkg)}[name]()
    spp = 32 if name == "adversarial" else int(sd.spp)
    hc = hip.HipScene(sd, library=hip_check)
    assert b"checking build" in hc.L.mcpt_version()
    fb_check, st_check = hc.render(spp=spp, seed=3)
    c = hc.debug_counters()
    skipped, nonzero = int(c[14]), int(c[15])
    print("\n[direct-skip check] %s: %d light samples at skipped vertices, %d non-zero" % (name, skipped, nonzero))
    assert nonzero == 0
    if name != "cornell_rc":  # (rough conductors only, never seen from inside: nothing is skipped there)
        assert skipped > 1000
    # the product build skips those vertices and renders the same frame
    fb, st = hip.HipScene(sd).render(spp=spp, seed=3)
    assert np.array_equal(fb, fb_check, equal_nan=True)
    assert st.direct_vertices < st_check.direct_vertices or skipped == 0
    assert hip.HipScene(sd).debug_counters().sum() == 0  # the product build counts nothing
    if name == "adversarial":  # and the scene itself is rendered correctly
        ref, _ = oracle.OracleScene(sd).render(spp=8, seed=3)
        gpu, _ = hip.HipScene(sd).render(spp=8, seed=3)
        assert pkg.pngio.psnr_u8(pkg.pngio.tonemap_u8(ref), pkg.pngio.tonemap_u8(gpu)) >= 60.0


@pytest.mark.parametrize("builder", ["sah", "lbvh", "reference"])
def test_retry_flavour_of_the_traversal_stack(pkg, hip, hip_check, builder):
    """Trees deeper than 24 levels are traversed with 16 stack entries in LDS; a ray that would need more loses an entry, is marked and
    put on the kernel's retrace list, and a small kernel launched right behind traces the listed rays again with a per-lane stack in
    scratch memory (MCPT_STK_PUSH / RetryList / k_retrace_* in csrc/mcpt_kernels.hip) -- which no ray of these scenes needs.  The
    checking build uses that flavour for every tree with FOUR LDS entries, so most of its rays (primary, continuation and shadow) go
    through the lists: same intersections, same frames, same counters."""
    rng = np.random.default_rng(11)
    for sd in (pkg.scenes.chess_scene(width=160, height=90, spp=4), pkg.scenes.chess_high(160, 90, 4), pkg.scenes.cornell_demo(64, 64, 4)):
        prod, chk = hip.HipScene(sd, builder=builder), hip.HipScene(sd, library=hip_check, builder=builder)
        assert prod.info()["bvh_height"] == chk.info()["bvh_height"]
        n = 30000
        w, h = int(sd.camera["width"]), int(sd.camera["height"])
        o, d = prod.camera_rays(rng.integers(0, w * h, n).astype(np.uint32), rng.integers(0, 64, n).astype(np.uint32), seed=3)
        d = d.copy()
        d[: n // 10, 1] = 0.0  # (a zero component: the NaN-faithful slab test, generic loop)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        a, b = prod.intersect(o, d), chk.intersect(o, d)
        for x, y in zip(a, b):
            assert np.array_equal(x, y, equal_nan=True)
        fa, sa = prod.render(spp=4, seed=5)
        fb, sb = chk.render(spp=4, seed=5)
        assert np.array_equal(fa, fb, equal_nan=True) and sa.vertices == sb.vertices and sa.shadow_rays == sb.shadow_rays
        prod.close()
        chk.close()


def _random_dirac_scene(pkg, rng):
    """Random emitters (1-3 quads and, sometimes, an emissive sphere) with random mirrors, glass spheres and glass slabs scattered around
    and close to them, over a rough floor: the configurations direct_is_zero's cone rule has to classify."""
    s = pkg.scenes
    P = s.material_presets()
    b = s._Builder()
    for k in range(int(rng.integers(1, 4))):
        c = rng.uniform([-3, 1.5, -3], [3, 4, 3])
        e = rng.uniform(0.2, 1.2)
        u, v = rng.normal(size=3), rng.normal(size=3)
        u /= np.linalg.norm(u)
        v -= u * (u @ v)
        v /= np.linalg.norm(v)
        light = s._mat(s.ROUGH_CONDUCTOR, emission=tuple(rng.uniform(10, 60, 3)))
        b.add_mesh(_quad(pkg, c - e * u - e * v, c + e * u - e * v, c + e * u + e * v, c - e * u + e * v), b.material("light%d" % k, light))
    b.add_mesh(_quad(pkg, (-8, 0, -8), (-8, 0, 8), (8, 0, 8), (8, 0, -8)), b.material("rough_white_conductor", P["rough_white_conductor"]))
    for k in range(int(rng.integers(3, 9))):
        c = rng.uniform([-4, 0.2, -4], [4, 4.5, 4])
        kind = int(rng.integers(0, 3))
        if kind == 0:  # a mirror quad with a random orientation
            e = rng.uniform(0.3, 1.5)
            u, v = rng.normal(size=3), rng.normal(size=3)
            u /= np.linalg.norm(u)
            v -= u * (u @ v)
            v /= np.linalg.norm(v)
            name = str(rng.choice(["silver_mirror", "gold_conductor"]))
            b.add_mesh(_quad(pkg, c - e * u - e * v, c + e * u - e * v, c + e * u + e * v, c - e * u + e * v), b.material(name, P[name]))
        elif kind == 1:
            name = str(rng.choice(["smooth_glass", "smooth_glass_gem", "silver_mirror"]))
            b.add_sphere(tuple(c), float(rng.uniform(0.15, 0.9)), b.material(name, P[name]))
        else:  # a thin glass slab: two faces, refraction from inside
            e, th = rng.uniform(0.4, 1.2), rng.uniform(0.05, 0.4)
            top, bot = c[1] + th, c[1]
            b.add_mesh(np.concatenate([_quad(pkg, (c[0] - e, top, c[2] - e), (c[0] - e, top, c[2] + e), (c[0] + e, top, c[2] + e), (c[0] + e, top, c[2] - e)),
                                       _quad(pkg, (c[0] - e, bot, c[2] - e), (c[0] + e, bot, c[2] - e), (c[0] + e, bot, c[2] + e), (c[0] - e, bot, c[2] + e))]),
                       b.material("smooth_glass", P["smooth_glass"]))
    cam = s.make_camera(80, 56, 60, tuple(rng.uniform([-2, 1, -7], [2, 3, -5])), (0.0, 1.5, 0.0))
    return s.SceneData(triangles=np.concatenate(b.tris).astype(s.TRI_DTYPE), materials=np.stack(b.mats).astype(s.MAT_DTYPE),
                       objects=np.stack(b.objs).astype(s.OBJ_DTYPE), background=np.float32([0.02, 0.02, 0.03]), camera=cam,
                       rr_rate=float(rng.choice([0.5, 0.8])), spp=8, name="random dirac scene")


def test_skipped_direct_lighting_is_zero_on_random_scenes(pkg, hip, hip_check):
    """direct_is_zero on geometry nobody designed: 24 random scenes of mirrors and glass around random emitters, rendered by the checking
    build (which evaluates every skipped vertex anyway): not one non-zero contribution among the skipped light samples; and the product
    build renders the same frames."""
    rng = np.random.default_rng(2024)
    tot_skipped = 0
    for k in range(24):
        sd = _random_dirac_scene(pkg, rng)
        hc = hip.HipScene(sd, library=hip_check)
        fb_check, _ = hc.render(spp=8, seed=k)
        c = hc.debug_counters()
        assert int(c[15]) == 0, (k, int(c[14]), int(c[15]))
        tot_skipped += int(c[14])
        fb, _ = hip.HipScene(sd).render(spp=8, seed=k)
        assert np.array_equal(fb, fb_check, equal_nan=True), k
        hc.close()
    print("\n[direct-skip check] 24 random scenes: %d light samples at skipped vertices, 0 non-zero" % tot_skipped)
    assert tot_skipped > 100000


def test_random_variants_render_the_plain_frame(pkg, hip, monkeypatch):
    """A bounded slice of tools/stress.py inside the suite: 20 seconds of random configurations (frame size, spp, light samples, roulette
    rate, pass size, pool size, rank count) under every builder (host SAH float / quantised / instanced, reference topology, GPU LBVH, GPU
    PLOC) with the sky cull and the LDS-resident flavour switched at random: every variant renders the plain configuration's frame (at
    most 3 box-grazing values apart).  The full run (`python tools/stress.py 240`: 1752 renders, 0 mismatches) is in profiles/r03_stress.txt."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("mcpt_stress", os.path.join(root, "tools", "stress.py"))
    mod = importlib.util.module_from_spec(spec)
    monkeypatch.chdir(root)
    spec.loader.exec_module(mod)
    n, bad = mod.run(20.0, 7)
    print("\n[stress] %d variant renders, %d mismatches" % (n, bad))
    assert n >= 30 and bad == 0
