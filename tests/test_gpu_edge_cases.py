"""Edge cases of the C ABI on the GPU, each checked against the CPU oracle: degenerate scenes (a single primitive, no
emitter, an emissive sphere), tiny and ragged frames, many ranks, deep Russian-roulette stacks and the overflow report."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene(pkg, tris=None, spheres=(), mats=None, cam=None, background=(0.2, 0.3, 0.4), rr=0.7):
    s = pkg.scenes
    b = s._Builder()
    P = s.material_presets()
    mats = mats or {}
    for name, m in mats.items():
        P[name] = m
    if tris is not None:
        for name, t in tris:
            b.add_mesh(t, b.material(name, P[name]))
    for name, c, r in spheres:
        b.add_sphere(c, r, b.material(name, P[name]))
    cam = cam if cam is not None else s.make_camera(24, 16, 50, (0, 0, -5), (0, 0, 0))
    sd = s.SceneData(triangles=np.concatenate(b.tris).astype(s.TRI_DTYPE) if b.tris else np.zeros(0, s.TRI_DTYPE),
                     materials=np.stack(b.mats).astype(s.MAT_DTYPE), objects=np.stack(b.objs).astype(s.OBJ_DTYPE),
                     background=np.asarray(background, np.float32), camera=cam, rr_rate=rr, spp=4)
    return sd


def _tri(pkg, v0, v1, v2):
    t = np.zeros(1, pkg.scenes.TRI_DTYPE)
    t["v0"], t["v1"], t["v2"] = v0, v1, v2
    return t


def _compare(pkg, oracle, hip, sd, spp=4, min_psnr=40.0, **kw):
    fb_ref, st_ref = oracle.OracleScene(sd).render(spp=spp, seed=2)
    fb_gpu, st_gpu = hip.HipScene(sd).render(spp=spp, seed=2, **kw)
    a, b = pkg.pngio.tonemap_u8(fb_ref), pkg.pngio.tonemap_u8(fb_gpu)
    psnr = pkg.pngio.psnr_u8(a, b)
    assert psnr >= min_psnr, "PSNR %.2f dB" % psnr
    return fb_ref, fb_gpu, st_ref, st_gpu


def test_single_sphere_no_light(pkg, oracle, hip):
    """One primitive: the scene BVH root is a leaf (BVH.cpp:36-41); no emitter: direct lighting adds nothing."""
    sd = _scene(pkg, spheres=[("rough_white_conductor", (0, 0, 0), 1.0)])
    fb_ref, fb_gpu, st_ref, st_gpu = _compare(pkg, oracle, hip, sd)
    assert st_gpu.shadow_rays == 0 and st_gpu.direct_vertices == 0
    assert (fb_gpu[0, 0] == np.float32([0.2, 0.3, 0.4])).all()  # a corner pixel sees the background colour


def test_single_triangle_and_emissive_sphere(pkg, oracle, hip):
    """An emissive sphere is a light object (Scene.hpp:106-108); Sphere::Sample leaves pos.emit unset (Sphere.hpp:64-74),
    so it lights nothing, but it is seen directly (Scene.cpp:102-107) and blocks indirect paths (Scene.cpp:135)."""
    light = pkg.scenes._mat(pkg.scenes.ROUGH_CONDUCTOR, emission=(5, 4, 3))
    sd = _scene(pkg, tris=[("rough_red_conductor", _tri(pkg, (-3, -1, 3), (3, -1, 3), (0, 2.5, 3)))],
                spheres=[("lamp", (0, 0.5, 0), 0.5)], mats={"lamp": light})
    fb_ref, fb_gpu, *_ = _compare(pkg, oracle, hip, sd, spp=8)
    assert fb_gpu.max() > 0.9 and fb_gpu.max() <= 1.0 + 1e-6  # the lamp is seen directly, clamped to 1 per channel (Scene.cpp:104)


def test_tiny_and_ragged_frames(pkg, oracle, hip):
    sd = pkg.scenes.cornell_rc(1, 1, 1)
    _compare(pkg, oracle, hip, sd, spp=1, min_psnr=30.0)
    sd = pkg.scenes.cornell_rc(37, 19, 3)  # not a multiple of the tile or the workgroup size
    fb_ref, fb_gpu, *_ = _compare(pkg, oracle, hip, sd, spp=3)
    hs = hip.HipScene(sd)
    parts = [hs.render(spp=3, seed=2, tile_size=8, rank=r, nranks=5)[0] for r in range(5)]
    assert np.array_equal(fb_gpu, sum(parts[1:], parts[0]))
    # more ranks than tiles: the extra ranks own nothing and return a zero frame
    empty, st = hs.render(spp=3, seed=2, tile_size=64, rank=3, nranks=4)
    assert not empty.any() and st.samples == 0


def test_deep_roulette_and_overflow_report(pkg, oracle, hip):
    """RussianRouletteRate 0.99 (the clamp of Scene::setRrRate): paths hundreds of vertices deep exercise the clamp stack;
    with a tiny max_depth the library reports MCPT_ERR_OVERFLOW instead of silently truncating."""
    sd = pkg.scenes.cornell_demo(24, 24, 2)
    sd.rr_rate = float(np.float32(0.99))
    fb_ref, fb_gpu, st_ref, st_gpu = _compare(pkg, oracle, hip, sd, spp=2, min_psnr=35.0)
    assert st_gpu.vertices / st_gpu.paths > 3 and st_gpu.overflow_paths == 0
    assert abs(st_gpu.vertices - st_ref.vertices) <= 0.03 * st_ref.vertices  # a single diverged long path moves this by ~1 %
    with pytest.raises(hip.McptError) as ei:
        hip.HipScene(sd).render(spp=2, seed=2, max_depth=3)
    assert ei.value.code == 5


def test_empty_inputs(pkg, hip):
    sd = pkg.scenes.cornell_rc(16, 16, 1)
    hs = hip.HipScene(sd)
    t, p = hs.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))
    assert len(t) == 0 and len(p) == 0
    out = hs.cast_rays(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), [], [], [])
    assert len(out) == 0
    with pytest.raises(hip.McptError):
        hs.cast_rays(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32), [0], [0], [7])  # channel out of range


def test_random_configurations_do_not_depend_on_the_schedule(pkg, hip, hip_check, monkeypatch):
    """Random frame sizes, spp, light samples, roulette rates and pass sizes: the frame and the work counters are the same
    with one stream, wait-then-launch, float nodes, a ring counter that wraps and pools of a few hundred paths; and an
    odd number of ranks partitions it exactly."""
    rng = np.random.default_rng(3)
    for _ in range(6):
        scene = str(rng.choice(["cornell_demo", "cornell_rc", "chess"]))
        w, h, spp = int(rng.integers(1, 90)), int(rng.integers(1, 70)), int(rng.integers(1, 20))
        sd = pkg.scenes.chess_scene(width=w, height=h, spp=spp) if scene == "chess" else getattr(pkg.scenes, scene)(w, h, spp)
        sd.rr_rate = float(rng.choice([0.2, 0.4, 0.7, 0.95]))
        kw = dict(spp=spp, seed=int(rng.integers(0, 1000)), n_dir_sample=int(rng.choice([1, 2, 4, 5, 9])), spp_per_pass=int(rng.integers(1, spp + 1)))
        ref, st0 = hip.HipScene(sd).render(**kw)
        for env, extra in [({"MCPT_OVERLAP": "0"}, {}), ({"MCPT_QUEUE_AHEAD": "0"}, {"pool_paths": 3 * 256}),
                           ({"MCPT_QUANT_NODES": "0"}, {"pool_paths": 3 * 1024}), ({"MCPT_RING_START": "0xffffff00"}, {"pool_paths": 3 * 512})]:
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            hook = "MCPT_RING_START" in env  # a test hook: only in the checking build
            fb, st = hip.HipScene(sd, library=hip_check if hook else None).render(**kw, **extra)
            for k in env:
                monkeypatch.delenv(k)
            assert np.array_equal(ref, fb, equal_nan=True), (scene, w, h, kw, env, extra)
            assert (st.vertices, st.shaded) == (st0.vertices, st0.shaded)
            assert hook or st.shadow_rays == st0.shadow_rays
        nr, ts = int(rng.choice([2, 3, 5, 7])), int(rng.choice([4, 8, 32]))
        hs = hip.HipScene(sd)
        parts = [hs.render(**kw, rank=r, nranks=nr, tile_size=ts)[0] for r in range(nr)]
        assert np.array_equal(ref, sum(parts[1:], parts[0]), equal_nan=True), (scene, w, h, kw, nr, ts)


def test_pool_shrinks_to_the_free_memory(pkg, hip, hip_check, monkeypatch):
    """The wavefront pool is sized from hipMemGetInfo: with little free memory (test hook of the checking build: MCPT_FAKE_FREE_MB) the
    same frame is rendered with a smaller pool -- more wavefront iterations, identical pixels -- instead of failing to allocate."""
    sd = pkg.scenes.chess_scene(width=480, height=270, spp=16)
    ref, st0 = hip.HipScene(sd, library=hip_check).render(spp=16, seed=8, spp_per_pass=16)
    monkeypatch.setenv("MCPT_FAKE_FREE_MB", "200")  # the default pool for this frame wants ~6 GB
    fb, st = hip.HipScene(sd, library=hip_check).render(spp=16, seed=8, spp_per_pass=16)
    assert np.array_equal(ref, fb, equal_nan=True)
    assert st.iterations > 2 * st0.iterations and (st.vertices, st.shaded) == (st0.vertices, st0.shaded)
