"""The LDS-resident flavour for scenes of a few KB (the Cornell configurations: DevScene::small, SMALL kernels in csrc/mcpt_kernels.hip).
It may not change a result: same tree, same tests, same arithmetic -- only where the bytes are read from.  Frames, path values, hits and
work counters must be identical with the flavour on and off (MCPT_SMALL_SCENE=0) and identical to the oracle; a scene just beyond the
limits must fall back by itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

COUNTERS = ("samples", "vertices", "shaded", "closest_rays", "shadow_rays", "direct_vertices", "ref_scene_rays")


def _render(hip, sd, monkeypatch, small, **kw):
    if small:
        monkeypatch.delenv("MCPT_SMALL_SCENE", raising=False)
    else:
        monkeypatch.setenv("MCPT_SMALL_SCENE", "0")
    hs = hip.HipScene(sd)
    assert hs.info()["lds_resident"] == (1 if small else 0)
    return hs.render(**kw)


@pytest.mark.parametrize("scene", ["cornell_demo", "cornell_rc", "cornell_demo_float_nodes", "cornell_demo_reference_tree"])
def test_small_scene_flavours_render_the_same_frame(pkg, oracle, hip, monkeypatch, scene):
    if scene == "cornell_rc":
        sd = pkg.scenes.cornell_rc(96, 96, 6)
    else:
        sd = pkg.scenes.cornell_demo(96, 80, 6)
    if scene.endswith("float_nodes"):
        monkeypatch.setenv("MCPT_QUANT_NODES", "0")  # 64-byte nodes in LDS
    if scene.endswith("reference_tree"):
        monkeypatch.setenv("MCPT_BVH", "reference")
    kw = dict(spp=6, seed=11, spp_per_pass=4)
    base, sb = _render(hip, sd, monkeypatch, False, **kw)
    fb, st = _render(hip, sd, monkeypatch, True, **kw)
    assert np.array_equal(base, fb, equal_nan=True), int((base != fb).sum())
    assert all(getattr(st, k) == getattr(sb, k) for k in COUNTERS), [(k, getattr(st, k), getattr(sb, k)) for k in COUNTERS]
    ref, so = oracle.OracleScene(sd).render(spp=6, seed=11)
    differing = int((~((base == ref) | (np.isnan(base) & np.isnan(ref)))).sum())
    print("\n[small] %s: %d of %d framebuffer values differ from the oracle" % (scene, differing, ref.size))
    assert differing <= (0 if scene.endswith("reference_tree") else 3)
    assert sb.vertices == so.vertices


def test_small_scene_hits_and_paths(pkg, oracle, hip, monkeypatch):
    """mcpt_intersect and mcpt_cast_rays through the SMALL kernels: bit-identical to the oracle's full traversal."""
    sd = pkg.scenes.cornell_demo(64, 64, 1)
    rng = np.random.default_rng(3)
    n = 20000
    o = rng.uniform([10, 10, -700], [540, 540, 540], (n, 3)).astype(np.float32)
    d = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[:50, 0] = 0.0  # zero direction components: the NaN-faithful slab chain; a run of 100 such rays inside one wave's chunk of the
    d[50:100, 1] = 0.0  # refill kernel, so that refills hand out nothing else (this lost the rest of the chunk before round 3)
    hs = hip.HipScene(sd)
    assert hs.info()["lds_resident"] == 1
    t, prim = hs.intersect(o, d)
    to, po = oracle.OracleScene(sd).intersect(o, d)
    assert np.array_equal(prim, po) and np.array_equal(t[prim >= 0], to[po >= 0])
    px = rng.integers(0, 64 * 64, n).astype(np.uint32)
    sm = rng.integers(0, 64, n).astype(np.uint32)
    ch = rng.integers(0, 3, n).astype(np.int32)
    a = hs.cast_rays(o, d, px, sm, ch, seed=5)
    b = oracle.OracleScene(sd).cast_rays(o, d, px, sm, ch, seed=5)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert (~same).sum() <= 2, int((~same).sum())


def test_without_shadows_and_with_many_light_samples(pkg, hip, monkeypatch):
    """includeShadow = false runs no shadow query; n_dir 9 takes the generic index arithmetic."""
    sd = pkg.scenes.cornell_demo(64, 48, 4)
    for n_dir, shadow in ((9, True), (4, False)):
        sd.enable_shadow = shadow
        a, sa = _render(hip, sd, monkeypatch, True, spp=4, seed=2, n_dir_sample=n_dir)
        b, sb = _render(hip, sd, monkeypatch, False, spp=4, seed=2, n_dir_sample=n_dir)
        assert np.array_equal(a, b, equal_nan=True) and sa.shadow_rays == sb.shadow_rays
        assert (sa.shadow_rays == 0) == (not shadow)


@pytest.mark.parametrize("n_dir", [8, 12])
def test_light_sample_counts_that_are_multiples_of_four(pkg, oracle, hip, monkeypatch, n_dir):
    """n_dir 8 (a power of two: k_direct splits its index with a shift) and 12 (k_shade sums the contributions from 16-byte loads, in the
    reference's order): the oracle's frame."""
    sd = pkg.scenes.cornell_demo(64, 48, 3)
    fb, st = _render(hip, sd, monkeypatch, True, spp=3, seed=4, n_dir_sample=n_dir)
    ref, so = oracle.OracleScene(sd).render(spp=3, seed=4, n_dir_sample=n_dir)
    differing = int((~((fb == ref) | (np.isnan(fb) & np.isnan(ref)))).sum())
    assert differing <= 3 and st.vertices == so.vertices and st.direct_vertices * n_dir >= st.shadow_rays > 0


def test_grid_caps_change_no_result(pkg, hip, monkeypatch):
    """k_trace_shadow and k_direct stride over their queues with whatever grid they get (MCPT_SHADOW_GRID_PER_CU, MCPT_DIRECT_GRID_PER_CU:
    one workgroup per CU here, i.e. every workgroup walks many chunks)."""
    for scene in (pkg.scenes.cornell_demo(160, 120, 4), pkg.scenes.chess_scene(width=160, height=90, spp=4)):
        monkeypatch.delenv("MCPT_SHADOW_GRID_PER_CU", raising=False)
        monkeypatch.delenv("MCPT_DIRECT_GRID_PER_CU", raising=False)
        a, sa = hip.HipScene(scene).render(spp=4, seed=9)
        monkeypatch.setenv("MCPT_SHADOW_GRID_PER_CU", "1")
        monkeypatch.setenv("MCPT_DIRECT_GRID_PER_CU", "1")
        b, sb = hip.HipScene(scene).render(spp=4, seed=9)
        assert np.array_equal(a, b, equal_nan=True)
        assert all(getattr(sa, k) == getattr(sb, k) for k in COUNTERS)


def test_scene_beyond_the_limits_is_not_lds_resident(pkg, hip):
    """65 triangles: one more than the SMALL kernels hold."""
    s = pkg.scenes
    P = s.material_presets()
    b = s._Builder()
    rng = np.random.default_rng(1)
    tri = np.zeros(63, s.TRI_DTYPE)
    base = rng.uniform(-20, 20, (63, 3)).astype(np.float32)
    tri["v0"], tri["v1"], tri["v2"] = base, base + rng.normal(0, 4, (63, 3)).astype(np.float32), base + rng.normal(0, 4, (63, 3)).astype(np.float32)
    b.add_mesh(tri, b.material("rough_plastic", P["rough_plastic"]))
    lt = np.zeros(2, s.TRI_DTYPE)
    lt["v0"], lt["v1"], lt["v2"] = [(-10, 40, -10)] * 2, [(10, 40, -10), (10, 40, 10)], [(10, 40, 10), (-10, 40, 10)]
    b.add_mesh(lt, b.material("light", s._mat(s.ROUGH_CONDUCTOR, emission=(20, 20, 20))))
    cam = s.make_camera(48, 32, 60, (0, 0, -70), (0, 0, 0))
    sd = b.finish(camera=cam, rr_rate=0.5, spp=2, name="65 triangles")
    hs = hip.HipScene(sd)
    assert hs.info()["lds_resident"] == 0 and hs.info()["n_prims"] >= 65
    fb, _ = hs.render(spp=2, seed=1)
    assert np.isfinite(fb).all()
