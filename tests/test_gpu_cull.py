"""Sky-pixel culling (csrc/mcpt_cull.hip): pixels that can only see the background are finished without tracing.  The classification is
conservative, so frames and reference-equivalent work counters must be identical with the culling on and off -- with and without
depth of field, with thin geometry crossing pixels, under tile partitions and progressive accumulation -- and identical to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _render(hip, sd, monkeypatch, cull, **kw):
    if cull:
        monkeypatch.delenv("MCPT_SKY_CULL", raising=False)
    else:
        monkeypatch.setenv("MCPT_SKY_CULL", "0")
    return hip.HipScene(sd).render(**kw)


def _thin_scene(pkg, dof):
    """Needles and slivers far thinner than a pixel, a small sphere, a floor: silhouettes everywhere."""
    s = pkg.scenes
    rng = np.random.default_rng(12)
    P = s.material_presets()
    b = s._Builder()
    n = 300
    tri = np.zeros(n, s.TRI_DTYPE)
    base = rng.uniform([-40, 0, -40], [40, 60, 40], (n, 3)).astype(np.float32)
    d1 = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d1 /= np.linalg.norm(d1, axis=1, keepdims=True)
    tri["v0"] = base
    tri["v1"] = base + d1 * rng.uniform(2, 30, (n, 1)).astype(np.float32)
    tri["v2"] = base + rng.normal(0, 0.02, (n, 3)).astype(np.float32)  # slivers 0.02 units wide
    b.add_mesh(tri, b.material("rough_white_conductor", P["rough_white_conductor"]))
    fl = np.zeros(2, s.TRI_DTYPE)
    fl["v0"], fl["v1"], fl["v2"] = [(-60, 0, -60)] * 2, [(-60, 0, 60), (60, 0, 60)], [(60, 0, 60), (60, 0, -60)]
    b.add_mesh(fl, b.material("gold_conductor", P["gold_conductor"]))
    light = s._mat(s.ROUGH_CONDUCTOR, emission=(30, 30, 30))
    lt = np.zeros(2, s.TRI_DTYPE)
    lt["v0"], lt["v1"], lt["v2"] = [(-10, 90, -10)] * 2, [(10, 90, -10), (10, 90, 10)], [(10, 90, 10), (-10, 90, 10)]
    b.add_mesh(lt, b.material("light", light))
    b.add_sphere((25, 40, 0), 1.5, b.material("smooth_glass", P["smooth_glass"]))
    cam = s.make_camera(160, 100, 65, (0, 30, -150), (0, 30, 0), (0, 1, 0), dof, 150.0, 4.0)
    return b.finish(camera=cam, rr_rate=0.5, spp=4, background=np.float32([0.3, 0.5, 0.8]), name="thin")


@pytest.mark.parametrize("name", ["chess", "chess_nodof", "thin_dof", "thin", "cornell_demo"])
def test_frames_identical_with_and_without_culling(pkg, oracle, hip, monkeypatch, name):
    if name.startswith("chess"):
        sd = pkg.scenes.chess_scene(width=320, height=180, spp=6)
        if name == "chess_nodof":
            sd.camera["use_dof"] = 0
    elif name.startswith("thin"):
        sd = _thin_scene(pkg, name == "thin_dof")
    else:
        sd = pkg.scenes.cornell_demo(64, 64, 4)
    kw = dict(spp=6, seed=3, spp_per_pass=4)
    a, sa = _render(hip, sd, monkeypatch, True, **kw)
    b, sb = _render(hip, sd, monkeypatch, False, **kw)
    assert np.array_equal(a, b, equal_nan=True), int((a != b).sum())
    assert (sa.samples, sa.vertices, sa.shaded, sa.ref_scene_rays, sa.shadow_rays) == (sb.samples, sb.vertices, sb.shaded, sb.ref_scene_rays, sb.shadow_rays)
    if name != "cornell_demo":
        assert sa.closest_rays < sb.closest_rays  # fewer rays actually traced
        culled = (sb.closest_rays - sa.closest_rays) / (sb.samples)
        print("\\n[cull] %s: %.1f %% of the samples finished without a ray" % (name, 100 * culled))
    else:
        assert sa.closest_rays == sb.closest_rays  # a closed box: nothing to cull
    ref, st = oracle.OracleScene(sd).render(spp=6, seed=3)
    same = (a == ref) | (np.isnan(a) & np.isnan(ref))
    assert (~same).sum() <= 3 and abs(int(sa.ref_scene_rays) - int(st.scene_rays)) <= 3 and sa.vertices == st.vertices


def test_culling_with_partitions_progressive_calls_and_an_all_sky_frame(pkg, hip, monkeypatch):
    sd = pkg.scenes.chess_scene(width=200, height=120, spp=8)
    full, _ = _render(hip, sd, monkeypatch, False, spp=8, seed=5)
    monkeypatch.delenv("MCPT_SKY_CULL", raising=False)
    hs = hip.HipScene(sd)
    parts = [hs.render(spp=8, seed=5, tile_size=16, rank=r, nranks=3)[0] for r in range(3)]
    assert np.array_equal(full, parts[0] + parts[1] + parts[2])
    fb, _ = hs.render(spp=5, spp_total=8, sample_offset=0, seed=5)
    fb, _ = hs.render(fb=fb, spp=3, spp_total=8, sample_offset=5, accumulate=1, seed=5)
    assert np.array_equal(full, fb)
    # a camera that looks away from everything: every pixel is culled, the frame is the background, no wavefront iteration runs
    sky = pkg.scenes.make_camera(64, 48, 40, (278, 5000, -2550), (278, 9000, -2550), (0, 0, 1), True, 3000.0, 10.0)
    fb, st = hs.render(camera=sky, spp=7, seed=1)
    acc = np.zeros(3, np.float32)
    for _ in range(7):
        acc += np.asarray(sd.background, np.float32) / np.float32(7)
    assert (fb == acc).all() and st.closest_rays == 0 and st.samples == 64 * 48 * 7 and st.vertices == 3 * st.samples
    # a camera matrix that is not orthonormal is outside what the bound covers: nothing is culled, the frame is still right
    odd = np.array(sd.camera, copy=True)
    odd["orientation"] = (np.asarray(sd.camera["orientation"]).reshape(3, 3) * np.float32(1.3)).reshape(-1)
    monkeypatch.delenv("MCPT_SKY_CULL", raising=False)
    a, sa = hip.HipScene(sd).render(camera=odd, spp=2, seed=1)
    b, sb = _render(hip, sd, monkeypatch, False, camera=odd, spp=2, seed=1)
    assert np.array_equal(a, b, equal_nan=True) and sa.closest_rays == sb.closest_rays
    # with an environment map the miss value depends on the direction: nothing is culled
    sd2 = pkg.scenes.chess_scene(width=96, height=54, spp=2)
    sd2.env_pixels = np.random.default_rng(0).random((8, 16, 3)).astype(np.float32)
    a, sa = _render(hip, sd2, monkeypatch, True, spp=2, seed=1)
    b, sb = _render(hip, sd2, monkeypatch, False, spp=2, seed=1)
    assert np.array_equal(a, b) and sa.closest_rays == sb.closest_rays
