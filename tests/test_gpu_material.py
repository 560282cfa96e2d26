"""Material.hpp function by function on the device (mcpt_debug_material) against the CPU restatement (orc_material_*): the same bits
for every material of the shipped scenes, on random configurations and on the ones the branches turn on (grazing directions,
total internal reflection, mirror / Snell configurations of the Dirac materials, n.h <= EPSILON, checkerboard cells).  The paths
already agree bit for bit (tests/test_gpu_parity.py); this pins each function on inputs the test scenes' paths rarely produce."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _unit(v):
    v = np.asarray(v, np.float32)
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


def _rows(rng, n):
    a, b, c = _unit(rng.normal(size=(n, 3))), _unit(rng.normal(size=(n, 3))), _unit(rng.normal(size=(n, 3)))
    k = n // 8
    c[:k] = [0, 0, 1]                                    # axis-aligned normal
    b[k:2 * k] = a[k:2 * k] * [-1, -1, 1]                 # wo = mirror image of wi about z ...
    c[k:2 * k] = [0, 0, 1]                                # ... with n = z: the Dirac reflect configuration (h = n)
    b[2 * k:3 * k] = c[2 * k:3 * k]                       # wo = n
    a[3 * k:4 * k] = c[3 * k:4 * k]                       # wi = n
    a[4 * k:5 * k] = _unit(np.cross(c[4 * k:5 * k], b[4 * k:5 * k]) + 1e-4 * c[4 * k:5 * k])  # wi almost perpendicular to n
    a[5 * k:6 * k] = -b[5 * k:6 * k]                      # wi = -wo: h = 0
    uv = rng.random((n, 2)).astype(np.float32)
    u = rng.random((n, 2)).astype(np.float32)
    u[:k, 1] = 0.0
    u[k:2 * k, 1] = np.float32(1.0) - np.float32(2.0 ** -24)
    return np.concatenate([a, b, c, uv, u], axis=1).astype(np.float32)


def _same(x, y):
    x, y = np.asarray(x, np.float32), np.asarray(y, np.float32)
    return (x.view(np.uint32) == y.view(np.uint32)) | (np.isnan(x) & np.isnan(y))


@pytest.mark.parametrize("scene", ["cornell_demo", "chess"])
def test_material_functions_bit_identical_to_the_oracle(pkg, oracle, hip, scene):
    sd = pkg.scenes.cornell_demo(16, 16, 1) if scene == "cornell_demo" else pkg.scenes.chess_scene(width=16, height=9, spp=1)
    hs = hip.HipScene(sd)
    L = oracle.lib()
    rng = np.random.default_rng(17)
    n_mat = len(sd.materials)
    per = 1536
    rows = np.concatenate([_rows(rng, per) for _ in range(n_mat)])
    sel = np.zeros((len(rows), 3), np.int32)
    sel[:, 0] = np.repeat(np.arange(n_mat), per)
    sel[:, 1] = rng.integers(0, 3, len(rows))
    sel[:, 2] = rng.integers(0, 2, len(rows))
    gpu = {k: hs.debug_material(k, rows, sel) for k in hs.MATERIAL_KINDS}
    mats = [np.ascontiguousarray(sd.materials[k]) for k in range(n_mat)]
    out3 = np.zeros(3, np.float32)
    bad = {k: 0 for k in gpu}
    for i in range(len(rows)):
        m = _p(mats[sel[i, 0]])
        a, b, c, uv = rows[i, 0:3].copy(), rows[i, 3:6].copy(), rows[i, 6:9].copy(), rows[i, 9:11].copy()
        ch, refl = int(sel[i, 1]), int(sel[i, 2])
        ev = L.orc_material_eval(m, _p(a), _p(b), _p(c), ch, _p(uv), refl)
        pd = L.orc_material_pdf(m, _p(a), _p(b), _p(c), ch, refl)
        bad["eval"] += not _same(ev, gpu["eval"][i, 0])
        bad["fresnel"] += not _same(L.orc_material_fresnel(m, _p(a), _p(b), ch), gpu["fresnel"][i, 0])
        L.orc_material_sample(m, _p(a), C.c_float(rows[i, 11]), C.c_float(rows[i, 12]), _p(out3))
        bad["sample"] += not _same(out3, gpu["sample"][i, :3]).all()
        L.orc_material_refract(m, _p(a), _p(b), ch, _p(out3))
        bad["refract"] += not _same(out3, gpu["refract"][i, :3]).all()
        # Material::pdf and the shading kernel's fused eval + pdf: rough materials only -- castRay never calls pdf() for a Dirac
        # material (Scene.cpp:137,164), and the device has no Dirac branch of it
        if int(sd.materials[sel[i, 0]]["type"]) in (pkg.scenes.ROUGH_CONDUCTOR, pkg.scenes.ROUGH_DIELECTRIC):
            bad["pdf"] += not _same(pd, gpu["pdf"][i, 0])
            bad["eval_pdf"] += not (_same(ev, gpu["eval_pdf"][i, 0]) and _same(pd, gpu["eval_pdf"][i, 1]))
    assert not any(bad.values()), bad
    # reflect(I, N) = 2 (N.I) N - I (Material.hpp:195-197), with Eigen's 3-term dot order, in float32
    I, N = rows[:, 0:3], rows[:, 3:6]
    d = (I[:, 0] * N[:, 0] + (I[:, 1] * N[:, 1] + I[:, 2] * N[:, 2])).astype(np.float32)
    want = (N * (np.float32(2) * d)[:, None]).astype(np.float32) - I
    assert _same(want, gpu["reflect"][:, :3]).all()
    assert np.isfinite(gpu["eval"][:, 0]).mean() > 0.9 and (gpu["eval"][:, 0] != 0).mean() > 0.05  # (the inputs reach the non-trivial branches)


def _quad(pkg, a, b, c, d):
    t = np.zeros(2, pkg.scenes.TRI_DTYPE)
    t["v0"], t["v1"], t["v2"] = [a, a], [b, c], [c, d]
    return t


def _many_lights(pkg):
    """Three emitters in insertion order -- a 1 x 1 quad, a sphere, a 3 x 2 fan of 12 triangles -- so that the light choice
    (Scene.cpp:28-36), the area walk of a mesh's tree (BVH.cpp:118-129) and Sphere::Sample all have something to choose from."""
    s = pkg.scenes
    P = s.material_presets()
    b = s._Builder()
    b.add_mesh(_quad(pkg, (-0.5, 2, -0.5), (0.5, 2, -0.5), (0.5, 2, 0.5), (-0.5, 2, 0.5)), b.material("l1", s._mat(s.ROUGH_CONDUCTOR, emission=(40, 35, 30))))
    b.add_sphere((2.0, 1.5, 0.0), 0.3, b.material("l2", s._mat(s.ROUGH_CONDUCTOR, emission=(5, 6, 7))))
    fan = np.zeros(12, s.TRI_DTYPE)
    for k in range(12):
        a0, a1 = 2 * np.pi * k / 12, 2 * np.pi * (k + 1) / 12
        fan["v0"][k], fan["v1"][k], fan["v2"][k] = (-3, 1, 0), (-3 + 1.5 * np.cos(a0), 1 + np.sin(a0), 0.1 * k), (-3 + 1.5 * np.cos(a1), 1 + np.sin(a1), 0.1 * k)
    b.add_mesh(fan, b.material("l3", s._mat(s.ROUGH_CONDUCTOR, emission=(1, 2, 3))))
    b.add_mesh(_quad(pkg, (-6, 0, -6), (-6, 0, 6), (6, 0, 6), (6, 0, -6)), b.material("floor", P["rough_white_conductor"]))
    cam = s.make_camera(32, 32, 55, (0.3, 1.3, -4.5), (0.2, 1.1, 0.0))
    return b.finish(background=np.float32([0.05, 0.05, 0.08]), camera=cam, rr_rate=0.8, spp=1, name="many_lights")


@pytest.mark.parametrize("scene", ["cornell_demo", "chess", "many_lights"])
def test_sample_light_bit_identical_to_the_oracle(pkg, oracle, hip, scene):
    sd = {"cornell_demo": lambda: pkg.scenes.cornell_demo(16, 16, 1), "chess": lambda: pkg.scenes.chess_scene(width=16, height=9, spp=1),
          "many_lights": lambda: _many_lights(pkg)}[scene]()
    rng = np.random.default_rng(5)
    u = (rng.integers(0, 1 << 24, size=(60000, 4)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)  # the path's uniforms
    u[:64] = rng.choice(np.float32([0.0, 1.0 - 2.0 ** -24, 0.5, 0.25]), size=(64, 4))
    ref = oracle.OracleScene(sd).sample_light(u)
    gpu = hip.HipScene(sd).sample_light(u)
    if scene == "many_lights":
        # Sphere::Sample leaves Intersection::emit as it was (Sphere.hpp:64-74); both sides start from zero
        assert len(np.unique(ref[:, 9])) == 3  # all three emitters were chosen (pdf = 1 / area of the chosen one)
    assert _same(ref, gpu).all(), int((~_same(ref, gpu)).sum())


def test_sample_env_bit_identical_to_the_oracle(pkg, oracle, hip):
    sd = pkg.scenes.cornell_rc(16, 16, 1)
    rng = np.random.default_rng(6)
    sd.env_pixels = rng.random((37, 64, 3)).astype(np.float32)
    d = _unit(rng.normal(size=(60000, 3)))
    d[:6] = [[0, 1, 0], [0, -1, 0], [1, 0, 0], [-1, 0, 0], [0, 0, 1], [0, 0, -1]]  # poles and the seam of the map
    d[6:3000, 1] = 0
    d[6:3000] = _unit(d[6:3000])
    d[3000:6000] *= rng.uniform(0.1, 9.0, (3000, 1)).astype(np.float32)            # (sampleEnv normalises its argument)
    ref = oracle.OracleScene(sd).sample_env(d)
    gpu = hip.HipScene(sd).sample_env(d)
    assert _same(ref, gpu).all(), int((~_same(ref, gpu)).sum())
    assert len(np.unique(ref[:, 0])) > 50000
    sd.env_pixels = None  # the constant background (Scene.hpp:61-63)
    assert np.array_equal(hip.HipScene(sd).sample_env(d[:10]), oracle.OracleScene(sd).sample_env(d[:10]))
