"""Pins the CPU oracle (oracle/mcpt_oracle.c).

The reference has no tests and no golden vectors, and it cannot be built here (Eigen3 is missing), so the
pins are, in decreasing strength:
  1. cornellbox_demo.png -- the one output image the reference ships for its DEMO scene (tests/golden/
     reference_cornellbox_demo.png, 384x384).  The oracle must reproduce it statistically.
  2. the reference's exact call counts per sample measured from its compiled sources during the survey
     (SURVEY.md Appendix D): rays, castRay invocations, BVH node visits and triangle tests per sample.
  3. closed-form values of the material functions at configurations where the reference's formulas
     (Material.hpp) can be evaluated by hand, including its documented quirks.
  4. Random123's published known-answer vectors for Philox4x32-10.
  5. committed golden renders of the oracle itself (regression guard; tests/golden/make_golden.py).
"""
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_philox_known_answers(oracle):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert [hex(x) for x in oracle.philox([0, 0, 0, 0], [0, 0])] == ["0x6627e8d5", "0xe169c58d", "0xbc57ac4c", "0x9b00dbd8"]
    assert [hex(x) for x in oracle.philox([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)] == ["0x408f276d", "0x41c83b0e", "0xa20bc7c6", "0x6d5451fd"]
    assert [hex(x) for x in oracle.philox([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])] == \
        ["0xd16cfe09", "0x94fdcceb", "0x5001e420", "0x24126ea1"]


def _lin(u8):
    return (u8.astype(np.float64) / 255.0) ** (1.0 / 0.45)


def _blocks(a, b):
    h, w, c = a.shape
    return a.reshape(h // b, b, w // b, b, c).mean(axis=(1, 3))


def test_oracle_reproduces_reference_demo_image(pkg, oracle):
    """Oracle render of the DEMO scene (main.cpp:99-129) vs the reference's own cornellbox_demo.png."""
    ref = pkg.pngio.read_png(os.path.join(GOLDEN, "reference_cornellbox_demo.png"))[:, :, :3]
    assert ref.shape == (384, 384, 3)
    sd = pkg.scenes.cornell_demo(192, 192, 40)  # each pixel integrates a 2x2 block of the 384x384 frame
    fb, st = oracle.OracleScene(sd).render(spp=40, seed=1)
    ours = pkg.pngio.tonemap_u8(fb)
    # compare 16x16 blocks of the reference with 8x8 blocks of ours, in linear radiance
    A = _blocks(_lin(ref), 16)
    B = _blocks(_lin(ours), 8)
    rel_mean = np.abs(A.mean(axis=(0, 1)) - B.mean(axis=(0, 1))) / A.mean(axis=(0, 1))
    assert (rel_mean < 0.03).all(), "per-channel mean radiance differs: %s" % rel_mean
    rel_l1 = np.abs(A - B).mean() / A.mean()
    assert rel_l1 < 0.05, "block-wise relative L1 %.4f" % rel_l1
    corr = np.corrcoef(A.ravel(), B.ravel())[0, 1]
    assert corr > 0.99, "block correlation %.4f" % corr


def test_oracle_call_counts_match_reference_cornell(pkg, oracle):
    """SURVEY.md Appendix D (gprof call counts of the compiled reference, DEMO scene 64x64 spp 8)."""
    sd = pkg.scenes.cornell_demo(64, 64, 8)
    _, st = oracle.OracleScene(sd).render(spp=8, seed=1)
    assert st.samples == 32768
    assert st.scene_rays / st.samples == pytest.approx(38.1, rel=0.02)
    assert st.vertices / st.samples == pytest.approx(6.96, rel=0.02)
    assert st.node_visits / st.scene_rays == pytest.approx(31.0, rel=0.01)
    assert st.tri_tests / st.scene_rays == pytest.approx(4.7, rel=0.02)


def test_oracle_call_counts_match_reference_chess(pkg, oracle):
    """SURVEY.md Appendix D (chess scene 96x54 spp 4, conf.json defaults, flat sky colour)."""
    sd = pkg.scenes.chess_scene(width=96, height=54, spp=4)
    _, st = oracle.OracleScene(sd).render(spp=4, seed=1)
    assert st.samples == 20736
    assert st.scene_rays / st.samples == pytest.approx(8.80, rel=0.03)
    assert st.vertices / st.samples == pytest.approx(3.28, rel=0.02)
    assert st.node_visits / st.scene_rays == pytest.approx(65.5, rel=0.02)
    assert st.tri_tests / st.scene_rays == pytest.approx(5.2, rel=0.03)


# ---- material known answers (Material.hpp), hand-evaluated
def _mat(pkg, name):
    return np.ascontiguousarray(pkg.scenes.material_presets()[name])


def _p(a):
    import ctypes
    return np.ascontiguousarray(a, dtype=np.float32).ctypes.data_as(ctypes.c_void_p)


def test_material_known_answers(pkg, oracle):
    import ctypes as C
    L = oracle.lib()
    n = np.array([0, 0, 1], np.float32)
    glass = _mat(pkg, "smooth_glass")
    gp = glass.ctypes.data_as(C.c_void_p)
    # Cauchy ior (Material.hpp:178-183) and normal-incidence Fresnel ((n-1)/(n+1))^2 (Material.hpp:198-226)
    for ch, wl in enumerate([0.700, 0.5461, 0.4358]):
        ior = np.float32(1.7) + np.float32(0.04) / (np.float32(wl) * np.float32(wl))
        kr = L.orc_material_fresnel(gp, _p([0, 0, -1]), _p(n), ch)
        assert kr == pytest.approx(((ior - 1) / (ior + 1)) ** 2, rel=1e-5)
    # conductors: fresnel() == 1 (Material.hpp:200-203)
    gold = _mat(pkg, "gold_conductor")
    assert L.orc_material_fresnel(gold.ctypes.data_as(C.c_void_p), _p([0, 0, -1]), _p(n), 0) == 1.0
    # total internal reflection from inside (Material.hpp:212-213): I.N > 0, sin_t >= 1
    I = np.array([np.sin(1.2), 0, np.cos(1.2)], np.float32)
    assert L.orc_material_fresnel(gp, _p(I), _p(n), 0) == 1.0
    # smooth conductor eval at the mirror configuration = Schlick(f0, cos) (Material.hpp:80-86,386-388)
    wo = np.array([np.sin(0.5), 0, np.cos(0.5)], np.float32)
    wi = np.array([-np.sin(0.5), 0, np.cos(0.5)], np.float32)
    f0 = 0.85
    c = float(wo[2])
    expect = f0 + (1 - f0) * (1 - c) ** 5
    got = L.orc_material_eval(gold.ctypes.data_as(C.c_void_p), _p(wi), _p(wo), _p(n), 1, _p([0, 0]), 1)
    assert got == pytest.approx(expect, rel=1e-5)
    # ... and 0 away from it (Material.hpp:382-384)
    wi2 = np.array([-np.sin(0.6), 0, np.cos(0.6)], np.float32)
    assert L.orc_material_eval(gold.ctypes.data_as(C.c_void_p), _p(wi2), _p(wo), _p(n), 1, _p([0, 0]), 1) == 0.0
    # rough conductor, wi = wo = n: D uses alpha (not alpha^2) next to tan^2 (Material.hpp:32):
    # D = a^2 / (pi * (1*(a+0))^2) = 1/pi ; G = 1 ; F = f0 ; denom = 4 + 1e-4
    white = _mat(pkg, "rough_white_conductor")
    got = L.orc_material_eval(white.ctypes.data_as(C.c_void_p), _p(n), _p(n), _p(n), 0, _p([0, 0]), 1)
    assert got == pytest.approx(0.725 * (1 / np.pi) / (4 + 1e-4), rel=1e-5)
    # rough pdf at the same configuration: D * (n.h) * 1/(4 |h.wo|) = (1/pi)/4 (Material.hpp:293-308)
    assert L.orc_material_pdf(white.ctypes.data_as(C.c_void_p), _p(n), _p(n), _p(n), 0, 1) == pytest.approx(1 / (4 * np.pi), rel=1e-5)
    # GGX sampling with u2 = 0 returns the normal itself (Material.hpp:111-123)
    out = np.zeros(3, np.float32)
    L.orc_material_sample(white.ctypes.data_as(C.c_void_p), _p(n), C.c_float(0.3), C.c_float(0.0), _p(out))
    assert np.allclose(out, n, atol=1e-6)
    # refraction at normal incidence keeps the direction (Material.hpp:227-242)
    L.orc_material_refract(gp, _p([0, 0, -1]), _p(n), 0, _p(out))
    assert np.allclose(out, [0, 0, -1], atol=1e-6)
    # checkerboard reflectance (Material.hpp:134-151): col 3..5, row <= 7, white iff (col+row) odd
    silver = _mat(pkg, "silver_mirror").copy()
    silver["textured"] = 1
    sp = silver.ctypes.data_as(C.c_void_p)
    cos = float(wo[2])

    def refl(u, v):
        val = L.orc_material_eval(sp, _p(wi), _p(wo), _p(n), 0, _p([u, v]), 1)
        return (val - (1 - cos) ** 5) / (1 - (1 - cos) ** 5)  # invert Schlick for f0

    assert refl(0.36, 0.01) == pytest.approx(0.9, abs=1e-4)   # col 3, row 0 -> odd -> white
    assert refl(0.46, 0.01) == pytest.approx(0.1, abs=1e-4)   # col 4, row 0 -> even
    assert refl(0.36, 0.70) == pytest.approx(0.1, abs=1e-4)   # row 8 > 7
    assert refl(0.20, 0.01) == pytest.approx(0.1, abs=1e-4)   # col 1


def test_tonemap_matches_reference_rule(pkg, oracle):
    fb = np.array([[[0.0, 1.0, 4.0], [0.25, np.nan, 1e-8]]], np.float32)
    a = oracle.tonemap(fb)[..., :3]
    b = pkg.pngio.tonemap_u8(fb)
    assert np.array_equal(a, b)
    assert a[0, 0].tolist() == [0, 255, 255] and a[0, 1, 1] == 255  # NaN clamps to the upper bound (global.hpp:16-18)
    assert a[0, 1, 0] == int(255 * 0.25 ** 0.45)


@pytest.mark.parametrize("name", ["cornell_demo_48x48_spp4", "chess_96x54_spp2"])
def test_oracle_golden_renders(pkg, oracle, name):
    """Regression guard: the oracle still produces its committed golden frames (bit-exact on this toolchain,
    1e-5 otherwise: libm differences)."""
    g = np.load(os.path.join(GOLDEN, "oracle_%s.npy" % name))
    if name.startswith("cornell"):
        sd, spp = pkg.scenes.cornell_demo(48, 48, 4), 4
    else:
        sd, spp = pkg.scenes.chess_scene(width=96, height=54, spp=2), 2
    fb, _ = oracle.OracleScene(sd).render(spp=spp, seed=1)
    close = np.isclose(fb, g, rtol=1e-5, atol=1e-6, equal_nan=True)
    assert close.mean() > 0.999, "only %.5f of the golden frame reproduced" % close.mean()


def test_oracle_tile_partition(pkg, oracle):
    sd = pkg.scenes.cornell_rc(64, 48, 2)
    s = oracle.OracleScene(sd)
    full, _ = s.render(spp=2, seed=3)
    parts = []
    for r in range(3):
        fb = np.zeros_like(full)
        s.render(fb=fb, spp=2, seed=3, tile_size=16, rank=r, nranks=3)
        parts.append(fb)
    assert np.array_equal(full, parts[0] + parts[1] + parts[2])
