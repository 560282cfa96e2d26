"""csrc/mcpt_fmath.h: the plain-IEEE sin/cos/atan2/acos shared by the GPU kernels and the CPU oracle.

CPU part: the host compilation of the header (through the oracle library's orc_fmath) against the correctly rounded value
(numpy in float64, rounded once to float32) -- at most 1 ulp anywhere, and bit-identical on all but a handful of arguments,
i.e. it is as faithful to the reference's std::sin/std::cos/std::atan2/std::acos as any libm.
GPU part: the device compilation of the same header gives the same bits (mcpt_debug_fmath), which is what makes
"same seed => same paths" hold between kernels and oracle.
"""
import numpy as np
import pytest


def _ulps(a, b):
    """Distance in float32 ulps (monotone integer mapping of the bit patterns)."""
    def key(v):
        i = v.view(np.int32).astype(np.int64)
        return np.where(i < 0, -(i & 0x7FFFFFFF), i)
    return np.abs(key(np.ascontiguousarray(a, np.float32)) - key(np.ascontiguousarray(b, np.float32)))


def _inputs():
    rng = np.random.default_rng(1)
    two_pi = np.float32(2.0) * np.float32(3.141592653589793)
    u = (rng.integers(0, 1 << 24, size=400000).astype(np.float32) * np.float32(1.0 / 16777216.0))  # the path's uniforms
    ang = np.concatenate([two_pi * u, np.float32(3.141592653589793) * u[:50000],
                          np.array([0.0, -0.0, 1e-30, 1e-8, 0.78539816, 1.5707964, 3.1415927, 4.712389, 6.2831855, 6.283186, -1.0, -7.5, 100.0, 1e4], np.float32)])
    d = rng.normal(size=(200000, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True).astype(np.float32)
    return ang.astype(np.float32), d.astype(np.float32)


def test_host_fmath_within_one_ulp_of_correct_rounding(oracle):
    ang, d = _inputs()
    for kind, f in (("sin", np.sin), ("cos", np.cos)):
        got = oracle.fmath(kind, ang)
        want = f(ang.astype(np.float64)).astype(np.float32)
        ul = _ulps(got, want)
        assert ul.max() <= 1, (kind, int(ul.max()))
        assert (ul > 0).mean() < 1e-4, (kind, float((ul > 0).mean()))
    got = oracle.fmath("atan2", d[:, 2], d[:, 0])
    want = np.arctan2(d[:, 2].astype(np.float64), d[:, 0].astype(np.float64)).astype(np.float32)
    ul = _ulps(got, want)
    assert ul.max() <= 1 and (ul > 0).mean() < 1e-4
    y = np.concatenate([d[:, 1], np.array([1.0, -1.0, 0.0, -0.0, 0.99999994, -0.99999994, 0.5, -0.5], np.float32)])
    got = oracle.fmath("acos", y)
    want = np.arccos(y.astype(np.float64)).astype(np.float32)
    ul = _ulps(got, want)
    assert ul.max() <= 1 and (ul > 0).mean() < 1e-4


def test_host_fmath_special_values(oracle):
    pi = np.float32(np.pi)
    z, nz = np.float32(0.0), np.float32(-0.0)
    a = oracle.fmath("atan2", [z, nz, z, nz, 1, -1, 1, -1], [1, 1, -1, -1, z, z, nz, nz])
    want = np.arctan2(np.array([z, nz, z, nz, 1, -1, 1, -1], np.float64), np.array([1, 1, -1, -1, z, z, nz, nz], np.float64)).astype(np.float32)
    assert np.array_equal(a.view(np.uint32), want.view(np.uint32))  # signed zeros, +-pi, +-pi/2
    assert np.array_equal(oracle.fmath("atan2", [z, z], [z, nz]), np.array([0, pi], np.float32))
    ac = oracle.fmath("acos", [1.0000001, -1.0000001, np.nan])
    assert np.isnan(ac).all()  # outside [-1, 1]: NaN, like std::acos
    assert np.isnan(oracle.fmath("sin", [np.nan])).all() and np.isnan(oracle.fmath("atan2", [np.nan], [1.0])).all()
    assert oracle.fmath("sin", [0.0])[0] == 0.0 and oracle.fmath("cos", [0.0])[0] == 1.0


def _tone_inputs():
    rng = np.random.default_rng(7)
    # every float whose image 255 c^0.45 lies within a few ulps of an integer boundary is where a last-bit difference would show:
    # take the boundary pre-images (k/255)^(1/0.45) with their neighbours, plus random radiances over the whole range
    k = np.arange(0, 256, dtype=np.float64)
    pre = ((k / 255.0) ** (1 / 0.45)).astype(np.float32)
    near = np.concatenate([np.nextafter(pre, np.float32(np.inf)), np.nextafter(pre, np.float32(-np.inf)), pre])
    for _ in range(3):
        near = np.concatenate([near, np.nextafter(near, np.float32(np.inf)), np.nextafter(near, np.float32(-np.inf))])
    rand = np.concatenate([rng.random(400000).astype(np.float32), (rng.random(100000) * 20).astype(np.float32),
                           (10.0 ** rng.uniform(-12, 1, 100000)).astype(np.float32)])
    special = np.array([0.0, -0.0, -1.0, np.nan, np.inf, 1.0, 1e-45, 1e-38, 3e38, 0.5], np.float32)
    return np.concatenate([near.astype(np.float32), rand, special])


def test_host_pow_and_tonemap_match_glibc(oracle):
    """mcpt_powf against the correctly rounded x^0.45 (<= 1 ulp), and the tone-map byte against the reference's expression with
    glibc's powf (orc_tonemap = Renderer.cpp:95-103 verbatim): identical on every tested input, including the pre-images of all
    256 byte boundaries and their neighbours."""
    x = _tone_inputs()
    fin = np.isfinite(x) & (x > 0)
    got = oracle.fmath("pow", x[fin], np.full(fin.sum(), 0.45, np.float32))
    want = (x[fin].astype(np.float64) ** np.float64(np.float32(0.45))).astype(np.float32)
    ul = _ulps(got, want)
    assert ul.max() <= 1 and (ul > 0).mean() < 1e-4, (int(ul.max()), float((ul > 0).mean()))
    ours = oracle.fmath("tonemap", x).astype(np.uint8)
    fb = np.stack([x, x, x], axis=-1)
    ref = oracle.tonemap(fb)[..., 0]
    bad = ours != ref
    assert not bad.any(), (int(bad.sum()), x[bad][:8], ours[bad][:8], ref[bad][:8])
    assert (ours[np.isnan(x)] == 255).all() and (ours[x == np.inf] == 255).all() and (ours[x < 0] == 255).all()  # NaN clamps to the upper bound


@pytest.mark.gpu
def test_device_tonemap_matches_the_reference_expression(oracle, hip, pkg):
    x = _tone_inputs()
    n = (len(x) // 3) * 3
    fb = x[:n].reshape(1, -1, 3)
    hs = hip.HipScene(pkg.scenes.cornell_rc(8, 8, 1))
    gpu = hs.tonemap(fb)
    ref = oracle.tonemap(fb)
    assert np.array_equal(gpu, ref), int((gpu != ref).sum())
    # device pow == host pow bit for bit
    fin = np.isfinite(x) & (x > 0)
    y = np.full(fin.sum(), 0.45, np.float32)
    a, b = oracle.fmath("pow", x[fin], y), hip.debug_fmath("pow", x[fin], y)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    # a rendered frame: on-GPU tone map == the Python/NumPy mirror == the oracle's
    sd = pkg.scenes.cornell_demo(64, 64, 4)
    frame, _ = hip.HipScene(sd).render(spp=4, seed=1)
    assert np.array_equal(hs.tonemap(frame)[..., :3], pkg.pngio.tonemap_u8(frame)) and np.array_equal(hs.tonemap(frame), oracle.tonemap(frame))


@pytest.mark.gpu
def test_device_fmath_is_bit_identical_to_host(oracle, hip):
    ang, d = _inputs()
    y = np.concatenate([d[:, 1], np.array([1.0, -1.0, 0.0, -0.0, 1.0000001, np.nan], np.float32)])
    for kind, x, x2 in (("sin", ang, None), ("cos", ang, None), ("atan2", d[:, 2], d[:, 0]), ("acos", y, None)):
        cpu = oracle.fmath(kind, x, x2)
        gpu = hip.debug_fmath(kind, x, x2)
        same = (cpu.view(np.uint32) == gpu.view(np.uint32)) | (np.isnan(cpu) & np.isnan(gpu))
        assert same.all(), (kind, int((~same).sum()), x[~same][:5], cpu[~same][:5], gpu[~same][:5])
