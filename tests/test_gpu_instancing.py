"""Node instancing on the GPU: the 14 soldiers share one subtree of traversal nodes (csrc/mcpt_scene.cpp), entered with a shifted
origin; primitive tests still use every object's own world-space triangles.  Hits, paths and frames are those of the plain tree."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rays(oracle_scene, sd, n, seed):
    rng = np.random.default_rng(seed)
    W, H = int(sd.camera["width"]), int(sd.camera["height"])
    pix = rng.integers(0, W * H, size=n).astype(np.uint32)
    smp = rng.integers(0, 64, size=n).astype(np.uint32)
    o, d = oracle_scene.camera_rays(pix, smp, seed=7)
    t, prim = oracle_scene.intersect(o, d)
    hit = prim >= 0
    with np.errstate(over="ignore", invalid="ignore"):
        p = (o + d * t[:, None].astype(np.float32)).astype(np.float32)
    d2 = rng.normal(size=(n, 3)).astype(np.float32)
    d2 /= np.linalg.norm(d2, axis=1, keepdims=True)
    o2 = np.where(hit[:, None], p, o).astype(np.float32)
    return np.concatenate([o, o2]), np.concatenate([d, d2.astype(np.float32)])


@pytest.mark.parametrize("name,quant", [("chess", -1), ("chess", 0), ("chess_high", -1)])
def test_instanced_tree_gives_the_same_hits_paths_and_frames(pkg, oracle, hip, name, quant, capsys):
    sd = pkg.scenes.chess_scene(width=160, height=90, spp=4) if name == "chess" else pkg.scenes.chess_high(160, 90, 4)
    os_ = oracle.OracleScene(sd)
    hs = hip.HipScene(sd, builder="sah", quantise=quant, instancing=True)
    plain = hip.HipScene(sd, builder="sah", quantise=quant, instancing=False)
    info, pinfo = hs.info(), plain.info()
    assert info["n_instances"] == 14 and pinfo["n_instances"] == 0 and info["n_nodes"] < pinfo["n_nodes"] // 3
    with capsys.disabled():
        print("\n[instancing] %s: %d -> %d nodes, scene %.1f -> %.1f MB, stack %d -> %d entries"
              % (name, pinfo["n_nodes"], info["n_nodes"], pinfo["scene_bytes"] / 1e6, info["scene_bytes"] / 1e6, pinfo["bvh_height"], info["bvh_height"]))
    o, d = _rays(os_, sd, 20000, 11)
    t_ref, p_ref = os_.intersect(o, d)
    t_gpu, p_gpu = hs.intersect(o, d)
    assert np.array_equal(p_ref, p_gpu), "primitive ids differ on %d rays" % int((p_ref != p_gpu).sum())
    assert np.array_equal(t_ref.view(np.uint64), t_gpu.view(np.uint64))
    # per-path values against the oracle
    rng = np.random.default_rng(2)
    n = 20000
    pix = rng.integers(0, 160 * 90, size=n).astype(np.uint32)
    smp = rng.integers(0, 500, size=n).astype(np.uint32)
    ch = rng.integers(0, 3, size=n).astype(np.int32)
    co, cd = os_.camera_rays(pix, smp, seed=9)
    ref = os_.cast_rays(co, cd, pix, smp, ch, seed=9)
    gpu = hs.cast_rays(co, cd, pix, smp, ch, seed=9)
    bad = ~((ref.view(np.uint32) == gpu.view(np.uint32)) | (np.isnan(ref) & np.isnan(gpu)))
    assert bad.sum() <= 2, int(bad.sum())
    # frames: instanced == plain
    a, sa = hs.render(spp=4, seed=3)
    b, sb = plain.render(spp=4, seed=3)
    differing = int((~((a == b) | (np.isnan(a) & np.isnan(b)))).sum())
    assert differing <= 3, differing
    assert abs(int(sa.vertices) - int(sb.vertices)) <= 3


def test_instancing_is_opt_in(pkg, hip):
    assert hip.HipScene(pkg.scenes.chess_scene(width=32, height=32, spp=1)).info()["n_instances"] == 0
    assert hip.HipScene(pkg.scenes.chess_high(32, 32, 1)).info()["n_instances"] == 0  # (measured slower than the plain tree: a memory feature)
    assert hip.HipScene(pkg.scenes.chess_high(32, 32, 1), instancing=True).info()["n_instances"] == 14
    assert hip.HipScene(pkg.scenes.cornell_demo(32, 32, 1), instancing=True).info()["n_instances"] == 0  # nothing repeats
