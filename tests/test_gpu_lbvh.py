"""The GPU BVH builders (csrc/mcpt_lbvh.hip: MCPT_BUILD_GPU_LBVH, the linear BVH, and MCPT_BUILD_GPU_PLOC, parallel locally-ordered
clustering, the one of near-SAH quality): the tree they leave in HBM satisfies the invariants of the host
builders' trees (tests/test_bvh_host.py: every primitive in exactly one leaf, child boxes contain what is below them, the
declared stack bound covers the height, quantised boxes contain the exact ones), closest hits are bit-identical to the oracle's
full traversal, and frames equal those rendered with the host-built SAH tree."""
import numpy as np
import pytest
from test_bvh_host import check_tree

pytestmark = pytest.mark.gpu


def _soup(pkg, n=3000, seed=5, duplicates=True):
    rng = np.random.default_rng(seed)
    base = pkg.scenes.cornell_rc(32, 32, 1)
    tri = np.zeros(n, dtype=base.triangles.dtype)
    c = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    if duplicates:  # many identical centroids: equal Morton codes, told apart by position only
        c[n // 2:] = c[n // 2]
    off = rng.normal(0, 0.5, (3, n, 3)).astype(np.float32)
    off[2] = -(off[0] + off[1])  # the three offsets sum to zero: the centroid of the box stays near c
    for k, name in enumerate(("v0", "v1", "v2")):
        tri[name] = c + off[k]
    obj = np.zeros(1, dtype=base.objects.dtype)
    obj["kind"], obj["material"], obj["first_tri"], obj["n_tri"] = 0, 0, 0, n
    return pkg.scenes.SceneData(triangles=tri, materials=base.materials[:1].copy(), objects=obj, background=base.background,
                                env_pixels=None, camera=base.camera, rr_rate=base.rr_rate)


def sah_cost(boxes, root_min, root_max):
    """Sum of the surface areas of all inner nodes' boxes (every child box that is not a leaf's, plus the root) over the root's: the
    expected number of node visits of a random ray through the root box -- what the SAH minimises."""
    def area(mn, mx):
        d = np.maximum(mx - mn, 0).astype(np.float64)
        return d[..., 0] * d[..., 1] + d[..., 1] * d[..., 2] + d[..., 2] * d[..., 0]
    b = boxes.reshape(-1, 2, 2, 3).astype(np.float64)  # node, child, (min, max), xyz
    node_mn, node_mx = b[:, :, 0].min(axis=1), b[:, :, 1].max(axis=1)
    return float(area(node_mn, node_mx).sum() / area(np.asarray(root_min, np.float64), np.asarray(root_max, np.float64)))


@pytest.mark.parametrize("builder", ["lbvh", "ploc"])
@pytest.mark.parametrize("name", ["cornell_demo", "chess", "soup", "chess_high"])
def test_gpu_built_tree_invariants_and_hits(pkg, oracle, hip, name, builder, capsys):
    sd = {"cornell_demo": lambda: pkg.scenes.cornell_demo(64, 64, 2), "chess": lambda: pkg.scenes.chess_scene(width=160, height=90, spp=2),
          "soup": lambda: _soup(pkg), "chess_high": lambda: pkg.scenes.chess_high(160, 90, 2)}[name]()
    hs = hip.HipScene(sd, builder=builder)
    info, boxes, children, qboxes = hs.dump_bvh()
    n_prims = len(sd.triangles) + int((sd.objects["kind"] == 1).sum())
    assert info["n_nodes"] == n_prims - 1 == len(boxes)
    h = check_tree(sd, info, boxes, children, qboxes)
    meta = hs.info()
    assert meta["builder"] == (2 if builder == "lbvh" else 3) and meta["bvh_height"] == info["stack_entries"] and h + 1 == info["stack_entries"]
    hsah = hip.HipScene(sd, builder="sah")
    sah = hsah.info()
    si, sboxes, _, _ = hsah.dump_bvh()
    cost, cost_sah = sah_cost(boxes, info["root_min"], info["root_max"]), sah_cost(sboxes, si["root_min"], si["root_max"])
    with capsys.disabled():
        print("\n[%s] %-12s %7d prims: GPU build %.2f ms (height %d, quantised %d, SAH cost %.2f)  |  host SAH build %.1f ms (height %d, SAH cost %.2f)"
              % (builder, name, n_prims, meta["build_ms"], meta["bvh_height"], meta["quantised"], cost, sah["build_ms"], sah["bvh_height"], cost_sah))
    if builder == "ploc" and name in ("chess", "chess_high"):
        assert cost < 1.25 * cost_sah  # (the linear BVH: 1.5-2x)
    # closest hits against the oracle's full traversal of the reference's tree: bit-exact
    rng = np.random.default_rng(3)
    n = 20000
    lo, hi = np.array(info["root_min"], np.float32), np.array(info["root_max"], np.float32)
    o = rng.uniform(lo - 0.2 * (hi - lo), hi + 0.2 * (hi - lo), size=(n, 3)).astype(np.float32)
    tgt = rng.uniform(lo, hi, size=(n, 3)).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    t_ref, p_ref = oracle.OracleScene(sd).intersect(o, d)
    t_gpu, p_gpu = hs.intersect(o, d)
    assert np.array_equal(p_ref, p_gpu), "primitive ids differ on %d rays" % int((p_ref != p_gpu).sum())
    assert np.array_equal(t_ref.view(np.uint64), t_gpu.view(np.uint64))
    assert (p_ref >= 0).mean() > (0.02 if name == "soup" else 0.2)


@pytest.mark.parametrize("name", ["cornell_demo", "chess"])
def test_frames_do_not_depend_on_the_builder(pkg, hip, name):
    sd = pkg.scenes.cornell_demo(96, 96, 8) if name == "cornell_demo" else pkg.scenes.chess_scene(width=240, height=135, spp=8)
    ref, st0 = hip.HipScene(sd, builder="sah").render(spp=8, seed=6)
    for builder, quant in (("lbvh", -1), ("lbvh", 0), ("ploc", -1), ("ploc", 0), ("reference", -1)):
        fb, st = hip.HipScene(sd, builder=builder, quantise=quant).render(spp=8, seed=6)
        differing = int((~((fb == ref) | (np.isnan(fb) & np.isnan(ref)))).sum())
        assert differing <= 3, (builder, quant, differing)  # (a box-grazing ray may take another branch)
        assert abs(int(st.vertices) - int(st0.vertices)) <= 3


def test_single_primitive_and_options_errors(pkg, hip):
    sd = _soup(pkg, n=1, duplicates=False)
    hs = hip.HipScene(sd, builder="lbvh")  # no inner node: nothing to build on the device
    assert hs.info()["n_nodes"] == 0
    t, p = hs.intersect(np.float32([[0, 0, -200]]), np.float32([[0, 0, 1]]))
    assert p[0] in (-1, 0)
    import ctypes as C
    opt = hip.BuildOptions(builder=9, quantise=-1)
    keep = []
    d = hip._make_desc(sd, keep)
    h = C.c_void_p()
    assert hip.lib().mcpt_scene_create_ex(C.byref(d), -1, C.byref(opt), C.byref(h)) == 1


def _chain(pkg, n):
    """n triangles that all cover the square [-1, 0]^2 of their plane z = const, the k-th one reaching out to 2^(k // 3) along the
    axis k % 3 (long in x, long in y, or far away in z): every centroid has its own leading bit in the interleaved Morton code, so
    the linear BVH is one chain of about n levels, and a ray along +z through that square meets every box of it."""
    base = pkg.scenes.cornell_rc(32, 32, 1)
    tri = np.zeros(n, dtype=base.triangles.dtype)
    for k in range(n):
        L = np.float32(3.0 * 2.0 ** (k // 3 + 1))
        a = k % 3
        z = np.float32(0.01 * k) if a < 2 else L
        if a == 0:
            v = [[-1, -1, z], [L, -1, z], [-1, 1, z]]
        elif a == 1:
            v = [[-1, -1, z], [1, -1, z], [-1, L, z]]
        else:
            v = [[-1, -1, z], [3, -1, z], [-1, 3, z]]
        tri["v0"][k], tri["v1"][k], tri["v2"][k] = np.float32(v)
    obj = np.zeros(1, dtype=base.objects.dtype)
    obj["kind"], obj["material"], obj["first_tri"], obj["n_tri"] = 0, 0, 0, n
    return pkg.scenes.SceneData(triangles=tri, materials=base.materials[:1].copy(), objects=obj, background=base.background,
                                env_pixels=None, camera=base.camera, rr_rate=base.rr_rate)


def test_deep_trees_are_exact_and_too_deep_ones_are_refused(pkg, oracle, hip):
    """Trees deeper than 24 levels run the retry flavour of the traversal stack (16 LDS entries, rays that need more are traced again
    with a scratch stack); a tree deeper than the scratch stack (48) is refused at creation, loudly, with the way out in the message."""
    rng = np.random.default_rng(4)
    heights = []
    for n in (24, 36, 45, 63):
        sd = _chain(pkg, n)
        hp = hip.HipScene(sd, builder="ploc")  # (merges by surface area: no chain, whatever the codes look like)
        assert hp.info()["bvh_height"] <= 48
        try:
            hs = hip.HipScene(sd, builder="lbvh")
        except RuntimeError as e:
            assert "deeper than the traversal stack" in str(e) and "MCPT_BUILD_SAH" in str(e)
            heights.append(None)
            continue
        h = hs.info()["bvh_height"]
        heights.append(h)
        assert h <= 48
        m = 20000
        o = np.concatenate([rng.uniform(-0.9, -0.1, (m, 2)), np.full((m, 1), -5.0)], axis=1).astype(np.float32)
        d = np.concatenate([rng.normal(0, 0.02, (m, 2)), np.ones((m, 1))], axis=1).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        t_ref, p_ref = oracle.OracleScene(sd).intersect(o, d)
        t_gpu, p_gpu = hs.intersect(o, d)
        assert np.array_equal(p_ref, p_gpu) and np.array_equal(t_ref[p_ref >= 0], t_gpu[p_ref >= 0])
        t_sah, p_sah = hip.HipScene(sd, builder="sah").intersect(o, d)
        assert np.array_equal(p_sah, p_gpu) and np.array_equal(hp.intersect(o, d)[1], p_gpu)
    print("\n[lbvh] chain scenes: heights", heights)
    assert any(h is not None and h > 24 for h in heights), heights  # the retry flavour was in use
