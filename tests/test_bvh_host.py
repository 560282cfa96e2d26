"""Host-side checks of the BVH builder (the data producer of the hot path) through mcpt_bvh_dump: no GPU needed.

Invariants of the traversal tree that mcpt_scene_create uploads (reference: BVHAccel::recursiveBuild, BVH.cpp:27-93,
whose two-level topology MCPT_BVH=reference keeps; the default is one binned-SAH tree over all primitives):
every primitive sits in exactly one leaf, every child box contains what is below it, the declared traversal-stack
bound covers the tree, and the quantised 16-bit boxes contain the exact ones."""
import numpy as np
import pytest


def prim_bounds(sd):
    tri = sd.triangles
    v = np.stack([tri["v0"], tri["v1"], tri["v2"]], axis=1).astype(np.float32)  # [n, 3, 3]
    mn, mx = [v.min(axis=1)], [v.max(axis=1)]
    obj = sd.objects
    sph_mn = (obj["center"] - obj["radius"][:, None]).astype(np.float32)
    sph_mx = (obj["center"] + obj["radius"][:, None]).astype(np.float32)
    return np.concatenate(mn + [sph_mn]), np.concatenate(mx + [sph_mx])  # sphere prim id = n_tri + object index


def check_tree(sd, info, boxes, children, qboxes):
    pmn, pmx = prim_bounds(sd)
    n_tri = len(sd.triangles)
    n_prims = n_tri + int((sd.objects["kind"] == 1).sum())
    seen = np.zeros(len(pmn), np.int32)
    eps = 1e-5

    n_leaf = info.get("n_leaf_prims") or (len(pmn) + 1)
    shifts, root_first = info.get("inst_shift"), info.get("inst_root_first")

    def subtree(ref, shift=None, base=0):  # -> (mn, mx, height) of everything below a child reference
        if ref < 0 and ~ref >= n_leaf:
            # an instance: its prototype's subtree holds boxes in the prototype's position and local triangle indices; every box
            # below must contain THIS object's triangle moved back by the shift (checked in float64), and the instance's own
            # leaf box (world position) must contain the object.  One extra stack entry for the exit marker.
            k = ~ref - n_leaf
            assert shift is None, "nested instances"
            cmn, cmx, ch = subtree(int(root_first[k][0]), shift=shifts[k].astype(np.float64), base=int(root_first[k][1]))
            return cmn + shifts[k], cmx + shifts[k], ch + 1
        if ref < 0:
            p = base + ~ref
            seen[p] += 1
            if shift is not None:
                return pmn[p].astype(np.float64) - shift, pmx[p].astype(np.float64) - shift, 0
            return pmn[p], pmx[p], 0
        b = boxes[ref]
        out_mn, out_mx, h = None, None, 0
        for side, cref in enumerate(children[ref]):
            cmn, cmx, ch = subtree(int(cref), shift, base)
            box_mn, box_mx = b[6 * side:6 * side + 3], b[6 * side + 3:6 * side + 6]
            assert (box_mn <= cmn + eps).all() and (box_mx >= cmx - eps).all(), (ref, side)
            out_mn = box_mn if out_mn is None else np.minimum(out_mn, box_mn)
            out_mx = box_mx if out_mx is None else np.maximum(out_mx, box_mx)
            h = max(h, ch)
        return out_mn, out_mx, h + 1

    import sys
    sys.setrecursionlimit(10000)
    mn, mx, height = subtree(info["root"])
    assert (np.array(info["root_min"]) <= mn + eps).all() and (np.array(info["root_max"]) >= mx - eps).all()
    used = seen[:n_tri].tolist() + seen[n_tri:][sd.objects["kind"] == 1].tolist()
    assert all(c == 1 for c in used) and sum(used) == n_prims
    assert height <= info["stack_entries"]  # near-first traversal leaves at most one entry per level on the stack
    if qboxes is not None:
        o = np.array(info["q_origin"], np.float32)
        c = np.array(info["q_cell"], np.float32)
        q = qboxes.astype(np.float32).reshape(-1, 4, 3)            # lmin, lmax, rmin, rmax
        deq = (o + q * c).astype(np.float32)                        # the device's float dequantisation
        ex = boxes.reshape(-1, 4, 3)
        assert (deq[:, 0] <= ex[:, 0]).all() and (deq[:, 2] <= ex[:, 2]).all(), "quantised minimum above the exact one"
        assert (deq[:, 1] >= ex[:, 1]).all() and (deq[:, 3] >= ex[:, 3]).all(), "quantised maximum below the exact one"
        assert (np.abs(deq - ex) <= 3.01 * c + 1e-6).all(), "quantised box inflated by more than three cells"
    return height


@pytest.mark.parametrize("scene", ["cornell_demo", "cornell_rc", "chess"])
@pytest.mark.parametrize("topology", ["sah", "reference"])
def test_tree_invariants(pkg, hip, monkeypatch, scene, topology):
    if topology == "reference":
        monkeypatch.setenv("MCPT_BVH", "reference")
    sd = pkg.scenes.chess_scene(width=64, height=64, spp=1) if scene == "chess" else getattr(pkg.scenes, scene)(64, 64, 1)
    info, boxes, children, qboxes = hip.bvh_dump(sd)
    assert info["n_nodes"] == len(boxes) > 0
    h = check_tree(sd, info, boxes, children, qboxes)
    if scene == "chess" and topology == "sah":
        assert info["quantised"] == 1 and h <= 24  # the chess tree fits the 20/24-entry traversal stacks


def test_reinsertion_pass_keeps_the_tree_valid(pkg, hip, monkeypatch):
    """MCPT_BVH_REINSERT: subtrees are re-hung where the area sum grows least; every primitive is still exactly one leaf, boxes are
    exact unions, and the tree stays within the stack class it had (check_tree), with a smaller sum of inner-node areas."""
    def area_sum(boxes, children):
        b = boxes.reshape(-1, 2, 6).astype(np.float64)
        d = b[:, :, 3:6] - b[:, :, 0:3]
        a = d[:, :, 0] * d[:, :, 1] + d[:, :, 1] * d[:, :, 2] + d[:, :, 2] * d[:, :, 0]
        return float(a[children >= 0].sum())
    sd = pkg.scenes.chess_scene(width=64, height=64, spp=1)
    info0, boxes0, children0, _ = hip.bvh_dump(sd)
    monkeypatch.setenv("MCPT_BVH_REINSERT", "3")
    info, boxes, children, qboxes = hip.bvh_dump(sd)
    h = check_tree(sd, info, boxes, children, qboxes)
    assert info["n_nodes"] == info0["n_nodes"] and info["stack_entries"] == info0["stack_entries"] and h <= info["stack_entries"]
    assert not np.array_equal(children, children0) and area_sum(boxes, children) < 0.99 * area_sum(boxes0, children0)
    for scene in ("cornell_demo", "cornell_rc"):
        s2 = getattr(pkg.scenes, scene)(32, 32, 1)
        check_tree(s2, *hip.bvh_dump(s2))


def test_quantisation_can_be_switched_off(pkg, hip, monkeypatch):
    monkeypatch.setenv("MCPT_QUANT_NODES", "0")
    info, boxes, children, qboxes = hip.bvh_dump(pkg.scenes.cornell_demo(32, 32, 1))
    assert info["quantised"] == 0 and qboxes is None


def test_random_soup_and_single_primitive(pkg, hip):
    rng = np.random.default_rng(5)
    base = pkg.scenes.cornell_rc(32, 32, 1)
    n = 500
    tri = np.zeros(n, dtype=base.triangles.dtype)
    c = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    for k in ("v0", "v1", "v2"):
        tri[k] = c + rng.normal(0, 0.5, (n, 3)).astype(np.float32)
    obj = np.zeros(1, dtype=base.objects.dtype)
    obj["kind"], obj["material"], obj["first_tri"], obj["n_tri"] = 0, 0, 0, n
    sd = pkg.scenes.SceneData(triangles=tri, materials=base.materials[:1].copy(), objects=obj, background=base.background,
                              env_pixels=None, camera=base.camera, rr_rate=base.rr_rate)
    info, boxes, children, qboxes = hip.bvh_dump(sd)
    assert info["n_nodes"] == n - 1  # a binary tree over n single-primitive leaves
    check_tree(sd, info, boxes, children, qboxes)
    one = pkg.scenes.SceneData(triangles=tri[:1].copy(), materials=sd.materials, objects=obj.copy(), background=sd.background,
                               env_pixels=None, camera=sd.camera, rr_rate=sd.rr_rate)
    one.objects["n_tri"] = 1
    info, boxes, children, qboxes = hip.bvh_dump(one)
    assert info["n_nodes"] == 0 and info["root"] == ~0


def test_instanced_soldiers_share_one_subtree(pkg, hip, monkeypatch):
    """Node instancing (host SAH builder): the 14 soldiers (one OBJ at 14 translations, main.cpp:248-271) become 14 instance leaves over ONE
    shared subtree; every other invariant holds for each instance with its own triangles.  Opt-in (measured slower than the plain
    tree: a memory feature)."""
    sd = pkg.scenes.chess_scene(width=64, height=64, spp=1)
    plain, *_ = hip.bvh_dump(sd)
    assert plain["n_instances"] == 0 and plain["n_nodes"] == 38457  # default: off
    monkeypatch.setenv("MCPT_INSTANCING", "1")
    info, boxes, children, qboxes = hip.bvh_dump(sd)
    assert info["n_instances"] == 14 and info["n_leaf_prims"] == len(sd.triangles) + len(sd.objects)
    assert info["n_nodes"] == (2560 - 1) + (14 + 2312 + 302 + 2 + 2 - 1)  # one soldier subtree + the tree over everything else
    assert len(set(info["inst_root_first"][:, 0].tolist())) == 1 and sorted(info["inst_root_first"][:, 1].tolist()) == [2560 * k for k in range(14)]
    h = check_tree(sd, info, boxes, children, qboxes)
    assert h <= info["stack_entries"] <= 32
    monkeypatch.setenv("MCPT_INSTANCING", "0")
    assert hip.bvh_dump(sd)[0]["n_instances"] == 0
    high = pkg.scenes.chess_high(64, 64, 1)
    assert hip.bvh_dump(high)[0]["n_instances"] == 0
    monkeypatch.setenv("MCPT_INSTANCING", "1")
    info, boxes, children, qboxes = hip.bvh_dump(high)
    assert info["n_instances"] == 14 and info["n_nodes"] == (20480 - 1) + (14 + 9248 + 302 + 2 + 2 - 1)
    check_tree(high, info, boxes, children, qboxes)
