"""Host-side logic that needs no GPU: scene assembly (the mirror of main.cpp), PNG I/O, configuration quirks."""
import json
import os

import numpy as np
import pytest


def test_cornell_demo_scene_layout(pkg):
    sd = pkg.scenes.cornell_demo()
    # main.cpp:106-125: 6 meshes (12+10+10+2+2+2 = 32... floor.obj holds floor, ceiling and back wall) + 3 spheres
    assert len(sd.objects) == 9 and len(sd.triangles) == 32
    assert sd.objects["kind"].tolist() == [0] * 6 + [1] * 3
    assert sd.objects["n_tri"][:6].tolist() == [6, 10, 10, 2, 2, 2]
    cam = sd.camera
    assert int(cam["width"]) == 384 and int(cam["height"]) == 384 and float(cam["fov"]) == 40.0
    R = cam["orientation"].reshape(3, 3)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-6)
    assert np.allclose(R[:, 2], [0, 0, 1])  # forward = +z (main.cpp:29-30)
    # light emission 3.9 * (47.83, 38.57, 31.08) (main.cpp:100-104)
    light = sd.materials[sd.objects["material"][5]]
    assert np.allclose(light["emission"], 3.9 * np.array([47.8348, 38.5664, 31.0808]), rtol=1e-4)
    assert sd.rr_rate == pytest.approx(0.7) and sd.n_dir_sample == 4 and sd.spp == 2048


def test_chess_scene_layout_and_conf_quirks(pkg):
    sd = pkg.scenes.chess_scene()
    # 14 soldiers x 2560 + light 2 + floor 2 + king 2312 + diamond 302 (mis-grouped quads)
    assert len(sd.objects) == 18
    assert sd.objects["n_tri"].tolist() == [2560] * 14 + [2, 2, 2312, 302]
    assert len(sd.triangles) == 38458
    assert int(sd.camera["use_dof"]) == 1 and float(sd.camera["focal_distance"]) == pytest.approx(3036.98)
    assert sd.rr_rate == pytest.approx(0.4, rel=1e-6)
    assert sd.n_dir_sample == 4  # conf "directLightSample": 32 is never read by the reference
    # soldier materials interleave left/right rows (main.cpp:263-270)
    mats = sd.materials[sd.objects["material"][:14]]
    assert mats["type"][0::2].tolist() == [2] * 7 and mats["type"][1::2].tolist() == [1] * 7
    # the floor is the only textured material and carries its uv's (main.cpp:282-285, Triangle.hpp:115-122)
    floor = sd.objects[15]
    assert sd.materials[floor["material"]]["textured"] == 1
    ft = sd.triangles[floor["first_tri"]:floor["first_tri"] + 2]
    assert ft["t1"].max() == 1.0 and sd.triangles["t0"][:100].max() == 0.0
    # light translated to (278, 1300, 0) + light.obj, brightness 100
    lt = sd.triangles[sd.objects[14]["first_tri"]]
    assert lt["v0"][1] == pytest.approx(1848.7)
    assert np.allclose(sd.materials[sd.objects["material"][14]]["emission"], 100 * np.array([47.8348, 38.5664, 31.0808]), rtol=1e-4)

    conf = json.loads(json.dumps(pkg.scenes.DEFAULT_CONF))
    conf["scene"]["addDiamond"] = False       # main.cpp:197-199 only checks that the key is a boolean
    conf["scene"]["lightBrightness"] = 100    # an integer is ignored (main.cpp:279: is_number_float)
    conf["scene"]["RussianRouletteRate"] = 1.5
    sd2 = pkg.scenes.chess_scene(conf)
    assert len(sd2.objects) == 18
    assert np.allclose(sd2.materials[sd2.objects["material"][14]]["emission"], [47.8348, 38.5664, 31.0808], rtol=1e-4)
    assert sd2.rr_rate == pytest.approx(0.99, rel=1e-6)  # Scene::setRrRate clamps (Scene.hpp:110-113)
    del conf["scene"]["addDiamond"]
    assert len(pkg.scenes.chess_scene(conf).objects) == 17


def test_obj_vertex_stream_groups_by_three(pkg):
    """Triangle.hpp:99-124 ignores the index buffer: diamond.obj (174 triangles + 96 quads) -> 906 vertices -> 302."""
    p, t = pkg.scenes.load_obj_vertex_stream(os.path.join(pkg.scenes.ASSETS, "diamond.obj"))
    assert len(p) == 174 * 3 + 96 * 4 == 906
    assert len(pkg.scenes.mesh_triangles(os.path.join(pkg.scenes.ASSETS, "diamond.obj"))) == 302
    p, t = pkg.scenes.load_obj_vertex_stream(os.path.join(pkg.scenes.ASSETS, "bottom.obj"))
    assert p.shape == (6, 3) and t.tolist() == [[0, 0], [1, 1], [0, 1], [0, 0], [1, 0], [1, 1]]


def test_png_roundtrip_and_env_loader(pkg, tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(17, 23, 3), dtype=np.uint8)
    path = str(tmp_path / "x.png")
    pkg.pngio.write_png(path, img)
    assert np.array_equal(pkg.pngio.read_png(path), img)
    env = pkg.pngio.load_env_map(path)
    assert env.shape == (17, 23, 3) and env.dtype == np.float32 and env.max() <= 1.0
    assert pkg.pngio.load_env_map(str(tmp_path / "missing.png")) is None  # Scene.hpp:42-46: error -> constant background
    # all five PNG filter types are decoded (the reference's demo image uses them)
    ref = pkg.pngio.read_png(os.path.join(os.path.dirname(__file__), "golden", "reference_cornellbox_demo.png"))
    assert ref.shape == (384, 384, 3) and 40 < ref.mean() < 120
    assert pkg.pngio.psnr_u8(img, img) == float("inf")


def test_pod_layouts_match_header(pkg):
    s = pkg.scenes
    assert s.TRI_DTYPE.itemsize == 60 and s.MAT_DTYPE.itemsize == 44 and s.OBJ_DTYPE.itemsize == 32 and s.CAM_DTYPE.itemsize == 72
