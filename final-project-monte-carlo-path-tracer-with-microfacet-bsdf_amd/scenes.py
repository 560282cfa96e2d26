"""Scene assembly: the Python mirror of the reference's app glue (src/main.cpp).

This is host-side plumbing for bench.py and the tests: it produces the flat POD arrays
(`include/mcpt.h`: mcpt_triangle / mcpt_material / mcpt_object / mcpt_camera) that
`mcpt_scene_create` / `mcpt_render` consume.  Nothing here is on the rendering hot path.

Reference behaviour mirrored (file:line are relative to /root/reference/src):
  * material presets ..................... main.cpp:34-97, Material.hpp:245-257
  * DEMO Cornell-box scene ............... main.cpp:99-129
  * conf.json chess scene ................ main.cpp:131-316  (insertion order: soldiers L0,R0,L1,R1..., light, floor, king, diamond)
  * OBJ -> triangles ..................... Triangle.hpp:83-135 + OBJ_Loader.hpp:363-619,633-729:
        the loader's per-face vertex stream is grouped by threes and the index buffer is ignored,
        so quads are NOT triangulated (diamond.obj -> 302 mis-grouped triangles).  Reproduced as is.
  * camera ............................... Camera.hpp:17-24, main.cpp:324-328
  * ignored conf keys .................... scene.directLightSample, scene.model_quality, renderer.path,
        renderer.parrallelism; scene.addDiamond only checks presence (main.cpp:191,197-202)
  * chess_scene(fixed=True) ("--fixed" of host/RayTracing): the configuration as its keys READ instead of as the
        shipped main executes it -- scene.directLightSample is applied (Scene::setDirectLightSample, Scene.hpp:114),
        scene.model_quality selects low_*/high_* models (main.cpp:24-26 composes the paths before the key is read,
        :200-202), scene.addDiamond:false leaves the diamond out (main.cpp:197-199 tests presence only).
"""
from __future__ import annotations

import json
import os
from dataclasses import dataclass, field

import numpy as np

f32 = np.float32

# ---- POD layouts of include/mcpt.h
TRI_DTYPE = np.dtype([("v0", f32, 3), ("v1", f32, 3), ("v2", f32, 3), ("t0", f32, 2), ("t1", f32, 2), ("t2", f32, 2)])
MAT_DTYPE = np.dtype([("type", np.int32), ("textured", np.int32), ("roughness", f32), ("iorA", f32), ("iorB", f32),
                      ("base_reflectance", f32, 3), ("emission", f32, 3)])
OBJ_DTYPE = np.dtype([("kind", np.int32), ("material", np.int32), ("first_tri", np.int32), ("n_tri", np.int32),
                      ("center", f32, 3), ("radius", f32)])
assert TRI_DTYPE.itemsize == 60 and MAT_DTYPE.itemsize == 44 and OBJ_DTYPE.itemsize == 32

SMOOTH_CONDUCTOR, ROUGH_CONDUCTOR, SMOOTH_DIELECTRIC, ROUGH_DIELECTRIC = 0, 1, 2, 3  # Material.hpp:13-18
OBJ_MESH, OBJ_SPHERE = 0, 1

ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets", "models")


# --------------------------------------------------------------------------- OBJ
def load_obj_vertex_stream(path):
    """Per-face vertex stream of the FIRST mesh, as objl::Loader builds it (OBJ_Loader.hpp:458-521,633-729).

    Returns (positions[n,3] float32, texcoords[n,2] float32).  A new mesh starts at an `o`/`g` line once
    the current one has faces (OBJ_Loader.hpp:415-451); the reference only reads LoadedMeshes[0]
    (Triangle.hpp:91)."""
    pos, tex = [], []
    out_p, out_t = [], []
    listening = False
    with open(path, "r") as fh:
        for raw in fh:
            line = raw.rstrip("\r\n")
            toks = line.split()
            if not toks:
                continue
            head = toks[0]
            if head in ("o", "g") or line[:1] == "g":
                if not listening:
                    listening = True
                elif out_p:
                    break  # second mesh: ignored by the reference
            if head == "v":
                pos.append([f32(float(t)) for t in toks[1:4]])
            elif head == "vt":
                tex.append([f32(float(t)) for t in toks[1:3]])
            elif head == "f":
                for vert in toks[1:]:
                    parts = vert.split("/")
                    idx = int(parts[0])
                    p = pos[idx + len(pos)] if idx < 0 else pos[idx - 1]
                    t = [f32(0), f32(0)]
                    if len(parts) >= 2 and parts[1] != "":
                        ti = int(parts[1])
                        t = tex[ti + len(tex)] if ti < 0 else tex[ti - 1]
                    out_p.append(p)
                    out_t.append(t)
    return np.asarray(out_p, dtype=f32).reshape(-1, 3), np.asarray(out_t, dtype=f32).reshape(-1, 2)


def mesh_triangles(path, translation=(0, 0, 0), zoom=1.0, textured=False):
    """MeshTriangle::MeshTriangle (Triangle.hpp:83-124): group the vertex stream by 3; v = zoom*vert + translation."""
    p, t = load_obj_vertex_stream(path)
    n = (len(p) // 3) * 3  # the reference would read past the end if the stream were not a multiple of 3
    p = p[:n]
    t = t[:n]
    v = (f32(zoom) * p + np.asarray(translation, dtype=f32)).astype(f32)
    tris = np.zeros(n // 3, dtype=TRI_DTYPE)
    tris["v0"], tris["v1"], tris["v2"] = v[0::3], v[1::3], v[2::3]
    if textured:  # Triangle.hpp:115-122
        tris["t0"], tris["t1"], tris["t2"] = t[0::3], t[1::3], t[2::3]
    return tris


# --------------------------------------------------------------------------- materials
def _mat(mtype, roughness=None, refl=(0, 0, 0), iorA=1.74, iorB=0.1, emission=(0, 0, 0), textured=0):
    m = np.zeros((), dtype=MAT_DTYPE)
    m["type"] = mtype
    m["textured"] = textured
    if roughness is None:  # Material.hpp:252-255
        roughness = 0.2 if mtype == ROUGH_DIELECTRIC else 1.0
    m["roughness"], m["iorA"], m["iorB"] = f32(roughness), f32(iorA), f32(iorB)
    m["base_reflectance"] = np.asarray(refl, dtype=f32)
    m["emission"] = np.asarray(emission, dtype=f32)
    return m


def material_presets():
    """The nine named presets of main.cpp:34-97 (insertion order kept)."""
    return {
        "rough_red_conductor": _mat(ROUGH_CONDUCTOR, 0.1, (1.0, 0.0, 0.0)),
        "rough_white_conductor": _mat(ROUGH_CONDUCTOR, 0.4, (0.725, 0.71, 0.68)),
        "green_mirror": _mat(ROUGH_CONDUCTOR, 0.01, (0.14, 1.0, 0.14)),
        "gold_conductor": _mat(SMOOTH_CONDUCTOR, 0.0001, (1.0, 0.85, 0.57)),
        "silver_mirror": _mat(SMOOTH_CONDUCTOR, 0.001, (0.972, 0.960, 0.915)),
        "smooth_glass": _mat(SMOOTH_DIELECTRIC, 0.01, iorA=1.7, iorB=0.04),
        "smooth_glass_gem": _mat(SMOOTH_DIELECTRIC, 0.001, iorA=1.3, iorB=0.2),
        "clear_rough_plastic": _mat(ROUGH_DIELECTRIC, 0.02, iorA=1.5, iorB=0.01),
        "rough_plastic": _mat(ROUGH_DIELECTRIC, 0.4, iorA=1.5, iorB=0.01),
    }


def light_emission(scale):
    """main.cpp:100-104 / 303-308, evaluated in float like the Eigen expression."""
    a = f32(8.0) * np.array([f32(0.747) + f32(0.058), f32(0.747) + f32(0.258), f32(0.747)], dtype=f32)
    b = f32(15.6) * np.array([f32(0.740) + f32(0.287), f32(0.740) + f32(0.160), f32(0.740)], dtype=f32)
    c = f32(18.4) * np.array([f32(0.737) + f32(0.642), f32(0.737) + f32(0.159), f32(0.737)], dtype=f32)
    return (f32(scale) * ((a + b).astype(f32) + c).astype(f32)).astype(f32)


# --------------------------------------------------------------------------- camera
CAM_DTYPE = np.dtype([("width", np.int32), ("height", np.int32), ("fov", f32), ("position", f32, 3),
                      ("orientation", f32, 9), ("use_dof", np.int32), ("focal_distance", f32), ("aperture_radius", f32)])
assert CAM_DTYPE.itemsize == 72


def _normalized(v):
    v = v.astype(f32)
    z = f32(v[0] * v[0] + f32(v[1] * v[1] + v[2] * v[2]))
    return (v / np.sqrt(z, dtype=f32)).astype(f32) if z > 0 else v


def _cross(a, b):
    a = a.astype(f32)
    b = b.astype(f32)
    return np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], dtype=f32)


def make_camera(width, height, fov, position, target, up=(0, 1, 0), use_dof=False, focal_distance=100.0,
                aperture_radius=5.0):
    """Camera + Camera::lookAt (Camera.hpp:6-26): orientation columns = left, up, forward."""
    pos = np.asarray(position, dtype=f32)
    forward = _normalized(np.asarray(target, dtype=f32) - pos)
    left = _normalized(_cross(np.asarray(up, dtype=f32), forward))
    new_up = _normalized(_cross(forward, left))
    cam = np.zeros((), dtype=CAM_DTYPE)
    cam["width"], cam["height"], cam["fov"] = width, height, f32(fov)
    cam["position"] = pos
    cam["orientation"] = np.stack([left, new_up, forward], axis=1).astype(f32).reshape(9)  # row-major, columns l/u/f
    cam["use_dof"], cam["focal_distance"], cam["aperture_radius"] = int(bool(use_dof)), f32(focal_distance), f32(aperture_radius)
    return cam


# --------------------------------------------------------------------------- scene container
@dataclass
class SceneData:
    triangles: np.ndarray
    materials: np.ndarray
    objects: np.ndarray
    background: np.ndarray = field(default_factory=lambda: np.zeros(3, dtype=f32))
    env_pixels: np.ndarray | None = None  # (H, W, 3) float32 in [0,1]
    camera: np.ndarray | None = None
    rr_rate: float = 0.7  # Scene.hpp:25
    enable_shadow: bool = True
    n_dir_sample: int = 4  # Scene.hpp:28 -- conf "directLightSample" is never read (main.cpp has no caller of Scene.hpp:114)
    spp: int = 2048  # Renderer.hpp:22
    name: str = ""


class _Builder:
    def __init__(self):
        self.tris, self.mats, self.objs, self.mat_index = [], [], [], {}

    def material(self, key, mat):
        if key not in self.mat_index:
            self.mat_index[key] = len(self.mats)
            self.mats.append(mat)
        return self.mat_index[key]

    def add_mesh(self, tris, mat_id):
        first = sum(len(t) for t in self.tris)
        self.tris.append(tris)
        o = np.zeros((), dtype=OBJ_DTYPE)
        o["kind"], o["material"], o["first_tri"], o["n_tri"] = OBJ_MESH, mat_id, first, len(tris)
        self.objs.append(o)

    def add_sphere(self, center, radius, mat_id):
        o = np.zeros((), dtype=OBJ_DTYPE)
        o["kind"], o["material"], o["center"], o["radius"] = OBJ_SPHERE, mat_id, np.asarray(center, dtype=f32), f32(radius)
        self.objs.append(o)

    def finish(self, **kw):
        return SceneData(triangles=np.concatenate(self.tris).astype(TRI_DTYPE), materials=np.stack(self.mats).astype(MAT_DTYPE),
                         objects=np.stack(self.objs).astype(OBJ_DTYPE), **kw)


# --------------------------------------------------------------------------- DEMO Cornell box (BASELINE config 1)
def cornell_demo(width=384, height=384, spp=2048, assets=ASSETS):
    """main.cpp:99-129 + defaults main.cpp:28-31 (384x384, camera (278,273,-800)->(278,273,0), fov 40, rr 0.7)."""
    P = material_presets()
    b = _Builder()
    cb = os.path.join(assets, "cornellbox")
    light = _mat(ROUGH_CONDUCTOR, emission=light_emission(3.9))

    def mesh(fn, key, mat):
        b.add_mesh(mesh_triangles(os.path.join(cb, fn)), b.material(key, mat))

    # Scene::Add order, main.cpp:117-125
    mesh("floor.obj", "rough_white_conductor", P["rough_white_conductor"])  # `back`
    mesh("shortbox.obj", "green_mirror", P["green_mirror"])
    mesh("tallbox.obj", "rough_plastic", P["rough_plastic"])
    mesh("left.obj", "rough_red_conductor", P["rough_red_conductor"])
    mesh("right.obj", "gold_conductor", P["gold_conductor"])
    mesh("light.obj", "light", light)
    b.add_sphere((400, 90, 3), 80, b.material("smooth_glass", P["smooth_glass"]))
    b.add_sphere((250, 260, 230), 60, b.material("clear_rough_plastic", P["clear_rough_plastic"]))
    b.add_sphere((120, 390, 400), 50, b.material("silver_mirror", P["silver_mirror"]))
    cam = make_camera(width, height, 40, (278, 273, -800), (278, 273, 0), (0, 1, 0), use_dof=False,
                      focal_distance=900, aperture_radius=40)
    return b.finish(camera=cam, rr_rate=0.7, spp=spp, name="cornell_demo")


# --------------------------------------------------------------------------- cornell-rc (BASELINE config 2)
def cornell_rc(width=784, height=784, spp=256, assets=ASSETS):
    """config-1 geometry minus the spheres, every non-light mesh ROUGH_CONDUCTOR."""
    P = material_presets()
    b = _Builder()
    cb = os.path.join(assets, "cornellbox")
    light = _mat(ROUGH_CONDUCTOR, emission=light_emission(3.9))

    def mesh(fn, key, mat):
        b.add_mesh(mesh_triangles(os.path.join(cb, fn)), b.material(key, mat))

    mesh("floor.obj", "rough_white_conductor", P["rough_white_conductor"])
    mesh("shortbox.obj", "green_mirror", P["green_mirror"])
    mesh("tallbox.obj", "rough_white_conductor", P["rough_white_conductor"])
    mesh("left.obj", "rough_red_conductor", P["rough_red_conductor"])
    mesh("right.obj", "rough_white_conductor", P["rough_white_conductor"])
    mesh("light.obj", "light", light)
    cam = make_camera(width, height, 40, (278, 273, -800), (278, 273, 0), (0, 1, 0))
    return b.finish(camera=cam, rr_rate=0.7, spp=spp, name="cornell_rc")


# --------------------------------------------------------------------------- chess scene (BASELINE configs 3-5)
DEFAULT_CONF = {
    "camera": {"width": 1920, "height": 1080, "fov": 70, "position": [278, 150, -2550], "target": [278, 0, 0],
               "up": [0, 1, 0], "useDOF": True, "focusDistance": 3036.98, "apertureRadius": 10},
    "renderer": {"spp": 32},
    "scene": {
        "includeShadow": True, "directLightSample": 32, "RussianRouletteRate": 0.4,
        # models/envoMaps/sky.png is missing from the reference snapshot (.MISSING_LARGE_BLOBS); the
        # constant colour is the alternative conf.json's own note_1 documents.
        "envMap": [0.235294, 0.67451, 0.843137],
        "model_quality": "low", "kingPosition": [0, 0, 0],
        "soldierLeftRowPosition": [-559, 0, -200], "soldierRightRowPosition": [160, 0, -200],
        "soldierXSpacing": 0, "soldierYSpacing": 0, "soldierZSpacing": -356, "soldierCountPerRow": 7,
        "kingMaterial": "gold_conductor",
        "soldierMaterials": ["smooth_glass"] * 7 + ["rough_white_conductor"] * 7,
        "wallMaterial": "rough_white_conductor", "floorMaterial": "silver_mirror", "floor_isTextured": True,
        "lightPosition": [278, 1300, 0], "lightBrightness": 100.0, "addDiamond": True,
    },
}


def _is_v3(d):
    return isinstance(d, list) and len(d) == 3 and all(isinstance(e, (int, float)) and not isinstance(e, bool) for e in d)


def chess_scene(conf=None, width=None, height=None, spp=None, assets=ASSETS, env_loader=None, fixed=False):
    """The conf.json scene, main.cpp:131-328.  `conf` is a parsed conf.json (dict) or a path; None = the shipped
    conf.json with envMap replaced by its documented constant colour (sky.png is missing)."""
    if conf is None:
        conf = DEFAULT_CONF
    elif isinstance(conf, (str, os.PathLike)):
        with open(conf, "r") as fh:
            conf = json.load(fh)
    P = material_presets()
    b = _Builder()
    # defaults, main.cpp:28-31,137-144
    w, h = 384, 384
    cam_pos, cam_target, cam_up = [278, 273, -800], [278, 273, 0], [0, 1, 0]
    fov, use_dof, focal, aperture = 40.0, False, 100.0, 5.0
    out_spp = 2048
    rr, shadow = 0.7, True
    background = np.zeros(3, dtype=f32)
    env_pixels = None
    king_pos, king_mat = [0, 0, 0], "rough_plastic"
    light_pos, floor_mat, brightness = [0, 200, 0], "rough_plastic", 1.0
    use_diamond = False
    floor_textured = False
    n_dir = 4  # Scene.hpp:28
    quality = "low"  # main.cpp:24: the default, which the shipped executable never leaves

    cc = conf.get("camera")
    if cc is not None:
        if isinstance(cc.get("width"), (int, float)): w = int(cc["width"])
        if isinstance(cc.get("height"), (int, float)): h = int(cc["height"])
        if isinstance(cc.get("fov"), (int, float)): fov = cc["fov"]
        if _is_v3(cc.get("position")): cam_pos = cc["position"]
        if _is_v3(cc.get("target")): cam_target = cc["target"]
        if _is_v3(cc.get("up")): cam_up = cc["up"]
        if isinstance(cc.get("useDOF"), bool): use_dof = cc["useDOF"]
        if use_dof and isinstance(cc.get("focusDistance"), (int, float)): focal = cc["focusDistance"]
        if use_dof and isinstance(cc.get("apertureRadius"), (int, float)): aperture = cc["apertureRadius"]
    cr = conf.get("renderer")
    if cr is not None and isinstance(cr.get("spp"), (int, float)):
        out_spp = int(cr["spp"])
    cs = conf.get("scene")
    if cs is not None:
        if isinstance(cs.get("addDiamond"), bool): use_diamond = cs["addDiamond"] if fixed else True  # main.cpp:197-199: presence, not value
        if fixed and isinstance(cs.get("directLightSample"), int) and cs["directLightSample"] > 0: n_dir = int(cs["directLightSample"])
        if fixed and cs.get("model_quality") in ("low", "high"): quality = cs["model_quality"]
        if isinstance(cs.get("includeShadow"), bool): shadow = cs["includeShadow"]
        if isinstance(cs.get("RussianRouletteRate"), (int, float)): rr = min(float(f32(cs["RussianRouletteRate"])), float(f32(0.99)))
        env = cs.get("envMap")
        if isinstance(env, str):
            if env_loader is not None:
                env_pixels = env_loader(env)  # Scene::loadEnvMap, Scene.hpp:39-57; failure => black background
        elif _is_v3(env):
            background = np.asarray(env, dtype=f32)
        if _is_v3(cs.get("kingPosition")): king_pos = cs["kingPosition"]
        if isinstance(cs.get("kingMaterial"), str): king_mat = cs["kingMaterial"]
        if isinstance(cs.get("floorMaterial"), str):  # main.cpp:282-285 (sets the shared preset's flag)
            floor_mat = cs["floorMaterial"]
            floor_textured = bool(cs.get("floor_isTextured"))
            P[floor_mat]["textured"] = int(floor_textured)
        if all(k in cs for k in ("soldierLeftRowPosition", "soldierRightRowPosition", "soldierMaterials")):
            lrow, rrow = cs["soldierLeftRowPosition"], cs["soldierRightRowPosition"]
            xs, ys, zs = f32(cs["soldierXSpacing"]), f32(cs["soldierYSpacing"]), f32(cs["soldierZSpacing"])
            count = int(cs["soldierCountPerRow"])
            names = cs["soldierMaterials"]
            soldier = os.path.join(assets, quality + "_soldier.obj")  # (not fixed: model_quality is ineffective, main.cpp:24-26)
            for i in range(count):  # main.cpp:248-271
                off = np.array([f32(i) * xs, f32(i) * ys, f32(i) * zs], dtype=f32)
                lpos = (np.asarray(lrow, dtype=f32) + off).astype(f32)
                rpos = (np.asarray(rrow, dtype=f32) + off).astype(f32)
                lm = names[i] if i < len(names) else "rough_plastic"
                rm = names[i + count] if (i + count) < len(names) else "rough_plastic"
                b.add_mesh(mesh_triangles(soldier, lpos), b.material(lm, P[lm]))
                b.add_mesh(mesh_triangles(soldier, rpos), b.material(rm, P[rm]))
        if _is_v3(cs.get("lightPosition")): light_pos = cs["lightPosition"]
        lb = cs.get("lightBrightness")
        if isinstance(lb, float): brightness = lb  # main.cpp:279: is_number_float only
    light = _mat(ROUGH_CONDUCTOR, emission=light_emission(brightness))
    # main.cpp:313-316: light, floor, king, diamond (the wall is built but never added)
    b.add_mesh(mesh_triangles(os.path.join(assets, "light.obj"), light_pos), b.material("light", light))
    b.add_mesh(mesh_triangles(os.path.join(assets, "bottom.obj"), textured=bool(P[floor_mat]["textured"])),
               b.material(floor_mat, P[floor_mat]))
    b.add_mesh(mesh_triangles(os.path.join(assets, quality + "_king.obj"), king_pos), b.material(king_mat, P[king_mat]))
    if use_diamond:
        b.add_mesh(mesh_triangles(os.path.join(assets, "diamond.obj")), b.material("smooth_glass_gem", P["smooth_glass_gem"]))
    if width is not None: w = width
    if height is not None: h = height
    if spp is not None: out_spp = spp
    cam = make_camera(w, h, fov, cam_pos, cam_target, cam_up, use_dof, focal, aperture)
    return b.finish(camera=cam, rr_rate=rr, enable_shadow=shadow, spp=out_spp, background=background,
                    env_pixels=env_pixels, n_dir_sample=n_dir, name="chess" if quality == "low" else "chess_high")


def chess_high(width=None, height=None, spp=None, n_dir=4, assets=ASSETS):
    """The conf.json scene with model_quality "high" honoured (fixed mode): 296 274 triangles instead of 38 458.
    n_dir stays a parameter (4 = what the shipped executable runs; conf.json says 32)."""
    conf = json.loads(json.dumps(DEFAULT_CONF))
    conf["scene"]["model_quality"] = "high"
    conf["scene"]["directLightSample"] = int(n_dir)
    return chess_scene(conf, width=width, height=height, spp=spp, assets=assets, fixed=True)
