"""Pins the CHESS configuration's scene assembly, camera and depth of field against the one chess artefact the reference holds.

`final_render_result_sky_with_dof.png` (copied as data to tests/golden/) was rendered by the reference's authors with conf.json as
shipped -- camera, DoF, soldier rows, king, diamond, floor: conf.json, main.cpp:131-328, Renderer.cpp:44-76, Camera.hpp:17-24 -- and an
environment map (models/envoMaps/sky.png) that is missing from the snapshot.  Its RADIOMETRY therefore cannot be reproduced (chess
radiometry stays "parity unpinned"), but its GEOMETRY can be checked: where the silhouettes of the objects fall on the screen, and how
blurred they are, is decided by the scene assembly, the camera and the lens model alone.

From the oracle (which the GPU kernels reproduce bit for bit): the primary visibility of the frame at 480x270, every DoF sample's camera
ray -> the object class it hits (sky / soldiers / light / floor / king / diamond), i.e. per-pixel coverage maps, and their gradient
magnitude (the "silhouette map").  From the PNG: the gradient magnitude of the linearised luminance, box-downsampled 4x.
  (a) the king (in focus, so its outline is sharp): windowed cross-correlation of the king's silhouette with the PNG's gradient map
      peaks at zero shift (+-1 px of 480x270; the verdict asked for +-2);
  (b) the far edge of the floor at the left and right image borders: within 1 px (of 270 rows) of the PNG's strongest darkening;
  (c) the normalised cross-correlation of the whole silhouette map with the PNG's gradient map, and its peak over +-3 px shifts at 0.
Negative controls -- the same features from deliberately wrong configurations -- must fail: fov +-3 deg, soldier spacing +-5 %, depth of
field off, aperture x2 and /2, focus at 600 instead of 3036.98, camera 30 units higher, king moved by 40 units.
NOT detectable, and said so instead of asserted: focusDistance +-10 %.  The near pawns (214 units from a camera focused at 3037) are
blurred by R (f/d - 1) / (pixel size at f) pixels, which for d << f is R / (d x pixel angle), independent of f; the king's blur changes
from 0.5 to 0.2 / 0.7 px at 1080p, invisible at 480x270.  The test prints the NCC of those two runs: equal to the baseline's to 3 digits.
"""
import json
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
W, H = 480, 270


def _grad(a):
    gy, gx = np.gradient(a.astype(np.float64), axis=(0, 1))
    return np.sqrt(gx * gx + gy * gy)


def _ncc(x, y, mask=None):
    if mask is not None:
        x, y = x[mask], y[mask]
    x = x - x.mean()
    y = y - y.mean()
    return float((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum()))


def _scan(E, G, R, win=None):
    """Best (ncc, dy, dx) of E shifted by up to R pixels against G (borders of width R left out)."""
    best = None
    for dy in range(-R, R + 1):
        for dx in range(-R, R + 1):
            m = np.zeros(E.shape, bool)
            m[R:-R, R:-R] = True
            if win is not None:
                m &= win
            v = _ncc(np.roll(np.roll(E, dy, 0), dx, 1), G, m)
            if best is None or v > best[0]:
                best = (v, dy, dx)
    return best


@pytest.fixture(scope="module")
def png(pkg):
    a = pkg.pngio.read_png(os.path.join(GOLDEN, "reference_final_render_result_sky_with_dof.png"))[:, :, :3]
    assert a.shape == (1080, 1920, 3)
    lin = ((a.astype(np.float64) + 0.5) / 255.0) ** (1 / 0.45)  # Renderer.cpp:99-101 truncates
    lum_full = lin @ np.array([0.2126, 0.7152, 0.0722])
    lum = lum_full.reshape(H, 4, W, 4).mean(axis=(1, 3))
    return {"lum_full": lum_full, "grad": _grad(lum)}


def _features(pkg, oracle, png, change=None, spp=32):
    """Coverage maps of the (possibly modified) conf.json scene and the features compared with the PNG."""
    conf = json.loads(json.dumps(pkg.scenes.DEFAULT_CONF))
    if change:
        section, key, value = change
        conf[section][key] = value
    sd = pkg.scenes.chess_scene(conf, width=W, height=H, spp=1)
    hits = oracle.OracleScene(sd).primary_hits(spp, seed=1)
    n_obj = len(sd.objects)  # Scene::Add order (main.cpp:248-316): soldiers..., light, floor, king, diamond
    cls = np.zeros(len(sd.triangles) + 1, np.int32)
    for k, ob in enumerate(sd.objects):
        cls[ob["first_tri"]:ob["first_tri"] + ob["n_tri"]] = 1 if k < n_obj - 4 else (2, 3, 4, 5)[k - (n_obj - 4)]
    c = np.where(hits < 0, 0, cls[np.clip(hits, 0, None)])
    cov = np.stack([(c == k).mean(axis=2) for k in range(6)], -1)
    E = np.sqrt(sum(_grad(cov[..., k]) ** 2 for k in range(6)))
    G = png["grad"]
    out = {"ncc": _ncc(E, G), "global_peak": _scan(E, G, 3)}
    king = cov[..., 4]
    ys, xs = np.where(king > 0.5)
    win = np.zeros(king.shape, bool)
    win[max(ys.min() - 8, 0):ys.max() + 9, max(xs.min() - 10, 0):xs.max() + 11] = True
    out["king_peak"] = _scan(_grad(king), G, 6, win)
    out["king_box"] = (int(xs.min()), int(xs.max()), int(ys.min()), int(ys.max()))
    # far edge of the floor in the eight leftmost / rightmost columns: first row where the floor's coverage crosses 1/2
    hz = []
    for cols in (slice(0, 8), slice(W - 8, W)):
        f = cov[:, cols, 3].mean(axis=1)
        cr = [y + (0.5 - f[y]) / (f[y + 1] - f[y]) for y in range(H - 1) if (f[y] - 0.5) * (f[y + 1] - 0.5) < 0]
        hz.append(cr[0] if cr else float("nan"))
    out["horizon"] = hz
    return out


def _png_horizon(png):
    """Row (in 270-row pixel-centre coordinates) of the strongest darkening in the 32 leftmost / rightmost columns of the 1080p PNG: the
    sky above the far edge of the floor is bright, the floor (a mirror with base reflectance 0.1 outside the checkerboard) dark."""
    out = []
    for cols in (slice(0, 32), slice(1920 - 32, 1920)):
        L = np.log(png["lum_full"][:, cols].mean(axis=1) + 1e-3)
        L = np.convolve(L, np.ones(4) / 4, mode="same")
        r = 8 + int(np.argmin(np.diff(L)[8:-8]))  # the edge lies between full-resolution rows r and r + 1 (the filter's borders left out)
        out.append((r + 1.0) / 4.0 - 0.5)
    return out


def test_chess_geometry_camera_and_dof_match_the_reference_image(pkg, oracle, png, capsys):
    base = _features(pkg, oracle, png, spp=64)
    hz_png = _png_horizon(png)
    with capsys.disabled():
        print("\n[chess pin] silhouette NCC %.4f, global peak %s, king peak %s, king box %s, floor far edge rows oracle %s / png %s"
              % (base["ncc"], base["global_peak"], base["king_peak"], base["king_box"], np.round(base["horizon"], 2), np.round(hz_png, 2)))
    # (c) whole frame
    assert base["ncc"] >= 0.40, base["ncc"]
    assert base["global_peak"][1:] == (0, 0), base["global_peak"]
    # (a) the king
    assert base["king_peak"][0] >= 0.45 and max(abs(base["king_peak"][1]), abs(base["king_peak"][2])) <= 1, base["king_peak"]
    # (b) far edge of the floor at both borders
    for o, p in zip(base["horizon"], hz_png):
        assert abs(o - p) <= 1.0, (base["horizon"], hz_png)

    # ---- negative controls: every one of them must be told apart from the shipped configuration
    DC = pkg.scenes.DEFAULT_CONF
    wrong = {
        "fov +3": (("camera", "fov", DC["camera"]["fov"] + 3), 0.30),
        "fov -3": (("camera", "fov", DC["camera"]["fov"] - 3), 0.30),
        "soldier spacing +5 %": (("scene", "soldierZSpacing", DC["scene"]["soldierZSpacing"] * 1.05), 0.30),
        "soldier spacing -5 %": (("scene", "soldierZSpacing", DC["scene"]["soldierZSpacing"] * 0.95), 0.30),
        "depth of field off": (("camera", "useDOF", False), 0.36),
        "aperture x2": (("camera", "apertureRadius", DC["camera"]["apertureRadius"] * 2), base["ncc"] - 0.015),
        "aperture /2": (("camera", "apertureRadius", DC["camera"]["apertureRadius"] / 2), base["ncc"] - 0.015),
        "focus at 600": (("camera", "focusDistance", 600.0), 0.36),
        "camera 30 units higher": (("camera", "position", [278, 180, -2550]), 0.30),
    }
    for name, (change, bound) in wrong.items():
        f = _features(pkg, oracle, png, change)
        with capsys.disabled():
            print("[chess pin] control %-24s NCC %.4f (bound < %.3f), king peak %s, floor far edge %s" % (name, f["ncc"], bound, f["king_peak"][1:], np.round(f["horizon"], 2)))
        assert f["ncc"] < bound and f["ncc"] < base["ncc"] - 0.015, (name, f["ncc"])
        if name == "camera 30 units higher":
            assert all(abs(o - p) > 1.0 for o, p in zip(f["horizon"], hz_png)), f["horizon"]
    f = _features(pkg, oracle, png, ("scene", "kingPosition", [40, 0, 0]))  # 3 px at 480x270
    assert max(abs(f["king_peak"][1]), abs(f["king_peak"][2])) >= 2, f["king_peak"]
    # not detectable (see the module docstring): focusDistance +-10 %
    for s in (1.1, 0.9):
        f = _features(pkg, oracle, png, ("camera", "focusDistance", DC["camera"]["focusDistance"] * s))
        with capsys.disabled():
            print("[chess pin] focusDistance x%.1f: NCC %.4f (undetectable at this resolution, by construction of the lens model)" % (s, f["ncc"]))
