"""Minimal PNG read/write (8-bit RGB/RGBA, non-interlaced) and the reference's tone map.

Host-side output path, not the hot path.  Mirrors Renderer.cpp:95-109 (gamma 0.45, 8-bit truncation,
alpha 255, PNG out) and the decode half of Scene::loadEnvMap (Scene.hpp:39-57); the reference uses the
vendored lodepng for both, this file only needs zlib from the standard library.
"""
from __future__ import annotations

import struct
import zlib

import numpy as np


def tonemap_u8(fb):
    """Renderer.cpp:95-103: raw = (unsigned char) clamp(0, 255, 255 * pow(c, 0.45f)).  NaN -> 255 (std::min/max)."""
    fb = np.asarray(fb, dtype=np.float32)
    with np.errstate(invalid="ignore"):
        v = np.float32(255) * np.power(fb, np.float32(0.45), dtype=np.float32)
    v = np.where(v < np.float32(255), v, np.float32(255))  # std::min(hi, v): NaN -> hi
    v = np.where(np.float32(0) < v, v, np.float32(0))      # std::max(lo, .)
    return v.astype(np.uint8)  # truncation toward zero


def write_png(path, rgb_u8):
    """rgb_u8: (H, W, 3|4) uint8."""
    a = np.ascontiguousarray(rgb_u8, dtype=np.uint8)
    h, w, c = a.shape
    ctype = {3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), dtype=np.uint8), a.reshape(h, w * c)], axis=1).tobytes()

    def chunk(tag, data):
        body = tag + data
        return struct.pack(">I", len(data)) + body + struct.pack(">I", zlib.crc32(body) & 0xFFFFFFFF)

    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n")
        fh.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)))
        fh.write(chunk(b"IDAT", zlib.compress(raw, 6)))
        fh.write(chunk(b"IEND", b""))


def read_png(path):
    """Returns (H, W, C) uint8 for 8-bit gray/RGB/RGBA/gray-alpha non-interlaced PNGs."""
    with open(path, "rb") as fh:
        data = fh.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("not a PNG: %s" % path)
    pos, idat, hdr = 8, [], None
    while pos < len(data):
        (n,) = struct.unpack(">I", data[pos:pos + 4])
        tag = data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        pos += 12 + n
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat.append(body)
        elif tag == b"IEND":
            break
    w, h, depth, ctype, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or ctype not in (0, 2, 4, 6):
        raise ValueError("unsupported PNG layout (depth %d, colour type %d, interlace %d)" % (depth, ctype, interlace))
    c = {0: 1, 2: 3, 4: 2, 6: 4}[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8).reshape(h, 1 + w * c)
    out = np.zeros((h, w * c), dtype=np.uint8)
    prev = np.zeros(w * c, dtype=np.int32)
    for y in range(h):
        ft = int(raw[y, 0])
        line = raw[y, 1:].astype(np.int32)
        if ft == 0:
            cur = line
        elif ft == 2:
            cur = (line + prev) & 255
        else:
            cur = np.zeros(w * c, dtype=np.int32)
            for x in range(w * c):
                a = cur[x - c] if x >= c else 0
                b = prev[x]
                cc = prev[x - c] if x >= c else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - cc
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - cc)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else cc)
                cur[x] = (line[x] + pred) & 255
        out[y] = cur.astype(np.uint8)
        prev = cur
    return out.reshape(h, w, c)


def load_env_map(path):
    """Scene::loadEnvMap (Scene.hpp:39-57): RGBA8 -> float3 / 255.  Returns (H, W, 3) float32 or None on failure."""
    try:
        img = read_png(path)
    except (OSError, ValueError, zlib.error):
        return None  # the reference prints the error and keeps the constant background
    if img.shape[2] == 1:
        img = np.repeat(img, 3, axis=2)
    return (img[:, :, :3].astype(np.float32) / np.float32(255.0)).astype(np.float32)


def psnr_u8(a, b):
    """10*log10(255^2 / MSE) over all channels of two uint8 images."""
    d = a.astype(np.float64) - b.astype(np.float64)
    mse = float(np.mean(d * d))
    return float("inf") if mse == 0 else 10.0 * np.log10(255.0 * 255.0 / mse)
