"""Deterministic procedural stand-ins for the reference's image inputs.

The reference binds a 4-layer 256x256 RGBA8 texture array (cobblestone, sand, check,
grass) and a 4096x3072 horizontal-cross cube-map skybox (raypng.c:74-81); both are
decoded from PNG to RGBA8 with A=255 (opencl_wrap.c:189-349).  The image files
themselves cannot travel to the GPU box, so benchmarks and parity tests use these
closed-form, integer-only generators with the same shapes and the same role
(layer 2 is a 32-px checker like ``check.png``).  Every consumer -- the oracle
harness, the golden generator, the tests and bench.py -- calls the same functions,
so inputs are bit-identical everywhere.
"""
from __future__ import annotations

import numpy as np


def _hash2(x: np.ndarray, y: np.ndarray, seed: int) -> np.ndarray:
    """32-bit integer hash of a lattice point (uint32 wrap-around arithmetic)."""
    h = (x.astype(np.uint64) * 73856093) ^ (y.astype(np.uint64) * 19349663) ^ np.uint64(seed * 83492791)
    h &= np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    return h.astype(np.uint32)


def texture_layers(size: int = 256) -> np.ndarray:
    """uint8 [4, size, size, 4] (layer, row, column, RGBA), A = 255."""
    y, x = np.meshgrid(np.arange(size, dtype=np.uint32), np.arange(size, dtype=np.uint32), indexing="ij")
    out = np.empty((4, size, size, 4), np.uint8)
    out[..., 3] = 255
    # layer 0: "cobblestone" -- 16-px cells with per-cell grey level and dark joints
    cell = _hash2(x // 16, y // 16, 1)
    joint = ((x % 16) < 2) | ((y % 16) < 2)
    g = 90 + (cell % 96)
    g = np.where(joint, 40, g)
    out[0, ..., 0] = g
    out[0, ..., 1] = g
    out[0, ..., 2] = (g * 7) // 8
    # layer 1: "sand" -- per-texel grain on a warm base
    n = _hash2(x, y, 2) % 48
    out[1, ..., 0] = 190 + n
    out[1, ..., 1] = 160 + n
    out[1, ..., 2] = 100 + n // 2
    # layer 2: "check" -- 32-px black / white checker (the floor of render.map samples this one)
    c = (((x // 32) + (y // 32)) & 1).astype(np.uint32)
    v = np.where(c == 1, 235, 20)
    out[2, ..., 0] = v
    out[2, ..., 1] = v
    out[2, ..., 2] = v
    # layer 3: "grass" -- green grain with 4-px blades
    n = _hash2(x // 2, y // 4, 3) % 80
    out[3, ..., 0] = 20 + n // 4
    out[3, ..., 1] = 110 + n
    out[3, ..., 2] = 25 + n // 3
    return out


def skybox_cross(width: int = 4096) -> np.ndarray:
    """uint8 [1, 3*width/4, width, 4]: horizontal-cross cube map, face = width/4.

    Face offsets are the ones map_to_cube expects (primitives.cl:33-101): in texel
    rows counted from the bottom, +X (2f,1f), -X (0,1f), +Y (1f,2f), -Y (1f,0),
    +Z (1f,1f), -Z (3f,1f).  Each face is a smooth two-axis gradient plus a coarse
    grid so neighbouring texels differ (index errors show up) while a one-texel
    shift changes a channel by at most a few counts.  Unused corners are mid-grey.
    """
    if width % 4:
        raise ValueError("skybox width must be a multiple of 4")
    f = width // 4
    height = 3 * f
    img = np.empty((1, height, width, 4), np.uint8)
    img[..., :3] = 128
    img[..., 3] = 255
    v, u = np.meshgrid(np.arange(f, dtype=np.uint32), np.arange(f, dtype=np.uint32), indexing="ij")
    a = (u * 255) // max(f - 1, 1)
    b = (v * 255) // max(f - 1, 1)
    grid = np.where(((u * 8 // f) + (v * 8 // f)) & 1, 24, 0).astype(np.uint32)
    faces = {  # (column offset, row offset measured from the TOP of the stored image)
        "+x": (2 * f, f), "-x": (0, f), "+y": (f, 0), "-y": (f, 2 * f), "+z": (f, f), "-z": (3 * f, f),
    }
    tint = {"+x": (200, 60, 40), "-x": (40, 200, 60), "+y": (70, 120, 230),
            "-y": (90, 70, 50), "+z": (220, 200, 90), "-z": (160, 70, 200)}
    for name, (cx, cy) in faces.items():
        t = tint[name]
        r = (t[0] * (255 + a) // 510 + grid).clip(0, 255)
        g = (t[1] * (255 + b) // 510 + grid).clip(0, 255)
        bl = (t[2] * (510 - a - b + 255) // 765).clip(0, 255)
        img[0, cy:cy + f, cx:cx + f, 0] = r
        img[0, cy:cy + f, cx:cx + f, 1] = g
        img[0, cy:cy + f, cx:cx + f, 2] = bl
    return img
