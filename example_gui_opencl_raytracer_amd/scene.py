"""Scene data in the reference's wire format.

Layouts follow the reference's device structs (src/cl/types.cl:4-59, host mirrors
src/cpu_obj.h:10-48): ``rmaterial`` 64 B, ``rsphere`` 96 B, ``rplane`` 96 B,
``rlight`` 48 B, float3 = 16 B.  The scene archive (``render.map``) is the raw
struct images behind one count byte each (src/cpu_obj.c:51-101).

This module only builds / parses bytes; it does no tracing.
"""
from __future__ import annotations

import numpy as np

# ---------------------------------------------------------------- wire dtypes
MATERIAL = np.dtype(
    {
        "names": ["rgb", "ambient", "diffuse", "specular", "shininess", "transperent",
                  "dielectric", "n", "reflectivity", "texture_id", "texture_scale"],
        "formats": [("<f4", 3), "<f4", "<f4", "<f4", "<u4", "<u4", "<u4", "<f4", "<f4", "<i4", "<f4"],
        "offsets": [0, 16, 20, 24, 28, 32, 36, 40, 44, 48, 52],
        "itemsize": 64,
    }
)
SPHERE = np.dtype(
    {"names": ["origin", "radius", "material"], "formats": [("<f4", 3), "<f4", MATERIAL],
     "offsets": [0, 16, 32], "itemsize": 96}
)
PLANE = np.dtype(
    {"names": ["normal", "point_in_plane", "material"], "formats": [("<f4", 3), ("<f4", 3), MATERIAL],
     "offsets": [0, 16, 32], "itemsize": 96}
)
LIGHT = np.dtype(
    {"names": ["origin", "radius", "intensity", "rgb"], "formats": [("<f4", 3), "<f4", "<f4", ("<f4", 3)],
     "offsets": [0, 16, 20, 32], "itemsize": 48}
)
RAY = np.dtype(
    {"names": ["origin", "dir", "rgb", "depth"], "formats": [("<f4", 3), ("<f4", 3), ("<f4", 3), "<i4"],
     "offsets": [0, 16, 32, 48], "itemsize": 64}
)


def _material(rgb, ambient, diffuse, specular, shininess, transperent, dielectric, n, reflectivity,
              texture_id=-1, texture_scale=0.0):
    m = np.zeros((), MATERIAL)
    m["rgb"] = rgb
    m["ambient"], m["diffuse"], m["specular"] = ambient, diffuse, specular
    m["shininess"] = shininess
    m["transperent"], m["dielectric"] = int(transperent), int(dielectric)
    m["n"], m["reflectivity"] = n, reflectivity
    m["texture_id"], m["texture_scale"] = texture_id, texture_scale
    return m


# Material presets: values of src/cpu_obj.c:6-49.
def stone():
    return _material((1, 1, 1), 0.4, 0.2, 0.6, 50, False, True, 1.57, 0.0, 0, 0.0)


def plastic():
    return _material((1, 1, 1), 0.3, 0.2, 0.6, 50, False, False, 1.4, 0.1, 0, 0.0)


def mirror():
    return _material((0.2, 0.2, 0.2), 0.3, 0.0, 0.6, 100, False, True, 1.0, 1.0, 0, 0.0)


def glass():
    return _material((0, 0, 0), 0.1, 0.0, 0.0, 20, True, True, 1.52, 0.04, 0, 0.0)


def _copy_fields(dst, src):
    for name in dst.dtype.names:
        if dst.dtype[name].names:
            _copy_fields(dst[name], src[name])
        else:
            dst[name] = src[name]


def _canonical(a, dt):
    """Copy `a` into a fresh record array of dtype `dt` whose struct padding is zero (numpy
    leaves padding bytes undefined when it copies structured arrays)."""
    a = np.asarray(a, dtype=dt).reshape(-1)
    out = np.zeros(len(a), dt)
    _copy_fields(out, a)
    return out


class Scene:
    """Spheres / planes / lights as numpy record arrays in wire layout (padding bytes zero)."""

    def __init__(self, spheres, planes, lights):
        self.spheres = _canonical(spheres, SPHERE)
        self.planes = _canonical(planes, PLANE)
        self.lights = _canonical(lights, LIGHT)

    @property
    def counts(self):
        return len(self.spheres), len(self.planes), len(self.lights)

    # ---- render.map (src/cpu_obj.c:51-101): u8 count + raw structs, three times
    def to_bytes(self) -> bytes:
        ns, np_, nl = self.counts
        if max(ns, np_, nl) > 255:
            raise ValueError("render.map stores one-byte counts (src/cpu_obj.c:62-68); use to_bytes_ext")
        return (bytes([ns]) + self.spheres.tobytes() + bytes([np_]) + self.planes.tobytes()
                + bytes([nl]) + self.lights.tobytes())

    @staticmethod
    def from_bytes(buf: bytes) -> "Scene":
        if buf[:8] == Scene.EXT_MAGIC:
            if len(buf) < 20:
                raise ValueError("truncated scene archive")
            counts = np.frombuffer(buf, "<u4", 3, 8)
            off, out = 20, []
            for n, dt in zip(counts, (SPHERE, PLANE, LIGHT)):
                end = off + int(n) * dt.itemsize
                if end > len(buf):
                    raise ValueError("truncated scene archive")
                out.append(np.frombuffer(buf, dtype=dt, count=int(n), offset=off).copy())
                off = end
            return Scene(*out)
        off = 0
        out = []
        for dt in (SPHERE, PLANE, LIGHT):
            if off >= len(buf):
                raise ValueError("truncated scene archive")
            n = buf[off]
            off += 1
            end = off + n * dt.itemsize
            if end > len(buf):
                raise ValueError("truncated scene archive")
            out.append(np.frombuffer(buf, dtype=dt, count=n, offset=off).copy())
            off = end
        return Scene(*out)

    # ---- extended archive: the one-byte counts cap a scene at 255 primitives per
    #      type.  Layout: 8-byte magic, three little-endian u32 counts, then the same raw struct arrays.
    #      Legacy files (first byte = sphere count) remain valid input everywhere.
    EXT_MAGIC = b"\xffRMAPv2\x00"

    def to_bytes_ext(self) -> bytes:
        ns, np_, nl = self.counts
        return (self.EXT_MAGIC + np.array([ns, np_, nl], "<u4").tobytes() + self.spheres.tobytes()
                + self.planes.tobytes() + self.lights.tobytes())

    def save(self, path, ext: bool | None = None):
        """ext=None: legacy format when the counts fit one byte, extended otherwise."""
        if ext is None:
            ext = max(self.counts) > 255
        with open(path, "wb") as f:
            f.write(self.to_bytes_ext() if ext else self.to_bytes())

    @staticmethod
    def load(path) -> "Scene":
        with open(path, "rb") as f:
            return Scene.from_bytes(f.read())


def render_map_scene() -> Scene:
    """The demo scene, regenerated from the literal values of scene_dump.c:8-71
    (4 spheres, 2 planes, 3 lights); struct padding is zero here, whereas the
    committed scenes/render.map carries uninitialised stack bytes in the pads."""
    s = np.zeros(4, SPHERE)
    s[0]["origin"], s[0]["radius"], s[0]["material"] = (4.5, 0.5, -1.0), 0.5, plastic()
    s[0]["material"]["rgb"] = (1, 0, 0)
    s[1]["origin"], s[1]["radius"], s[1]["material"] = (-1.0, 1.0, 4.5), 0.8, plastic()
    s[1]["material"]["rgb"] = (0, 0, 1)
    s[2]["origin"], s[2]["radius"], s[2]["material"] = (0.8, 0.8, 1.5), 0.8, glass()
    s[3]["origin"], s[3]["radius"], s[3]["material"] = (-0.6, 0.8, -1.0), 0.8, glass()
    s[3]["material"]["rgb"] = (0, 1, 0)
    s[3]["material"]["ambient"] = 0.05
    for k in range(4):
        s[k]["material"]["texture_id"] = -1

    p = np.zeros(2, PLANE)
    p[0]["point_in_plane"], p[0]["normal"], p[0]["material"] = (0, 0, 0), (0, 1, 0), stone()
    p[0]["material"]["rgb"] = (0, 0, 0)
    p[0]["material"]["texture_scale"] = 100.0
    p[0]["material"]["texture_id"] = 2
    p[1]["point_in_plane"], p[1]["normal"], p[1]["material"] = (0, 0, 7), (0, 0, -1), mirror()
    p[1]["material"]["ambient"] = 0.3
    p[1]["material"]["shininess"] = 150
    p[1]["material"]["specular"] = 0.4
    p[1]["material"]["rgb"] = (0.3, 0.3, 0.3)
    p[1]["material"]["texture_id"] = -1

    l = np.zeros(3, LIGHT)
    l[0]["origin"], l[0]["intensity"], l[0]["radius"], l[0]["rgb"] = (-2, 3, 2), 8.0, 0.1, (0, 1, 0)
    l[1]["origin"], l[1]["intensity"], l[1]["radius"], l[1]["rgb"] = (2, 1.5, 0.2), 50.3, 0.1, (1, 1, 1)
    l[2]["origin"], l[2]["intensity"], l[2]["radius"], l[2]["rgb"] = (1, 4, 3), 20.5, 0.1, (0, 0, 1)
    return Scene(s, p, l)


def _floor_and_lights():
    base = render_map_scene()
    return base.planes[:1].copy(), base.lights.copy()


def dielectric_field_scene(grid: int = 8) -> Scene:
    """Config C3: grid x grid glass spheres r=0.45, pitch 1.0, y=0.5,
    over the textured floor with render.map's three lights.  Divergence stress."""
    s = np.zeros(grid * grid, SPHERE)
    k = 0
    for iz in range(grid):
        for ix in range(grid):
            s[k]["origin"] = (float(ix), 0.5, float(iz))
            s[k]["radius"] = 0.45
            s[k]["material"] = glass()
            s[k]["material"]["texture_id"] = -1
            k += 1
    planes, lights = _floor_and_lights()
    return Scene(s, planes, lights)


def sphere_grid_scene(nx: int = 100, nz: int = 100) -> Scene:
    """Config C4: nx*nz opaque plastic spheres r=0.3, pitch 1.0, colours
    from an integer hash; floor plane; three lights.  Needs wide counts."""
    n = nx * nz
    s = np.zeros(n, SPHERE)
    idx = np.arange(n, dtype=np.uint32)
    h = (idx * np.uint32(2654435761)) & np.uint32(0xFFFFFFFF)
    s["origin"][:, 0] = (idx % nx).astype(np.float32) - np.float32(nx / 2)
    s["origin"][:, 1] = 0.3
    s["origin"][:, 2] = (idx // nx).astype(np.float32)
    s["radius"] = 0.3
    m = plastic()
    for name in MATERIAL.names:
        s["material"][name] = m[name]
    s["material"]["rgb"][:, 0] = ((h >> 0) & 255).astype(np.float32) / np.float32(255)
    s["material"]["rgb"][:, 1] = ((h >> 8) & 255).astype(np.float32) / np.float32(255)
    s["material"]["rgb"][:, 2] = ((h >> 16) & 255).astype(np.float32) / np.float32(255)
    s["material"]["texture_id"] = -1
    planes, lights = _floor_and_lights()
    return Scene(s, planes, lights)
