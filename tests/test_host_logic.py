"""Host-side logic that needs no GPU: scene wire format, strip partition, camera, PNG codec,
prepared geometry, and that the C-ABI library loads and exports every declared symbol."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, CAM, REFERENCE_ROOT, ROOT
from example_gui_opencl_raytracer_amd import api, scene as S, textures as T
from example_gui_opencl_raytracer_amd.renderer import strip_rows


# ------------------------------------------------------------------ scene archive (render.map)
def test_render_map_layout_and_roundtrip(demo_scene, tmp_path):
    assert (S.MATERIAL.itemsize, S.SPHERE.itemsize, S.PLANE.itemsize, S.LIGHT.itemsize, S.RAY.itemsize) == (64, 96, 96, 48, 64)
    blob = demo_scene.to_bytes()
    assert len(blob) == 1 + 4 * 96 + 1 + 2 * 96 + 1 + 3 * 48 == 723
    assert (blob[0], blob[1 + 384], blob[1 + 384 + 1 + 192]) == (4, 2, 3)
    back = S.Scene.from_bytes(blob)
    assert back.to_bytes() == blob
    p = tmp_path / "render.map"
    demo_scene.save(p)
    assert S.Scene.load(p).to_bytes() == blob


def test_truncated_archive_is_rejected(demo_scene):
    blob = demo_scene.to_bytes()
    for cut in (0, 1, 100, 385, 722):
        with pytest.raises(ValueError):
            S.Scene.from_bytes(blob[:cut])


def test_one_byte_counts_limit():
    big = S.sphere_grid_scene(20, 20)            # 400 spheres > 255
    with pytest.raises(ValueError):
        big.to_bytes()


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="reference tree absent")
def test_regenerated_scene_equals_committed_render_map(demo_scene):
    """Field by field (the committed file has uninitialised bytes in the struct padding)."""
    ref = S.Scene.load(os.path.join(REFERENCE_ROOT, "scenes", "render.map"))
    assert ref.counts == demo_scene.counts == (4, 2, 3)

    def fields(a):
        out = []
        for name in a.dtype.names:
            v = a[name]
            out += fields(v) if v.dtype.names else [np.ascontiguousarray(v).reshape(len(a), -1).view(np.uint32)]
        return out
    for mine, theirs in ((demo_scene.spheres, ref.spheres), (demo_scene.planes, ref.planes), (demo_scene.lights, ref.lights)):
        assert np.array_equal(np.concatenate(fields(mine), 1), np.concatenate(fields(theirs), 1))


def test_extended_archive_roundtrip(demo_scene, tmp_path):
    big = S.sphere_grid_scene(20, 20)
    blob = big.to_bytes_ext()
    assert blob[:8] == S.Scene.EXT_MAGIC and len(blob) == 20 + 400 * 96 + 96 + 3 * 48
    assert S.Scene.from_bytes(blob).to_bytes_ext() == blob
    p = tmp_path / "big.map"
    big.save(p)                                   # picks the extended format on its own
    assert S.Scene.load(p).counts == (400, 1, 3)
    small = S.Scene.from_bytes(demo_scene.to_bytes_ext())      # small scenes may use it too
    assert small.to_bytes() == demo_scene.to_bytes()
    with pytest.raises(ValueError):
        S.Scene.from_bytes(blob[:1000])


def test_benchmark_scene_generators():
    c3 = S.dielectric_field_scene(8)
    assert c3.counts == (64, 1, 3) and (c3.spheres["material"]["transperent"] == 1).all()
    c4 = S.sphere_grid_scene(100, 100)
    assert c4.counts == (10000, 1, 3) and c4.spheres.nbytes == 960000        # > 160 KB LDS
    assert (c4.spheres["material"]["transperent"] == 0).all()


# ------------------------------------------------------------------ strips
@pytest.mark.parametrize("height", [1, 7, 8, 9, 480, 600, 1080, 4096, 8192])
@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_strip_partition(height, world):
    parts = [strip_rows(height, world, r) for r in range(world)]
    assert parts[0][0] == 0 and sum(n for _, n in parts) == height
    for (a0, an), (b0, _) in zip(parts, parts[1:]):
        assert a0 + an == b0                                  # contiguous, ordered, no overlap
    for r0, n in parts:
        assert r0 % 8 == 0 or n == 0                          # tile-aligned starts
    if height >= 8 * world:
        ns = [n for _, n in parts]
        assert max(ns) - min(ns) <= 8                         # balanced to within one tile row


# ------------------------------------------------------------------ camera (rgen_perspective)
@pytest.mark.parametrize("cam", [
    dict(origin=(0.8, 2.5, -8.0), look=(0.2, 0.0, 1.0), fov=90.0, focal=1.0, w=800, h=600),
    dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=60.0, focal=1.0, w=4096, h=4096),
    dict(origin=(0.0, 1.0, 0.0), look=(1.0, 0.3, -0.2), fov=120.0, focal=2.5, w=1920, h=1080),
])
def test_camera_matches_oracle_bit_for_bit(oracle, cam):
    mine = api.perspective(cam["origin"], cam["look"], cam["fov"], cam["focal"], cam["w"], cam["h"])
    ref = oracle.camera(cam["origin"], cam["look"], cam["fov"], cam["focal"], cam["w"], cam["h"])
    assert bytes(mine) == bytes(ref)


def test_camera_matches_the_references_own_rgen_perspective():
    """tests/golden/camera.npz holds what the reference's OWN src/cpu_ray.c (rinit_camera + rgen_perspective,
    cpu_ray.c:24-35, 42-106; compiled into oracle/_ref/libref_cpu_ray.so by oracle/gen_golden.py) returns for twelve
    cameras: the product's clw_host_perspective must give the same bytes (H1 pinned against the reference itself)."""
    g = np.load(os.path.join(GOLDEN, "camera.npz"))
    assert len(g["inputs"]) >= 5
    for row, want in zip(g["inputs"], g["outputs"]):
        o, l, fov, focal, w, h = row[0:3], row[3:6], float(row[6]), float(row[7]), int(row[8]), int(row[9])
        cam = api.perspective(tuple(o), tuple(l), fov, focal, w, h)
        got = np.array(list(cam.im_corner) + list(cam.origin) + list(cam.up) + list(cam.right) + [cam.w_factor, cam.h_factor], np.float32)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (row, got, want)


def test_camera_rejections():
    for bad in (dict(look=(0, 1, 0), fov=90.0), dict(look=(0, 0, 1), fov=180.0), dict(look=(0, 0, 1), fov=0.0)):
        with pytest.raises(ValueError):
            api.perspective((0, 0, 0), bad["look"], bad["fov"], 1.0, 64, 64)     # cpu_ray.c:58-63


# ------------------------------------------------------------------ PNG codec
def test_png_roundtrip(tmp_path):
    rng = np.random.default_rng(1)
    for (w, h) in ((1, 1), (7, 5), (256, 256), (300, 513)):
        img = rng.integers(0, 2**24, w * h, dtype=np.uint32)
        p = str(tmp_path / f"t_{w}x{h}.png")
        api.write_png(p, img, w, h)
        back = api.read_png(p)
        assert back.shape == (h, w, 4) and (back[..., 3] == 255).all()
        packed = (back[..., 0].astype(np.uint32) << 16) | (back[..., 1].astype(np.uint32) << 8) | back[..., 2]
        assert np.array_equal(packed.reshape(-1), img)
    from PIL import Image
    pil = np.asarray(Image.open(p).convert("RGB"))            # an independent decoder agrees
    assert np.array_equal(pil, back[..., :3])


def test_png_writer_parallel_path_for_big_frames(tmp_path):
    """Frames of >= 4 Mpixel (config C5's 8192 x 8192 output) are deflated band by band on all cores and stitched into
    one zlib stream (png_codec.c): the file must decode to the same pixels with this library's reader AND with an
    independent decoder, also when the last band is short."""
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    rng = np.random.default_rng(7)
    for (w, h) in ((2048, 2048), (4096, 1024 + 77)):
        y, x = np.mgrid[0:h, 0:w]
        img = (((x * 255 // w).astype(np.uint32) << 16) | ((y * 255 // h).astype(np.uint32) << 8)
               | (rng.integers(0, 256, (h, w), dtype=np.uint32) * (x % 64 < 8))).reshape(-1).astype(np.uint32)
        p = str(tmp_path / f"big_{w}x{h}.png")
        api.write_png(p, img, w, h)
        back = api.read_png(p)
        packed = (back[..., 0].astype(np.uint32) << 16) | (back[..., 1].astype(np.uint32) << 8) | back[..., 2]
        assert np.array_equal(packed.reshape(-1), img)
        assert np.array_equal(np.asarray(Image.open(p).convert("RGB")), back[..., :3])


def test_png_reader_accepts_filtered_files_and_rejects_non_rgb8(tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    arr = (np.add.outer(np.arange(64), np.arange(96))[..., None] * np.array([1, 2, 3]) + rng.integers(0, 8, (64, 96, 3))).astype(np.uint8)
    p = str(tmp_path / "pil.png")
    Image.fromarray(arr, "RGB").save(p, optimize=True)        # PIL picks adaptive filters (sub/up/avg/paeth)
    assert np.array_equal(api.read_png(p)[..., :3], arr)
    Image.fromarray(arr[..., 0], "L").save(str(tmp_path / "grey.png"))
    with pytest.raises(ValueError, match=r"\(3\)"):           # WPNG_ERR_FORMAT: "must have a depth of 8 bits and be RGB"
        api.read_png(str(tmp_path / "grey.png"))
    (tmp_path / "junk.png").write_bytes(b"not a png at all")
    with pytest.raises(ValueError, match=r"\(2\)"):
        api.read_png(str(tmp_path / "junk.png"))
    with pytest.raises(ValueError, match=r"\(1\)"):
        api.read_png(str(tmp_path / "missing.png"))


def test_png_reader_rejects_sizes_that_overflow(tmp_path):
    """A crafted IHDR whose width x height products wrap size_t must be refused before anything is allocated."""
    import struct
    import zlib

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data))
    for (w, h) in ((0xFFFFFFFF, 0xFFFFFFFF), (0x80000000, 1), (0x7FFFFFFF, 0x7FFFFFFF), (1, 0x80000001), (0x40000000, 0x40000000)):
        ihdr = struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b"")
        p = tmp_path / f"big_{w}_{h}.png"
        p.write_bytes(png)
        with pytest.raises(ValueError):
            api.read_png(str(p))
    # a chunk length beyond 2^31 (it would be truncated by the (long) cast of the skip) is refused as well
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 2, 0, 0, 0)) + struct.pack(">I", 0xFFFFFFF0) + b"tEXt"
    (tmp_path / "len.png").write_bytes(png)
    with pytest.raises(ValueError):
        api.read_png(str(tmp_path / "len.png"))


@pytest.mark.skipif(not os.path.isdir(REFERENCE_ROOT), reason="reference tree absent")
def test_png_reader_on_the_reference_assets():
    from PIL import Image
    for name in ("check.png", "cobblestone.png", "sand.png", "grass.png"):
        path = os.path.join(REFERENCE_ROOT, "assets", name)
        mine = api.read_png(path)
        assert mine.shape == (256, 256, 4)
        assert np.array_equal(mine[..., :3], np.asarray(Image.open(path).convert("RGB")))


# ------------------------------------------------------------------ procedural inputs
def test_procedural_textures_are_deterministic_and_shaped_like_the_assets():
    t = T.texture_layers()
    assert t.shape == (4, 256, 256, 4) and t.dtype == np.uint8 and (t[..., 3] == 255).all()
    assert np.array_equal(t, T.texture_layers())
    assert set(np.unique(t[2, ..., 0])) == {20, 235}          # layer 2 is the checker
    s = T.skybox_cross(512)
    assert s.shape == (1, 384, 512, 4)
    import zlib
    assert zlib.crc32(t.tobytes()) == 0x32d4f5ae and zlib.crc32(s.tobytes()) == 0xef0355fd


# ------------------------------------------------------------------ prepared geometry
def test_prepared_geometry_stream(demo_scene):
    L = api.load_library()
    L.wprep_geom_f4.restype = C.c_size_t
    L.wprep_geom_f4.argtypes = [C.c_uint32] * 3
    ns, npl, nl = demo_scene.counts
    n4 = L.wprep_geom_f4(ns, npl, nl)
    assert n4 == ns + 2 * npl + 2 * nl
    geom = np.zeros((n4, 4), np.float32)
    ptex = np.zeros((2 * npl, 4), np.float32)
    L.wprep_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.wprep_build(demo_scene.spheres.ctypes.data, ns, demo_scene.planes.ctypes.data, npl, demo_scene.lights.ctypes.data, nl,
                  geom.ctypes.data, ptex.ctypes.data)
    r = demo_scene.spheres["radius"]
    assert np.array_equal(geom[:ns, :3], demo_scene.spheres["origin"])
    assert np.array_equal(np.abs(geom[:ns, 3]), r * r)
    assert np.array_equal(np.signbit(geom[:ns, 3]), demo_scene.spheres["material"]["transperent"] != 0)   # glass spheres flagged
    assert np.array_equal(geom[ns:ns + 2 * npl:2, :3], demo_scene.planes["normal"])
    assert list(geom[ns:ns + 2 * npl:2, 3]) == [1.0, 0.0]                         # only the floor is textured
    lg = geom[ns + 2 * npl:]
    li = demo_scene.lights
    assert np.array_equal(lg[0::2, :3], li["origin"]) and np.array_equal(lg[0::2, 3], li["radius"] * li["radius"])
    want = (li["rgb"] * li["intensity"][:, None]) * np.float32(0.31830988618379067154)
    assert np.array_equal(lg[1::2, :3], want.astype(np.float32)) and np.array_equal(lg[1::2, 3], li["radius"])
    # floor n = (0,1,0): first qualifying axis is X -> b0 = X x n = (0,0,1), b1 = n x b0 = (1,0,0)
    assert list(ptex[0, :3]) == [0.0, 0.0, 1.0] and ptex[0, 3] == 100.0 and list(ptex[1, :3]) == [1.0, 0.0, 0.0]
    assert ptex[1, 3:].view(np.int32)[0] == 2 and ptex[3, 3:].view(np.int32)[0] == -1


# ------------------------------------------------------------------ the C-ABI library
def test_library_exports_every_declared_symbol():
    L = api.load_library()
    declared = set()
    for hdr in ("opencl_wrap.h", "hip_wrap_ext.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b((?:cl_wrap|clw_ext|clw_host)_\w+)\s*\(", text))
    assert len(declared) >= 29 and declared == set(api.SYMBOLS)
    for name in sorted(declared):
        assert hasattr(L, name), f"libopencl_wrap_hip.so does not export {name}"
    assert b"gfx950" in L.clw_ext_version()


def test_struct_mirror_matches_header():
    assert C.sizeof(api.cl_wrap) == 8 + 4 + 4 * 16 + 4 * 16 * 32 + 4 + 8 * 16 * 32   # impl, num, nums, ids, (pad), handles
    assert api.cl_wrap.buffers.offset % 8 == 0
