"""The reference's only committed OUTPUT, out/scene.png (800x600, depth 15, rendered by its author on an unknown OpenCL GPU with
raypng.c), as a known-answer test.  tests/golden/reference_scene/ holds that image and the inputs raypng.c reads for it
(raypng.c:28-29, 74-81): scenes/render.map and the five asset PNGs -- data files of the reference, inputs and expected output.

Finding (a one-parameter search, kept as a test below): with the kernels as they are in the reference today 88.5 % of the image's
pixels are reproduced bit for bit and 6 % are more than 1 LSB off, ALL of the systematic difference lying in the floor region
shadowed by the two glass spheres; with the transparent-shadow factor TRANSPERENT_THROUGH (primitives.cl:7, :419) at 1.0 instead
of 0.8 -- a version of the kernels in which glass does not attenuate shadow rays -- 99.28 % are reproduced bit for bit and
99.62 % within 1 LSB: the committed image predates that constant.  Everything else about the path (camera, ray generation,
intersection, Phong terms, the xorshift soft-shadow sampling, reflection / refraction to depth 15, texture and skybox lookup,
the truncating pack) is pinned by this image at that level, on a GPU and compiler nobody here has seen.

The CPU test runs the oracle; tests/test_gpu_api.py::test_unchanged_raypng_driver_reproduces_the_references_committed_render
runs the reference's unchanged raypng.c driver on the HIP path against the same image."""
import os

import numpy as np
import pytest

from conftest import CAM, GOLDEN, channel_diff, frame_mask

FIX = os.path.join(GOLDEN, "reference_scene")
KEY = "scene_png_800x600_d15"
W, H, DEPTH = 800, 600, 15        # raypng.c:8-9, raytracing.cl:9


def load_fixture():
    """-> (scene, texture layers [4,h,w,4], skybox [1,h,w,4], committed frame as packed 0x00RRGGBB, discontinuity mask)"""
    from example_gui_opencl_raytracer_amd import api
    from example_gui_opencl_raytracer_amd.scene import Scene
    tex = np.stack([api.read_png(os.path.join(FIX, n + ".png")) for n in ("cobblestone", "sand", "check", "grass")])   # raypng.c:74-78
    sky = api.read_png(os.path.join(FIX, "stormydays.png"))[None]                                                      # raypng.c:80-81
    img = api.read_png(os.path.join(FIX, "scene.png"))
    assert img.shape == (H, W, 4)
    want = (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)
    gm = dict(np.load(os.path.join(GOLDEN, "masks_scenes.npz")))
    return Scene.load(os.path.join(FIX, "render.map")), tex, sky, want, frame_mask(gm, KEY, W * H), gm


def fixture_stats(got, want, mask):
    d = channel_diff(got, want)
    return dict(exact=float((d == 0).mean()), within_1lsb=float((d <= 1).mean()), outliers=int((d > 1).sum()),
                outliers_off_mask=int(((d > 1) & ~mask).sum()), max_diff=int(d.max()), mask=float(mask.mean()))


# the bar both the oracle and the HIP path are held to against the committed image (measured for the oracle: 0.99280 / 0.99621,
# 1 821 pixels more than 1 LSB off, 838 of them off the discontinuity mask -- two thin bands on the floor)
BAR = dict(exact=0.990, within_1lsb=0.995, outliers_off_mask=1000)


def check_fixture_bar(st, what):
    assert st["exact"] >= BAR["exact"] and st["within_1lsb"] >= BAR["within_1lsb"], (what, st)
    assert st["outliers_off_mask"] <= BAR["outliers_off_mask"], (what, st)


def test_fixture_files_are_the_references_own():
    """Where the reference tree is present: the committed copies are byte-identical to the reference's files."""
    from conftest import REFERENCE_ROOT
    if not os.path.isdir(REFERENCE_ROOT):
        pytest.skip("reference tree absent")
    pairs = {"scene.png": "out/scene.png", "render.map": "scenes/render.map", "stormydays.png": "assets/bg/stormydays.png"}
    pairs.update({n + ".png": f"assets/{n}.png" for n in ("cobblestone", "sand", "check", "grass")})
    for mine, theirs in pairs.items():
        assert open(os.path.join(FIX, mine), "rb").read() == open(os.path.join(REFERENCE_ROOT, theirs), "rb").read(), mine


def test_oracle_reproduces_the_references_committed_render(oracle):
    scene, tex, sky, want, mask, gm = load_fixture()
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, W, H)
    try:
        oracle.set_transparent_through(1.0)
        got, _, cnt = oracle.render(cam, scene, tex, sky, DEPTH)
    finally:
        oracle.set_transparent_through(0.8)
    import zlib
    assert zlib.crc32(got.tobytes()) == int(gm[KEY + "_crc32"][0])        # the frame the mask was built for
    assert cnt.oob_reads == 0 and cnt.int_cast_oor == 0
    st = fixture_stats(got, want, mask)
    print("oracle (factor 1.0) vs out/scene.png:", st)
    check_fixture_bar(st, "oracle")
    # ... and with today's constant: the glass shadows differ, nothing else (the finding above, kept checkable)
    now, _, _ = oracle.render(cam, scene, tex, sky, DEPTH)
    st08 = fixture_stats(now, want, mask)
    print("oracle (factor 0.8) vs out/scene.png:", st08)
    assert 0.87 < st08["exact"] < 0.90 and st08["within_1lsb"] < 0.95
    differs = (now != got)
    d = channel_diff(now, want)
    assert ((d > 1) & ~differs & ~mask).sum() <= BAR["outliers_off_mask"]   # outside the glass shadows the two agree with the image alike
