"""Pins the CPU restatement against the reference's OWN kernel sources compiled for the host
(oracle/_ref/libref_cl.so, built in place from /root/reference by oracle/Makefile).
Runs wherever that binary exists; everywhere else the committed fixtures carry the pin
(test_oracle_golden.py).  Bar: bit-exact."""
import os

import numpy as np
import pytest

from conftest import CAM, REFERENCE_ROOT, channel_diff


@pytest.mark.parametrize("w,h,depth", [(96, 64, 1), (96, 64, 2), (96, 64, 3), (200, 150, 4), (96, 64, 8), (200, 150, 15)])
def test_frames_bit_exact_against_reference_kernels(oracle, reference, demo_scene, tex, sky, w, h, depth):
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    want, oob = reference.render(cam, demo_scene, tex, sky, depth)
    got, _, cnt = oracle.render(cam, demo_scene, tex, sky, depth)
    assert oob == 0 and cnt.oob_reads == 0
    assert np.array_equal(got, want)


@pytest.mark.parametrize("origin,look,fov", [((0.8, 2.5, -8.0), (0.0, 0.0, 1.0), 90.0),      # rayinteractive.c:111-115
                                             ((-3.0, 0.6, 0.5), (1.0, 0.05, 0.3), 70.0),       # low, grazing the floor
                                             ((0.8, 0.8, 1.5), (0.3, -0.2, 1.0), 100.0),       # camera INSIDE glass sphere #2
                                             ((1.0, 9.0, 1.0), (0.01, -1.0, 0.02), 60.0)])    # looking down
def test_other_cameras(oracle, reference, demo_scene, tex, sky, origin, look, fov):
    cam = oracle.camera(origin, look, fov, 1.0, 128, 96)
    want, _ = reference.render(cam, demo_scene, tex, sky, 15)
    got, _, _ = oracle.render(cam, demo_scene, tex, sky, 15)
    assert np.array_equal(got, want)


def test_glass_field_scene(oracle, reference, tex, sky):
    """Config C3's scene (64 dielectric spheres): deep refraction trees, total internal reflection."""
    from example_gui_opencl_raytracer_amd import scene
    sc = scene.dielectric_field_scene(8)
    cam = oracle.camera((3.5, 3.0, -6.0), (0.0, -2.5, 9.5), 90.0, 1.0, 96, 96)
    want, _ = reference.render(cam, sc, tex, sky, 8)
    got, _, cnt = oracle.render(cam, sc, tex, sky, 8)
    assert np.array_equal(got, want)
    assert cnt.pushes > 1000 and cnt.max_stack >= 4


def test_textures_off_and_all_layers(oracle, reference, demo_scene, tex, sky):
    from example_gui_opencl_raytracer_amd.scene import Scene
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 96, 64)
    for tid in (-1, 0, 1, 3):
        planes = demo_scene.planes.copy()
        planes["material"]["texture_id"][0] = tid
        sc = Scene(demo_scene.spheres, planes, demo_scene.lights)
        want, _ = reference.render(cam, sc, tex, sky, 4)
        got, _, cnt = oracle.render(cam, sc, tex, sky, 4)
        assert np.array_equal(got, want)
        assert (cnt.texel_fetches == 0) == (tid < 0)


def test_committed_fixtures_are_what_the_reference_produces_now(oracle, reference, demo_scene, tex, sky, golden_masks):
    """Regenerates, from the reference builds in oracle/_ref, the camera fixtures and one mask set, and compares them
    with the committed files (so the fixtures cannot drift from oracle/gen_golden.py)."""
    import os
    from conftest import GOLDEN
    from oracle import gen_golden as G
    from oracle.oracle_py import REF_FMA_SO, RefCamera, Reference
    if not (RefCamera.available() and os.path.exists(REF_FMA_SO)):
        pytest.skip("oracle/_ref camera / fma builds absent")
    g = np.load(os.path.join(GOLDEN, "camera.npz"))
    rc = RefCamera()
    for (o, l, fov, focal, w, h), want in zip(G.CAMERAS, g["outputs"]):
        assert np.array_equal(rc.perspective(o, l, fov, focal, w, h).view(np.uint32), want.view(np.uint32))
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 320, 240)
    _, mk = G.make_masks(reference, Reference(REF_FMA_SO), oracle, cam, demo_scene, tex, sky, 4)
    for name, m in mk.items():
        assert np.array_equal(np.packbits(m), golden_masks[f"render_map_320x240_d4_{name}"]), name


def test_raygen_bit_exact(oracle, reference):
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 200, 150)
    assert np.array_equal(oracle.raygen(cam).view(np.uint32), reference.raygen(cam).view(np.uint32))


# (the known-answer test against the reference's committed out/scene.png lives in tests/test_reference_fixture.py: it runs from
#  committed fixtures, on the GPU box too)


@pytest.mark.parametrize("seed", range(40))
def test_random_scenes_bit_exact(oracle, reference, tex, sky, seed):
    """Fuzz: random primitive counts (including none), random materials / textures / cameras / depths."""
    from fuzz_scenes import random_scene
    sc, cam, depth = random_scene(seed)
    c = oracle.camera(cam["origin"], cam["look"], cam["fov"], 1.0, 72, 48)
    want, _ = reference.render(c, sc, tex, sky, depth)
    got, _, cnt = oracle.render(c, sc, tex, sky, depth)
    # the x86 build of the reference converts out-of-range floats to INT_MIN where the oracle saturates like AMD
    # hardware; only compare scenes in which no such conversion happened (the counters tell)
    if cnt.int_cast_oor == 0 and cnt.oob_reads == 0:
        assert np.array_equal(got, want), f"seed {seed}: {(got != want).sum()} pixels differ"
