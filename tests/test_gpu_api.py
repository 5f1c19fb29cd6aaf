"""The drop-in boundary on a real GPU: call protocol of raypng.c / rayinteractive.c, the lazily
materialised ray buffer, PNG ingest, strips, timing, and the reference's print+exit(1) errors."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import CAM, ROOT, channel_diff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import torch  # noqa: F401
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    return Renderer


def test_two_kernel_path_equals_fused_path(R, demo_scene, tex, sky):
    """CLWRAP_FUSE=0 runs raygen -> 64 B/px ray buffer -> raytracer like the reference; same bits."""
    for strict in (True, False):
        outs = []
        for fuse in (True, False):
            r = R(demo_scene, tex, sky, 200, 150, depth=15, strict=strict, fuse=fuse)
            r.look(**CAM)
            outs.append(r.render())
            r.release()
        assert np.array_equal(outs[0], outs[1])


def test_ray_buffer_readback_is_the_raygen_kernel_output(R, oracle, demo_scene, tex, sky, golden_frames):
    """buffers[0][8] stays virtual in fused mode until somebody reads it; the bytes are raygen.cl's."""
    want = golden_frames["raygen_160x120"].view(np.uint32)
    for strict in (True, False):
        for fuse in (True, False):
            r = R(demo_scene, tex, sky, 160, 120, depth=4, strict=strict, fuse=fuse)
            r.look(**CAM)
            r.render()
            rays = r.read_rays()
            r.release()
            assert np.array_equal(rays.view(np.uint32), want)


def test_camera_args_can_be_reset_between_frames(R, oracle, demo_scene, tex, sky):
    """rayinteractive.c:98-103 re-sends raygen args 0-5 on every key event."""
    w, h = 160, 120
    r = R(demo_scene, tex, sky, w, h, depth=4, strict=True)
    cams = [CAM, dict(origin=(0.9, 2.5, -7.9), look=(0.15, -0.05, 1.0), fov=90.0, focal=1.0),
            dict(origin=(1.2, 2.6, -7.5), look=(-0.1, -0.1, 1.0), fov=90.0, focal=1.0)]
    for cam in cams + cams[:1]:
        r.look(**cam)
        got = r.render()
        want, _, _ = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), demo_scene, tex, sky, 4)
        assert (channel_diff(got, want) == 0).mean() >= 0.999
    r.release()


def test_png_ingest_equals_raw_ingest(R, demo_scene, tex, sky, tmp_path):
    from example_gui_opencl_raytracer_amd import api
    paths = []
    for i in range(4):
        p = str(tmp_path / f"layer{i}.png")
        api.write_png_rgba(p, tex[i])
        paths.append(p)
    skyp = str(tmp_path / "sky.png")
    api.write_png_rgba(skyp, sky[0])
    a = R(demo_scene, tex, sky, 160, 120, depth=4, strict=True)
    a.look(**CAM)
    b = R(demo_scene, None, None, 160, 120, depth=4, strict=True, texture_paths=paths, skybox_path=skyp)
    b.look(**CAM)
    assert np.array_equal(a.render(), b.render())
    a.release(); b.release()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_row_strips_are_bit_identical_to_the_full_frame(R, demo_scene, tex, sky, world):
    """Multi-GPU parity: global ids under sharding -> strips equal the 1-GPU rows."""
    from example_gui_opencl_raytracer_amd.renderer import strip_rows
    w, h, depth = 320, 200, 4
    for strict in (True, False):
        full = R(demo_scene, tex, sky, w, h, depth=depth, strict=strict)
        full.look(**CAM)
        want = full.render()
        full.release()
        for rank in range(world):
            r0, rows = strip_rows(h, world, rank)
            s = R(demo_scene, tex, sky, w, h, depth=depth, strict=strict, first_row=r0, rows=rows)
            s.look(**CAM)
            assert np.array_equal(s.render(), want[r0 * w:(r0 + rows) * w])
            s.release()


@pytest.mark.parametrize("world", [2, 8])
def test_interleaved_bands_are_bit_identical_to_the_full_frame(R, demo_scene, tex, sky, world):
    """bench.py's sharding: rank r renders every world-th 8-row band with global ids."""
    w, h, depth = 200, 16 * world, 4
    full = R(demo_scene, tex, sky, w, h, depth=depth, strict=False)
    full.look(**CAM)
    want = full.render().reshape(h // 8, 8 * w)
    full.release()
    for rank in range(world):
        for fuse in (True, False):
            s = R(demo_scene, tex, sky, w, h, depth=depth, strict=False, bands=(world, rank), fuse=fuse)
            s.look(**CAM)
            got = s.render().reshape(-1, 8 * w)
            assert np.array_equal(got, want[rank::world])
            s.release()


def test_bands_with_the_grid_the_deep_build_and_the_sorted_dispatch(R, tex, sky):
    """Everything at once: a 400-sphere scene (uniform grid), depth 6 (deep build with the sparse-tail loop),
    interleaved bands, and a second frame per renderer (cost-sorted dispatch order) -- still the single-GPU bits."""
    from example_gui_opencl_raytracer_amd import scene
    sc = scene.sphere_grid_scene(20, 20)
    g = scene.glass()
    for name in ("ambient", "diffuse", "specular", "shininess", "transperent", "dielectric", "n", "reflectivity"):
        sc.spheres["material"][name][::2] = g[name]
    sc = scene.Scene(sc.spheres, sc.planes, sc.lights)
    cam = dict(origin=(0.0, 5.0, -5.0), look=(0.0, -0.5, 1.0), fov=90.0, focal=1.0)
    w, h, world = 200, 96, 3
    for strict in (True, False):
        full = R(sc, tex, sky, w, h, depth=6, strict=strict)
        full.look(**cam)
        want = full.render()
        assert np.array_equal(full.render(), want)            # second frame: sorted order
        full.release()
        want = want.reshape(h // 8, 8 * w)
        for rank in range(world):
            s = R(sc, tex, sky, w, h, depth=6, strict=strict, bands=(world, rank))
            s.look(**cam)
            for _ in range(2):
                assert np.array_equal(s.render().reshape(-1, 8 * w), want[rank::world])
            s.release()


def test_external_framebuffer_and_stream(R, demo_scene, tex, sky):
    """bench.py's plumbing: torch owns the framebuffer and the stream, the shim renders into it."""
    import torch
    w, h = 160, 120
    fb = torch.zeros(w * h, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    r = R(demo_scene, tex, sky, w, h, depth=4, strict=True, framebuffer_ptr=fb.data_ptr())
    r.w.set_stream(side.cuda_stream)
    r.w.set_async(True)
    r.look(**CAM)
    r.render(readback=False)
    side.synchronize()
    own = R(demo_scene, tex, sky, w, h, depth=4, strict=True)
    own.look(**CAM)
    assert np.array_equal(fb.cpu().numpy().view(np.uint32), own.render())
    r.release(); own.release()


def test_partial_launch_and_reinit(R, oracle, demo_scene, tex, sky):
    """array_size smaller than the frame: the reference launches ceil(n/256)*256 work-items guarded by
    id < total (opencl_wrap.c:374, raytracing.cl:24) -- the rest of the framebuffer stays untouched.  Then the
    same cl_wrap struct is released and initialised again (opencl_wrap.c:400-416)."""
    import ctypes as C
    from example_gui_opencl_raytracer_amd import api
    w, h = 160, 120
    want, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), demo_scene, tex, sky, 4)
    r = R(demo_scene, tex, sky, w, h, depth=4, strict=True)
    r.look(**CAM)
    n = 5000                                             # rounds up to 5120 work-items
    full = r.render()                                    # the whole frame through the normal path
    assert (channel_diff(full, want) == 0).mean() >= 0.999
    r.release()
    # a fresh wrap with another camera fills its framebuffer; a PARTIAL launch with the test camera then
    # re-traces ids [0, 5120) only: those equal the full frame, the rest keeps the other camera's pixels
    r = R(demo_scene, tex, sky, w, h, depth=4, strict=True)
    r.look(origin=(1.5, 2.0, -6.0), look=(-0.1, -0.1, 1.0), fov=90.0, focal=1.0)
    other = r.render()
    r.look(**CAM)
    part = np.empty(w * h, np.uint32)
    r.w.output(n, 0, 0, 0, 0, None)
    r.w.output(n, part.nbytes, 1, 1, 10, part)
    assert np.array_equal(part[:5120], full[:5120]) and np.array_equal(part[5120:], other[5120:])
    assert not np.array_equal(other[:5120], full[:5120])
    r.release()
    assert r.w.w.impl is None or r.w.w.impl == 0
    again = R(demo_scene, tex, sky, w, h, depth=4, strict=True)          # re-init after release
    again.look(**CAM)
    assert np.array_equal(again.render(), full)
    again.release()


def test_timing_log(R, demo_scene, tex, sky):
    r = R(demo_scene, tex, sky, 320, 240, depth=4)
    r.look(**CAM)
    r.render(readback=False)
    r.w.timing_reset()
    r.w.set_async(True)
    for _ in range(5):
        r.render(readback=False)
    r.w.sync()
    n, ms = r.w.timing_get(1)
    assert n == 5 and 0.0 < ms < 1000.0
    n0, _ = r.w.timing_get(0)
    assert n0 == 0                                    # fused: the raygen launch does no device work
    r.release()


def test_pipelined_readback_returns_the_same_frame(R, demo_scene, tex, sky):
    """A blocking cl_wrap_output of a >= 16 MB frame at depth <= 4 renders four strips and copies each while the next renders
    (clw_ext_set_pipeline, default on): the frame in host memory is the one the single launch + single copy gives,
    on the first frame (no tile costs yet), on later ones (cost-sorted strips) and after a camera move."""
    w, h = 2560, 1664
    frames = {}
    for on in (1, 0):
        r = R(demo_scene, tex, sky, w, h, depth=4)
        r.w.set_pipeline(on)
        r.look(**CAM)
        a = r.render().copy()
        b = r.render().copy()
        r.look(origin=(1.5, 2.0, -7.0), look=(0.1, -0.1, 1.0), fov=80.0, focal=1.0)
        c = r.render().copy()
        frames[on] = (a, b, c)
        r.release()
    for x, y in zip(frames[1], frames[0]):
        assert np.array_equal(x, y)
    assert np.array_equal(frames[1][0], frames[1][1]) and not np.array_equal(frames[1][0], frames[1][2])
    # a frame whose height is not a multiple of the strip granularity, and one too small to be split
    for (ww, hh) in ((2304, 1900), (640, 480)):
        outs = []
        for on in (1, 0):
            r = R(demo_scene, tex, sky, ww, hh, depth=3)
            r.w.set_pipeline(on)
            r.look(**CAM)
            r.render()
            outs.append(r.render().copy())
            r.release()
        assert np.array_equal(outs[0], outs[1])


# ------------------------------------------------------------ error behaviour: print "ERROR:\t..." + exit(1)
def _run(snippet):
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from example_gui_opencl_raytracer_amd import api\n" % ROOT) + snippet
    return subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)


def test_duplicate_buffer_argument_exits_like_the_reference():
    p = _run("w = api.ClWrap()\nw.load_global_data(1, 1, np.zeros(96, np.uint8))\nw.load_global_data(1, 1, np.zeros(96, np.uint8))\nprint('unreachable')")
    assert p.returncode == 1 and "ERROR:\tGiven kernel argument already in use" in p.stdout and "unreachable" not in p.stdout


def test_by_value_write_to_a_buffer_argument_exits():
    p = _run("w = api.ClWrap()\nw.load_global_data(0, 8, None, 64)\nw.load_single_data(0, 8, np.uint32(1))")
    assert p.returncode == 1 and "ERROR:\tGiven kernel argument already in use" in p.stdout       # opencl_wrap.c:176-181


def test_unknown_kernel_name_and_missing_name_exit():
    p = _run("api.ClWrap('a.cl', 'no_such_kernel')")
    assert p.returncode == 1 and "ERROR:\tCouldn't create the CL kernel from: no_such_kernel" in p.stdout
    p = _run("api.ClWrap('src/cl/raygen.cl')")
    assert p.returncode == 1 and "ERROR:\tSource file was not followed by kernel name" in p.stdout


def test_bad_image_files_exit(tmp_path):
    junk = tmp_path / "junk.png"
    junk.write_bytes(b"definitely not a png")
    p = _run(f"w = api.ClWrap()\nw.load_images(1, 8, {str(junk)!r})")
    assert p.returncode == 1 and "is not a PNG file" in p.stdout
    p = _run("w = api.ClWrap()\nw.load_images(1, 8, '/nonexistent/x.png')")
    assert p.returncode == 1 and "ERROR:\tCannot open file" in p.stdout


def test_launch_with_missing_arguments_exits():
    p = _run("w = api.ClWrap()\nw.output(64, 0, 1, 1, 10, None)")
    assert p.returncode == 1 and "ERROR:\tCouldn't run the kernel" in p.stdout


def test_argument_id_limit_exits():
    p = _run("w = api.ClWrap()\nw.load_global_data(1, 32, np.zeros(4, np.uint8))")      # __MAX_BUFFERS = 32 (opencl_wrap.h:7)
    assert p.returncode == 1 and "ERROR:\tWrong kernel ID given" in p.stdout               # the reference's wording (opencl_wrap.c:144)
    p = _run("w = api.ClWrap()\nw.output(64, 0, 7, 0, 0, None)")                          # kernel id out of range
    assert p.returncode == 1 and "ERROR:" in p.stdout


def test_environment_knobs(oracle, demo_scene, tex, sky):
    """CLWRAP_DEPTH / CLWRAP_STRICT / CLWRAP_FUSE are read by cl_wrap_init: an unchanged driver is configured by
    the environment alone (the reference bakes MAX_DEPTH into the kernel source, raytracing.cl:9)."""
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np\n"
        "import example_gui_opencl_raytracer_amd as pkg\n"
        "from example_gui_opencl_raytracer_amd import api, scene, textures\n"
        "from example_gui_opencl_raytracer_amd.renderer import Renderer\n"
        "w = api.ClWrap(); print('depth', w.get_depth()); w.release()\n"
        "r = Renderer.__new__(Renderer)\n" % ROOT)
    env = dict(os.environ, CLWRAP_DEPTH="3", CLWRAP_STRICT="1", CLWRAP_FUSE="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and "depth 3" in p.stdout, p.stdout + p.stderr
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, CLWRAP_DEPTH="99"), timeout=300)
    assert p.returncode == 1 and "ERROR:" in p.stdout
    p = _run("import ctypes as C\nL = api.load_library(); w = api.cl_wrap()\n"
             "L.cl_wrap_init(C.byref(w), C.c_uint64(1 << 1), C.c_char_p(b'a.cl'), C.c_char_p(b'raygen'), C.c_char_p(None))")   # CL_DEVICE_TYPE_CPU
    assert p.returncode == 1 and "ERROR:\tCannot find a device of the given type" in p.stdout      # opencl_wrap.c:31-34


def test_depth_out_of_range_exits():
    p = _run("w = api.ClWrap()\nw.set_depth(33)")
    assert p.returncode == 1 and "ERROR:" in p.stdout


# ------------------------------------------------------------ the reference's own driver, unchanged
REF_RAYPNG = os.path.join(ROOT, "oracle", "_ref", "raypng_hip")


@pytest.mark.skipif(not os.path.exists(REF_RAYPNG), reason="oracle/_ref/raypng_hip not built (needs /root/reference)")
def test_unchanged_raypng_driver_links_and_renders(oracle, demo_scene, tex, tmp_path):
    """The reference's raypng.c + cpu_ray.c + cpu_obj.c compiled UNCHANGED against include/opencl_wrap.h
    and linked with libopencl_wrap_hip.so (oracle/Makefile target `raypng`): runs from a scratch cwd
    holding scenes/render.map and procedural assets under the file names raypng.c hard-codes."""
    from example_gui_opencl_raytracer_amd import api, textures
    for d in ("scenes", "assets/bg", "out", "src/cl"):
        os.makedirs(tmp_path / d)
    demo_scene.save(tmp_path / "scenes" / "render.map")
    for i, name in enumerate(("cobblestone", "sand", "check", "grass")):          # raypng.c:74-78
        api.write_png_rgba(str(tmp_path / "assets" / f"{name}.png"), tex[i])
    sky = textures.skybox_cross(1024)
    api.write_png_rgba(str(tmp_path / "assets" / "bg" / "stormydays.png"), sky[0])
    p = subprocess.run([REF_RAYPNG], cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "Done, took:" in p.stdout                                               # raypng.c:96
    img = api.read_png(str(tmp_path / "out" / "scene.png"))
    assert img.shape == (600, 800, 4)
    got = (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)
    want, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 800, 600), demo_scene, tex, sky, 15)
    d = channel_diff(got, want)
    assert (d == 0).mean() >= 0.995 and (d <= 1).mean() >= 0.998                  # default = fast build


@pytest.mark.skipif(not os.path.exists(REF_RAYPNG), reason="oracle/_ref/raypng_hip not built (needs /root/reference)")
def test_unchanged_raypng_driver_reproduces_the_references_committed_render(oracle, tmp_path):
    """The HIP path against the one OUTPUT the reference holds: out/scene.png (tests/golden/reference_scene/, see
    tests/test_reference_fixture.py).  The reference's unchanged raypng.c runs in a scratch tree holding the reference's own
    render.map and asset PNGs, with the transparent-shadow factor of the version that rendered the image (CLWRAP_THROUGH=1.0) --
    first the strict build, which must ALSO equal the oracle bit for bit, then the default fast build; both are held to the bar
    the oracle itself meets against that image (>= 99.0 % of pixels bit-equal, >= 99.5 % within 1 LSB, <= 1000 pixels more than
    1 LSB off outside the frame's discontinuity mask)."""
    import shutil
    from example_gui_opencl_raytracer_amd import api
    from test_reference_fixture import DEPTH, FIX, H, W, check_fixture_bar, fixture_stats, load_fixture
    from test_gpu_parity import report
    scene, tex, sky, want, mask, _ = load_fixture()
    for d in ("scenes", "assets/bg", "out"):
        os.makedirs(tmp_path / d)
    shutil.copy(os.path.join(FIX, "render.map"), tmp_path / "scenes" / "render.map")
    for n in ("cobblestone", "sand", "check", "grass"):
        shutil.copy(os.path.join(FIX, n + ".png"), tmp_path / "assets" / (n + ".png"))
    shutil.copy(os.path.join(FIX, "stormydays.png"), tmp_path / "assets" / "bg" / "stormydays.png")
    try:
        oracle.set_transparent_through(1.0)
        orc, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, W, H), scene, tex, sky, DEPTH)
    finally:
        oracle.set_transparent_through(0.8)
    for strict in ("1", "0"):
        env = dict(os.environ, CLWRAP_THROUGH="1.0", CLWRAP_STRICT=strict)
        p = subprocess.run([REF_RAYPNG], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "Done, took:" in p.stdout, p.stdout + p.stderr
        img = api.read_png(str(tmp_path / "out" / "scene.png"))
        assert img.shape == (H, W, 4)
        got = (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)
        st = fixture_stats(got, want, mask)
        report(dict(test="raypng_hip vs the reference's out/scene.png", strict=int(strict), **st,
                    differs_from_oracle=int((got != orc).sum())))
        check_fixture_bar(st, f"raypng_hip strict={strict}")
        if strict == "1":
            assert (got != orc).sum() <= 2, int((got != orc).sum())      # (ocml vs glibc sinf / cosf / powf: see check_exact)
        os.remove(tmp_path / "out" / "scene.png")


REF_INTERACTIVE = os.path.join(ROOT, "oracle", "_ref", "rayinteractive_hip")


def _scratch_tree(tmp_path, demo_scene, tex, sky):
    from example_gui_opencl_raytracer_amd import api
    for d in ("scenes", "assets/bg", "out"):
        os.makedirs(tmp_path / d, exist_ok=True)
    demo_scene.save(tmp_path / "scenes" / "render.map")
    for i, name in enumerate(("cobblestone", "sand", "check", "grass")):
        api.write_png_rgba(str(tmp_path / "assets" / f"{name}.png"), tex[i])
    api.write_png_rgba(str(tmp_path / "assets" / "bg" / "stormydays.png"), sky[0])


@pytest.mark.skipif(not os.path.exists(REF_INTERACTIVE), reason="oracle/_ref/rayinteractive_hip not built (needs /root/reference)")
def test_unchanged_rayinteractive_driver_runs_headless(oracle, demo_scene, tex, tmp_path):
    """The reference's rayinteractive.c, unchanged, with the headless minifb stand-in (tools/minifb_stub):
    the frame loop of rayinteractive.c:183-197 and the per-key argument re-upload of :98-103."""
    from example_gui_opencl_raytracer_amd import api, textures
    sky = textures.skybox_cross(1024)
    _scratch_tree(tmp_path, demo_scene, tex, sky)

    def run(keys, frames, dump):
        env = dict(os.environ, MFB_STUB_FRAMES=str(frames), MFB_STUB_KEYS=keys, MFB_STUB_DUMP=str(tmp_path / dump))
        p = subprocess.run([REF_INTERACTIVE], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        assert f"minifb-stub: {frames} frames" in p.stdout
        img = api.read_png(str(tmp_path / dump))
        assert img.shape == (600, 800, 4)
        return (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)

    still = run("", 5, "still.png")                          # no key: camera of rayinteractive.c:111-115
    want, _, _ = oracle.render(oracle.camera((0.8, 2.5, -8.0), (0.0, 0.0, 1.0), 90.0, 1.0, 800, 600), demo_scene, tex, sky, 15)
    d = channel_diff(still, want)
    assert (d == 0).mean() >= 0.995 and (d <= 1).mean() >= 0.998
    moved = run("WWWWllllSZ", 40, "moved.png")               # forward x4, turn left x4, back, down ... cycled
    assert (channel_diff(moved, still) > 8).mean() > 0.2     # the camera really moved


RAYBENCH = os.path.join(ROOT, "tools", "raybench")


@pytest.mark.skipif(not os.path.exists(RAYBENCH), reason="tools/raybench not built")
def test_pure_c_bench_driver(oracle, demo_scene, tex, tmp_path):
    """tools/raybench.c: host side in C, sizes as size_t, both scene archive formats, PNG out."""
    import json
    from example_gui_opencl_raytracer_amd import api, scene, textures
    sky = textures.skybox_cross(512)
    _scratch_tree(tmp_path, demo_scene, tex, sky)
    p = subprocess.run([RAYBENCH, "-w", "320", "-h", "200", "-d", "4", "-n", "5", "-S", "-o", "out/c.png"], cwd=tmp_path,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["frame"] == "320x200" and line["spheres"] == 4 and line["trace_kernel_ms"] > 0
    img = api.read_png(str(tmp_path / "out" / "c.png"))
    got = (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)
    want, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 320, 200), demo_scene, tex, sky, 4)
    assert (channel_diff(got, want) == 0).mean() >= 0.999
    big = scene.sphere_grid_scene(20, 20)                           # 400 spheres: extended archive + wide counts
    big.save(tmp_path / "scenes" / "big.map")
    p = subprocess.run([RAYBENCH, "-w", "96", "-h", "64", "-d", "2", "-n", "2", "-S", "-s", "scenes/big.map", "-o", "out/b.png"],
                       cwd=tmp_path, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert json.loads(p.stdout.strip().splitlines()[-1])["spheres"] == 400
    img = api.read_png(str(tmp_path / "out" / "b.png"))
    got = (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)
    want, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 96, 64), big, tex, sky, 2)
    assert (channel_diff(got, want) == 0).mean() >= 0.999


def test_scene_arrays_are_snapshotted_until_invalidated(R, oracle, demo_scene, tex, sky):
    """The shim prepares its geometry once per (scene buffers, counts).  A caller that owns a scene buffer on the device
    (clw_ext_bind_device_buffer) and rewrites it in place must call clw_ext_invalidate_scene: before that the old scene is
    still rendered, after it the new one."""
    import torch
    from example_gui_opencl_raytracer_amd import api
    from example_gui_opencl_raytracer_amd.scene import RAY, Scene
    w, h, depth = 96, 64, 4
    sph = torch.from_numpy(np.frombuffer(demo_scene.spheres.tobytes(), np.uint8).copy()).cuda()
    cw = api.ClWrap()
    cw.set_depth(depth); cw.set_strict(True)
    cam = api.perspective(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    f3 = lambda v: np.array([v[0], v[1], v[2], 0.0], np.float32)
    for a, v in enumerate((f3(cam.im_corner), f3(cam.origin), f3(cam.up), f3(cam.right), np.float32(cam.w_factor), np.float32(cam.h_factor),
                           np.uint32(w), np.uint32(h))):
        cw.load_single_data(0, a, v)
    cw.load_global_data(0, 8, None, RAY.itemsize * w * h)
    cw.load_single_data(1, 0, cw.buffer_handle(0, 8))
    cw.bind_device_buffer(1, 1, sph.data_ptr(), sph.numel())            # caller-owned sphere array
    cw.load_global_data(1, 2, demo_scene.planes)
    cw.load_global_data(1, 3, demo_scene.lights)
    for a, n in ((4, len(demo_scene.spheres)), (5, len(demo_scene.planes)), (6, len(demo_scene.lights))):
        cw.load_single_data(1, a, np.uint8(n))
    cw.load_single_data(1, 7, np.uint32(w * h))
    cw.load_images_raw(1, 8, tex); cw.load_images_raw(1, 9, sky)
    cw.load_global_data(1, 10, None, 4 * w * h)

    def frame():
        out = np.empty(w * h, np.uint32)
        cw.output(w * h, 0, 0, 0, 0, None)
        cw.output(w * h, out.nbytes, 1, 1, 10, out)
        return out
    first = frame()
    moved = demo_scene.spheres.copy()
    moved["origin"][0] = (1.5, 0.5, -2.0)                       # sphere 0 moves into view
    sph.copy_(torch.from_numpy(np.frombuffer(moved.tobytes(), np.uint8).copy()))
    torch.cuda.synchronize()
    assert np.array_equal(frame(), first)                       # still the snapshot
    cw.invalidate_scene()
    second = frame()
    cw.release()
    ocam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    want, _, _ = oracle.render(ocam, Scene(moved, demo_scene.planes, demo_scene.lights), tex, sky, depth)
    assert np.array_equal(second, want) and not np.array_equal(second, first)


def test_traced_ray_counter(R, oracle, demo_scene, tex, sky):
    """clw_ext_read_counters_ex: word 8 = the shadow rays really traced; the reference casts 2 per light per shaded hit,
    the kernel elides those of surfaces whose specular and diffuse coefficients are both zero (glass)."""
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    w, h, depth = 160, 120, 4
    r = Renderer(demo_scene, tex, sky, w, h, depth=depth, strict=True)
    r.look(**CAM)
    r.w.enable_counters(1)
    r.render(readback=False)
    c = r.w.read_counters()
    r.release()
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    _, _, cnt = oracle.render(cam, demo_scene, tex, sky, depth)
    assert c["shadow_rays"] == cnt.shadow_rays and c["segments"] == cnt.segments
    assert 0 < c["shadow_rays_traced"] < c["shadow_rays"] and c["shadow_rays_traced"] % 2 == 0
