"""GPU parity tests: the HIP path, called through the C-ABI (cl_wrap_*), against the oracle.

Bars (written here as the tolerances):
  strict build  (no contraction, IEEE divide/sqrt)   EVERY pixel bit-exact (np.array_equal), ray counts equal
  fast build    (explicit FMAs, native rcp/sqrt, ...)  >= 99.8 % bit-exact, >= 99.9 % within 1 LSB per channel on
                                                     render.map frames (99.9 / 99.95 % at the full C2 size; looser on
                                                     the chaotic glass-field scene); every pixel that differs by more
                                                     than 1 LSB lies on the frame's discontinuity mask
                                                     (tests/golden/masks.npz) and the float radiance of EVERY pixel off
                                                     the mask is within 1e-4 (north_star's tolerance)
  integer / index work (raygen records are fp but must be bit-exact; ids, packing) bit-exact.
"""
import numpy as np
import pytest

from conftest import CAM, channel_diff, frame_mask

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def R():
    import torch  # noqa: F401  (the shim then shares torch's ROCm runtime)
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    return Renderer


def gpu_frame(R, sc, tex, sky, w, h, depth, strict, cam=CAM, rgb=False, **kw):
    r = R(sc, tex, sky, w, h, depth=depth, strict=strict, **kw)
    r.look(**cam)
    out = r.render_rgb() if rgb else r.render()
    r.release()
    return out


def check_exact(got, want, what="", allow=0):
    """The strict build's bar: every pixel equal.  (At the multi-megapixel sizes a handful of pixels differ where the device libm and the
    oracle's glibc round one sinf / cosf / powf value differently: those frames go through pin_strict_residual instead.)"""
    bad = int((got != want).sum())
    report(dict(test=what, pixels=int(got.size), differing=bad, allowed=allow))
    assert got.shape == want.shape and bad <= allow, f"{what}: {bad} of {got.size} pixels differ from the oracle (allowed {allow})"


def device_libm_rows(w, log):
    """The rows of an oracle libm log (oracle_py.Oracle.trace_libm) with the DEVICE's results: the same inputs through the strict kernel's
    own sin / cos / pow (clw_ext_unit ops 8 and 9; the angles are formed from the xorshift values exactly as the kernel forms them)."""
    out = []
    pairs = log[log[:, 0] == 0.0]
    if len(pairs):
        th = w.unit(8, pairs[:, 1:2], 2, aux=1)          # sin, cos of fl32(2 pi u1)
        ph = w.unit(8, pairs[:, 2:3], 2, aux=0)          # sin, cos of fl32(pi u2)
        g = log[(log[:, 0] > 0.0) & (log[:, 0] < 5.0)].reshape(len(pairs), 4, 4)    # per sample: sin phi, cos phi, sin theta, cos theta
        for k in range(len(pairs)):
            for j, dev in enumerate((ph[k, 0], ph[k, 1], th[k, 0], th[k, 1])):
                out.append((g[k, j, 0], g[k, j, 1], 0.0, dev))
    pw = log[log[:, 0] == 5.0]
    if len(pw):
        dv = w.unit(9, pw[:, 1:3], 1)
        out += [(5.0, pw[k, 1], pw[k, 2], dv[k, 0]) for k in range(len(pw))]
    return np.array(out, np.float32).reshape(-1, 4)


def pin_strict_residual(w, got, want, oracle, cam, sc, tex, sky, depth, what, limit=16):
    """The strict build is IEEE- and order-exact; what is left at multi-megapixel sizes are pixels where the device libm (ocml) and
    the oracle's glibc round a sinf / cosf / powf value one ulp apart and that ulp decides a shadow sample or the 8-bit truncation.  This
    turns that sentence into a test.  For every pixel that differs: the oracle re-traces it with its libm calls logged, the same inputs go
    through the kernel's own strict sin / cos / pow, and the oracle traces the pixel AGAIN with the device's results substituted for
    glibc's (repeated while the substitution makes the pixel evaluate inputs it had not seen): the result must be the GPU's pixel, bit for
    bit -- nothing but those libm roundings separates the two.  `w` = a live strict ClWrap; at most `limit` pixels may differ.  The
    divergent calls are appended to gpurun_out/libm_divergence.jsonl (examples are committed in tests/golden/libm_divergence.json)."""
    import json
    import os
    from conftest import ROOT
    bad = np.nonzero(got != want)[0]
    report(dict(test=what, pixels=int(got.size), differing=int(bad.size), allowed=limit))
    assert got.shape == want.shape and bad.size <= limit, f"{what}: {bad.size} of {got.size} pixels differ from the oracle"
    bits = lambda x: int(np.float32(x).view(np.uint32))
    names = {1: "sinf(pi*u)", 2: "cosf(pi*u)", 3: "sinf(2pi*u)", 4: "cosf(2pi*u)", 5: "powf"}
    found = []
    for pid in bad:
        px, log = oracle.trace_libm(cam, sc, tex, sky, depth, int(pid))
        assert px == int(want[pid])
        table = device_libm_rows(w, log)
        for _ in range(6):
            px2, log2 = oracle.trace_libm(cam, sc, tex, sky, depth, int(pid), overrides=table)
            calls = log2[log2[:, 0] > 0.0]
            have = {(r[0], bits(r[1]), bits(r[2])) for r in table}
            if all((r[0], bits(r[1]), bits(r[2])) in have for r in calls):
                break
            table = np.concatenate([table, device_libm_rows(w, log2)])      # the substituted run took another path: its new inputs too
        assert px2 == int(got[pid]), (f"{what}: pixel {int(pid)}: GPU {int(got[pid]):06x}, oracle {px:06x}, oracle with the device's libm "
                                      f"values {px2:06x} -- something other than libm rounding separates them")
        glibc = log[log[:, 0] > 0.0]
        dev = {(r[0], bits(r[1]), bits(r[2])): r[3] for r in table}
        for r in glibc:
            d = dev[(r[0], bits(r[1]), bits(r[2]))]
            if bits(d) != bits(r[3]):
                found.append(dict(frame=what, pixel=int(pid), fn=names[int(r[0])], a_bits=bits(r[1]), b_bits=bits(r[2]), glibc_bits=bits(r[3]), ocml_bits=bits(d)))
        assert any(f["pixel"] == int(pid) for f in found)
    d = os.path.join(ROOT, "gpurun_out")
    if found and os.path.isdir(d):
        with open(os.path.join(d, "libm_divergence.jsonl"), "a") as f:
            for r in found:
                f.write(json.dumps(r) + "\n")


def report(rec):
    """Append one JSON line of measured parity figures to gpurun_out/parity_report.jsonl (copied to profiles/)."""
    import json
    import os
    from conftest import ROOT
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_report.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")


def check(got, want, exact_min, le1_min=None):
    d = channel_diff(got, want)
    exact, le1 = (d == 0).mean(), (d <= 1).mean()
    assert exact >= exact_min, f"only {exact:.5f} of pixels bit-exact (need {exact_min})"
    if le1_min is not None:
        assert le1 >= le1_min, f"only {le1:.5f} within 1 LSB (need {le1_min})"
    return exact


# ------------------------------------------------------------ golden frames (reference kernels' own output)
GOLDEN_FRAMES = [(160, 120, 1), (160, 120, 4), (160, 120, 15), (320, 240, 4), (640, 480, 1)]   # (640, 480, 1) = BASELINE config C1


@pytest.mark.parametrize("w,h,depth", GOLDEN_FRAMES)
def test_strict_matches_golden_frames(R, demo_scene, tex, sky, golden_frames, w, h, depth):
    got = gpu_frame(R, demo_scene, tex, sky, w, h, depth, strict=True)
    check_exact(got, golden_frames[f"render_map_{w}x{h}_d{depth}"], f"strict {w}x{h} d{depth}")


@pytest.mark.parametrize("w,h,depth", GOLDEN_FRAMES)
def test_fast_matches_golden_frames(R, demo_scene, tex, sky, golden_frames, w, h, depth):
    got = gpu_frame(R, demo_scene, tex, sky, w, h, depth, strict=False)
    check(got, golden_frames[f"render_map_{w}x{h}_d{depth}"], 0.998, 0.999)


MASK_REPORT = {}


@pytest.mark.parametrize("w,h,depth", GOLDEN_FRAMES + [(1280, 720, 4)])
def test_fast_outliers_lie_on_the_discontinuity_mask(R, oracle, demo_scene, tex, sky, golden_frames, golden_masks, w, h, depth):
    """SURVEY.md 8(c) bar (2)+(3) for the BENCHMARKED build.  The mask of a frame (oracle/gen_golden.py) marks the pixels
    where the reference's own output is discontinuous or decided within rounding error: two legal builds of the
    reference kernels (contraction off / on) disagree there (dilated 1 px), or the oracle's radiance is not locally
    linear under camera shifts of 1/64 .. 1/1024 pixel, or a sphere discriminant / texel-index truncation has a
    relative margin below 1e-5 / 2e-6.  Claim: every fast-build pixel more than 1 LSB away from the reference frame lies
    on the mask, and OFF the mask the float radiance of every pixel is within 1e-4 of the oracle's."""
    import zlib
    key = f"render_map_{w}x{h}_d{depth}"
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    want, want_rgb, _ = oracle.render(cam, demo_scene, tex, sky, depth, want_rgb=True)
    assert zlib.crc32(want.tobytes()) == int(golden_masks[key + "_crc32"][0])     # = the reference kernels' frame
    got, rgb = gpu_frame(R, demo_scene, tex, sky, w, h, depth, strict=False, rgb=True)
    mask = frame_mask(golden_masks, key, w * h)
    d = channel_diff(got, want)
    outl = d > 1
    with np.errstate(invalid="ignore"):
        err = np.abs(rgb - want_rgb).max(1)
    err = np.where(np.isnan(rgb).any(1) & np.isnan(want_rgb).any(1), 0.0, np.nan_to_num(err, nan=np.inf))
    far = err > 1e-4
    parts = {n: frame_mask(golden_masks, key, w * h, (n,)) for n in ("fma", "jitter", "margin")}
    MASK_REPORT[key] = dict(mask=float(mask.mean()), differing=int((d > 0).sum()), outliers=int(outl.sum()),
                            outliers_off_mask=int((outl & ~mask).sum()), rgb_far=int(far.sum()), rgb_far_off_mask=int((far & ~mask).sum()),
                            outliers_on={n: int((outl & m).sum()) for n, m in parts.items()}, max_err_off_mask=float(err[~mask].max()))
    report(dict(test="fast vs discontinuity mask", frame=key, **MASK_REPORT[key]))
    assert mask.mean() < (0.10 if w >= 640 else 0.20)
    assert (outl & ~mask).sum() == 0, f"{key}: {(outl & ~mask).sum()} outliers (> 1 LSB) off the mask: {MASK_REPORT[key]}"
    assert (far & ~mask).sum() == 0, f"{key}: {(far & ~mask).sum()} pixels off the mask differ by more than 1e-4: {MASK_REPORT[key]}"


@pytest.mark.parametrize("key", ["c3_1024x1024_d8", "c4_480x270_d4"])
def test_fast_outliers_lie_on_the_discontinuity_mask_c3_c4(R, oracle, tex, sky, key):
    """The same claim as above for the BENCHMARKED build on the scenes of configs C3 and C4, at sizes the oracle finishes in
    seconds (tests/golden/masks_scenes.npz, oracle/gen_golden.py `scenes`): every pixel more than 1 LSB away from the oracle's
    frame lies on the frame's discontinuity mask, and off the mask the float radiance is within 1e-4.
    C3 (1024x1024, depth 8, 64 glass spheres): mask 13.5 % of the pixels.
    C4 (480x270, depth 4, 10 000 spheres): the scene IS a discontinuity field at any affordable size -- spheres of a few pixels, rays
    that pass them at up to a hundred units (b*b - 4ac of primitives.cl:181 a difference of numbers equal to 5-7 digits): the
    mask holds 77 % of the pixels, so for this scene the sharper statement is the second one below: the fast build is no
    further from the oracle than the oracle's own FMA-contracted build is (96.1 % of the pixels bit-equal), two legal builds
    of one source."""
    import os
    import zlib
    from conftest import GOLDEN
    from oracle.gen_golden import SCENE_FRAMES
    gm = dict(np.load(os.path.join(GOLDEN, "masks_scenes.npz")))
    f = SCENE_FRAMES[key]
    sc, w, h, depth = f["scene"](), f["w"], f["h"], f["depth"]
    cam = dict(origin=f["origin"], look=f["look"], fov=90.0, focal=1.0)
    want, want_rgb, _ = oracle.render(oracle.camera(f["origin"], f["look"], 90.0, 1.0, w, h), sc, tex, sky, depth, want_rgb=True)
    assert zlib.crc32(want.tobytes()) == int(gm[key + "_crc32"][0])
    got, rgb = gpu_frame(R, sc, tex, sky, w, h, depth, strict=False, cam=cam, rgb=True)
    mask = frame_mask(gm, key, w * h)
    d = channel_diff(got, want)
    outl = d > 1
    with np.errstate(invalid="ignore"):
        err = np.abs(rgb - want_rgb).max(1)
    err = np.where(np.isnan(rgb).any(1) & np.isnan(want_rgb).any(1), 0.0, np.nan_to_num(err, nan=np.inf))
    far = err > 1e-4
    rec = dict(mask=float(mask.mean()), exact=float((d == 0).mean()), within_1lsb=float((d <= 1).mean()), outliers=int(outl.sum()),
               outliers_off_mask=int((outl & ~mask).sum()), rgb_far_off_mask=int((far & ~mask).sum()),
               max_err_off_mask=float(err[~mask].max()), oracle_fma_agree=float(gm[key + "_fma_agree"][0]))
    report(dict(test="fast vs discontinuity mask", frame=key, **rec))
    assert (outl & ~mask).sum() == 0, rec
    assert (far & ~mask).sum() == 0, rec
    if key.startswith("c3"):
        assert mask.mean() < 0.15 and rec["exact"] >= 0.995, rec
    else:
        assert rec["exact"] >= rec["oracle_fma_agree"] - 0.01, rec
    strict = gpu_frame(R, sc, tex, sky, w, h, depth, strict=True, cam=cam)
    check_exact(strict, want, f"strict {key}")


def test_fast_float_radiance_within_1e_4(R, oracle, demo_scene, tex, sky):
    """north_star's "1e-4 per-channel float tolerance", on the optional float output."""
    w, h, depth = 320, 240, 4
    _, rgb = gpu_frame(R, demo_scene, tex, sky, w, h, depth, strict=False, rgb=True)
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    _, want, _ = oracle.render(cam, demo_scene, tex, sky, depth, want_rgb=True)
    err = np.abs(rgb - want).max(1)
    assert (err <= 1e-4).mean() >= 0.995
    _, rgb_s = gpu_frame(R, demo_scene, tex, sky, w, h, depth, strict=True, rgb=True)
    # the strict build's radiance: the same bits as the oracle's except where ocml and glibc round sinf / cosf / powf
    # differently (a last-place difference in one light's term)
    same = (rgb_s.view(np.uint32) == want.view(np.uint32)).all(1)
    with np.errstate(invalid="ignore"):
        worst = float(np.nanmax(np.abs(rgb_s - want)))
    report(dict(test="strict float radiance 320x240 d4", bit_equal=float(same.mean()), max_abs_err=worst,
                fast_within_1e_4=float((err <= 1e-4).mean())))
    assert same.mean() >= 0.99 and worst <= 2e-6


# ------------------------------------------------------------ other scenes / cameras vs the oracle
CAMERAS = [((0.8, 2.5, -8.0), (0.0, 0.0, 1.0), 90.0),        # rayinteractive.c:111-115
           ((-3.0, 0.6, 0.5), (1.0, 0.05, 0.3), 70.0),       # low, grazing the floor
           ((0.9, 0.7, 1.4), (0.3, -0.2, 1.0), 100.0),       # inside glass sphere #2
           ((1.0, 9.0, 1.0), (0.01, -1.0, 0.02), 60.0)]      # looking down


@pytest.mark.parametrize("origin,look,fov", CAMERAS)
def test_cameras(R, oracle, demo_scene, tex, sky, origin, look, fov):
    w, h, depth = 128, 96, 15
    cam = dict(origin=origin, look=look, fov=fov, focal=1.0)
    want, _, _ = oracle.render(oracle.camera(origin, look, fov, 1.0, w, h), demo_scene, tex, sky, depth)
    check_exact(gpu_frame(R, demo_scene, tex, sky, w, h, depth, True, cam=cam), want, f"camera {origin}")
    check(gpu_frame(R, demo_scene, tex, sky, w, h, depth, False, cam=cam), want, 0.99, 0.995)


def test_degenerate_camera_nan_propagation(R, oracle, demo_scene, tex, sky):
    """Camera at the exact centre of glass sphere #2: view and light directions cancel exactly for half the
    pixels, normalize(0) = NaN poisons the radiance and the pack turns it into 0 (clamp of NaN).  The strict
    build must reproduce that bit for bit.  (The fast build legitimately differs here: a 1-ulp change removes
    the exact cancellation, so it is not compared.)"""
    w, h, depth = 128, 96, 15
    origin, look, fov = (0.8, 0.8, 1.5), (0.3, -0.2, 1.0), 100.0
    want, rgb, _ = oracle.render(oracle.camera(origin, look, fov, 1.0, w, h), demo_scene, tex, sky, depth, want_rgb=True)
    assert np.isnan(rgb).any()
    got = gpu_frame(R, demo_scene, tex, sky, w, h, depth, True, cam=dict(origin=origin, look=look, fov=fov, focal=1.0))
    check_exact(got, want, "NaN propagation")


@pytest.mark.parametrize("depth", [1, 2, 4, 5, 8, 15, 32])
def test_depths_including_the_scratch_stack(R, oracle, demo_scene, tex, sky, depth):
    """depth <= 4 keeps the DFS stack in LDS; deeper levels spill to scratch (other kernel build)."""
    w, h = 160, 120
    want, _, cnt = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), demo_scene, tex, sky, depth)
    check_exact(gpu_frame(R, demo_scene, tex, sky, w, h, depth, True), want, f"depth {depth}")
    if depth >= 8:
        assert cnt.max_stack > 4          # the deep levels really are exercised


def test_glass_field_divergence_scene(R, oracle, tex, sky):
    """Config C3's scene at a size the oracle renders in seconds: 64 dielectric spheres, depth 8."""
    from example_gui_opencl_raytracer_amd import scene
    sc = scene.dielectric_field_scene(8)
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
    w, h, depth = 256, 256, 8
    want, _, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky, depth)
    assert cnt.pushes > 10000 and cnt.max_stack >= 4
    check_exact(gpu_frame(R, sc, tex, sky, w, h, depth, True, cam=cam), want, "glass field")
    check(gpu_frame(R, sc, tex, sky, w, h, depth, False, cam=cam), want, 0.98, 0.99)


def test_many_spheres_wide_counts_and_global_geometry_path(R, oracle, tex, sky):
    """Config C4's scene shrunk (40x40 = 1600 spheres > 255 -> 4-byte counts; > 1024 float4 of
    geometry -> read from global memory instead of LDS)."""
    from example_gui_opencl_raytracer_amd import scene
    sc = scene.sphere_grid_scene(40, 40)
    cam = dict(origin=(0.0, 6.0, -6.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
    w, h, depth = 96, 54, 2
    want, _, _ = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky, depth)
    check_exact(gpu_frame(R, sc, tex, sky, w, h, depth, True, cam=cam), want, "1600 spheres")
    check(gpu_frame(R, sc, tex, sky, w, h, depth, False, cam=cam), want, 0.99, 0.995)


GRID_CAMS = [dict(origin=(0.0, 6.0, -6.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0),        # above, looking across
             dict(origin=(0.5, 0.3, 10.5), look=(1.0, 0.0, 0.0), fov=90.0, focal=1.0),          # INSIDE the grid, axis-parallel rays
             dict(origin=(3.0, 25.0, 12.0), look=(0.0, -1.0, 0.001), fov=70.0, focal=1.0),      # straight down
             dict(origin=(-30.0, 0.31, 12.0), look=(1.0, 0.0, 0.0), fov=40.0, focal=1.0)]       # grazing along the rows


@pytest.mark.parametrize("cam", GRID_CAMS)
@pytest.mark.parametrize("kind", ["opaque", "glass", "mixed"])
def test_uniform_grid_equals_linear_scan(R, oracle, tex, sky, cam, kind):
    """Scenes with > 256 spheres go through the uniform grid; the image must be bit-identical to the linear
    scan (same arithmetic per test, same tie rule, transparent spheres counted once), and match the oracle."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sc = scene.sphere_grid_scene(24, 24)                       # 576 spheres, r 0.3, pitch 1
    if kind != "opaque":
        g = scene.glass()
        sel = slice(None) if kind == "glass" else slice(0, None, 3)
        for name in ("ambient", "diffuse", "specular", "shininess", "transperent", "dielectric", "n", "reflectivity"):
            sc.spheres["material"][name][sel] = g[name]
        sc.spheres["radius"][sel] = 0.55                       # overlapping neighbours: spheres span several cells
        sc = scene.Scene(sc.spheres, sc.planes, sc.lights)
    w, h, depth = 160, 100, 4
    outs = {}
    for strict in (True, False):
        for grid in (1, 0):
            r = Renderer(sc, tex, sky, w, h, depth=depth, strict=strict)
            r.w.set_grid(grid)
            r.look(**cam)
            outs[(strict, grid)] = r.render()
            r.release()
        assert np.array_equal(outs[(strict, 1)], outs[(strict, 0)])
    want, _, _ = oracle.render(oracle.camera(cam["origin"], cam["look"], cam["fov"], 1.0, w, h), sc, tex, sky, depth)
    check_exact(outs[(True, 1)], want, f"grid {kind}")


@pytest.mark.parametrize("kind", ["opaque", "mixed"])
def test_light_spheres_seen_through_the_grid(R, oracle, tex, sky, kind):
    """findLightIntersection's visibility check (is an opaque sphere met before the light sphere? primitives.cl:296-318)
    walks the grid in the big-scene build.  Big light spheres above, at the edge of and INSIDE a field of 576 spheres, seen
    from above and grazing along the rows (where field spheres hide them, transparent ones must not): grid == linear scan
    in both builds, strict == oracle, and the lights really are in view."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sc = scene.sphere_grid_scene(24, 24)
    sp, li = sc.spheres.copy(), sc.lights.copy()
    if kind == "mixed":
        g = scene.glass()
        for name in ("ambient", "diffuse", "specular", "shininess", "transperent", "dielectric", "n", "reflectivity"):
            sp["material"][name][::2] = g[name]
    li["origin"][0] = (0.0, 3.0, 12.0);  li["radius"][0] = 1.5
    li["origin"][1] = (-11.0, 0.4, 12.0); li["radius"][1] = 0.9       # at the edge, half hidden by the first rows
    li["origin"][2] = (0.5, 0.3, 10.5);  li["radius"][2] = 0.19       # between four spheres of the field
    sc = scene.Scene(sp, sc.planes, li)
    w, h, depth = 160, 100, 3
    for cam in (GRID_CAMS[0], GRID_CAMS[3], dict(origin=(0.5, 0.3, -4.0), look=(0.0, 0.0, 1.0), fov=50.0, focal=1.0)):
        want, _, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], cam["fov"], 1.0, w, h), sc, tex, sky, depth)
        assert cnt.light_probes - cnt.segments > 50          # probes that ended ON a light sphere (no solid search after them)
        outs = {}
        for strict in (True, False):
            for grid in (1, 0):
                r = Renderer(sc, tex, sky, w, h, depth=depth, strict=strict)
                r.w.set_grid(grid)
                r.look(**cam)
                outs[(strict, grid)] = r.render()
                r.release()
            assert np.array_equal(outs[(strict, 1)], outs[(strict, 0)]), (kind, cam, strict)
        check_exact(outs[(True, 1)], want, f"lights through the grid, {kind}")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_sphere_cloud_through_the_grid(R, oracle, tex, sky, seed):
    """600 spheres scattered in a box (not a flat field): radii over a decade, a third of them glass, many overlapping, two
    of the lights INSIDE the cloud, the camera inside it too for one view.  Grid == linear scan in both builds and strict ==
    oracle: the per-axis bounds, the cell lists of big spheres, the paired entry loads and the light-visibility walk."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    rng = np.random.default_rng(seed)
    base = scene.sphere_grid_scene(25, 24)                      # 600 plastic spheres: overwrite the geometry
    sp, li = base.spheres.copy(), base.lights.copy()
    n = len(sp)
    sp["origin"][:, 0] = rng.uniform(-9.0, 9.0, n).astype(np.float32)
    sp["origin"][:, 1] = rng.uniform(0.2, 7.0, n).astype(np.float32)
    sp["origin"][:, 2] = rng.uniform(2.0, 20.0, n).astype(np.float32)
    sp["radius"] = (0.12 * 10.0 ** rng.uniform(0.0, 1.0, n)).astype(np.float32)      # 0.12 .. 1.2
    g = scene.glass()
    sel = rng.random(n) < 0.33
    for name in ("ambient", "diffuse", "specular", "shininess", "transperent", "dielectric", "n", "reflectivity"):
        sp["material"][name][sel] = g[name]
    li["origin"][0] = (0.0, 3.5, 11.0);  li["radius"][0] = 0.4
    li["origin"][1] = (-4.0, 1.5, 6.0);  li["radius"][1] = 0.25
    li["origin"][2] = (3.0, 9.0, 4.0)
    sc = scene.Scene(sp, base.planes, li)
    w, h, depth = 128, 96, 4
    cams = [dict(origin=(0.0, 4.0, -6.0), look=(0.0, -0.1, 1.0), fov=80.0, focal=1.0),
            dict(origin=(1.0, 3.0, 10.0), look=(-0.3, 0.1, 1.0), fov=100.0, focal=1.0)]          # inside the cloud
    for cam in cams:
        want, _, _ = oracle.render(oracle.camera(cam["origin"], cam["look"], cam["fov"], 1.0, w, h), sc, tex, sky, depth)
        outs = {}
        for strict in (True, False):
            for grid in (1, 0):
                r = Renderer(sc, tex, sky, w, h, depth=depth, strict=strict)
                r.w.set_grid(grid)
                r.look(**cam)
                outs[(strict, grid)] = r.render()
                r.release()
            assert np.array_equal(outs[(strict, 1)], outs[(strict, 0)]), (seed, cam, strict)
        check_exact(outs[(True, 1)], want, f"sphere cloud {seed}")


def test_nan_rays_on_the_grid_path_end_like_the_linear_scan(R, oracle, tex, sky):
    """A scene of more than 256 spheres (uniform-grid build) whose rays go NaN: the camera sits at the exact centre
    of a glass sphere, so view and light directions cancel, normalize(0) = NaN poisons radiance and directions, and
    the next segments are traced with NaN rays.  The grid walk must end (it has no cells to walk for such a ray and
    takes the reference's in-order scan instead) and give the same pixels as the linear-scan build and the oracle."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sc = scene.sphere_grid_scene(24, 24)
    g = scene.glass()
    for name in ("ambient", "diffuse", "specular", "shininess", "transperent", "dielectric", "n", "reflectivity"):
        sc.spheres["material"][name][:] = g[name]
    sc.spheres["radius"][:] = 0.55
    sc = scene.Scene(sc.spheres, sc.planes, sc.lights)
    c = sc.spheres["origin"][300]
    cam = dict(origin=(float(c[0]), float(c[1]), float(c[2])), look=(0.3, -0.2, 1.0), fov=100.0, focal=1.0)
    w, h, depth = 96, 64, 6
    want, rgb, _ = oracle.render(oracle.camera(cam["origin"], cam["look"], cam["fov"], 1.0, w, h), sc, tex, sky, depth, want_rgb=True)
    assert np.isnan(rgb).any()
    outs = {}
    for grid in (1, 0):
        r = Renderer(sc, tex, sky, w, h, depth=depth, strict=True)
        r.w.set_grid(grid)
        r.look(**cam)
        outs[grid] = r.render()
        r.release()
    assert np.array_equal(outs[1], outs[0])
    check_exact(outs[1], want, "NaN rays on the grid path")


def test_lds_and_global_geometry_paths_agree_exactly(R, demo_scene, tex, sky):
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    outs = []
    for variant in (0, 1, 2, 3):               # bit 0: geometry from global memory, bit 1: linear id mapping
        r = Renderer(demo_scene, tex, sky, 160, 120, depth=15, strict=True)
        r.w.set_variant(variant)
        r.look(**CAM)
        outs.append(r.render())
        r.release()
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


@pytest.mark.parametrize("strict", [True, False])
def test_plane_test_skipping_is_pure_work_skipping(R, demo_scene, tex, sky, strict):
    """The light / plane side table lets a wave skip plane tests whose outcome is certain (scene_prep.c, wt_shadow_batch).
    Same bits with the table off (variant 128) -- on render.map, where every such test is skipped, and on scenes whose
    lights sit on both sides of, inside and far behind the planes."""
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    from example_gui_opencl_raytracer_amd.scene import Scene
    from fuzz_scenes import random_scene
    cases = [(demo_scene, CAM, 15)]
    lights = demo_scene.lights.copy()
    lights["origin"][0] = (-2.0, 3.0, 7.05)          # light 0 straddles the mirror wall (z = 7): never skippable
    lights["origin"][1] = (2.0, -1.5, 0.2)           # light 1 below the floor: the floor is between it and everything
    lights["origin"][2] = (1.0, 0.1001, 3.0)         # light 2 (r = 0.1) a hair above the floor
    cases.append((Scene(demo_scene.spheres, demo_scene.planes, lights), CAM, 4))
    for seed in (1, 5, 9, 14, 23):
        sc, cam, depth = random_scene(seed)
        cases.append((sc, cam, depth))
    for sc, cam, depth in cases:
        outs = []
        for variant in (0, 128):
            r = Renderer(sc, tex, sky, 200, 120, depth=depth, strict=strict)
            r.w.set_variant(variant)
            r.look(**cam)
            outs.append(r.render())
            r.release()
        assert np.array_equal(outs[0], outs[1])


def _vis_cases(demo_scene):
    """Scenes for the visibility-class tests: render.map under its two driver cameras and close-ups of the shadow edges, lights
    moved next to / inside / behind spheres and planes, glass in front of lights, and fuzz scenes."""
    from example_gui_opencl_raytracer_amd.scene import Scene
    from fuzz_scenes import random_scene
    cases = [(demo_scene, CAM, 4), (demo_scene, dict(origin=(0.8, 2.5, -8.0), look=(0.0, 0.0, 1.0), fov=90.0, focal=1.0), 15),
             (demo_scene, dict(origin=(-3.0, 0.6, 0.5), look=(1.0, 0.05, 0.3), fov=70.0, focal=1.0), 4),
             (demo_scene, dict(origin=(1.0, 9.0, 1.0), look=(0.01, -1.0, 0.02), fov=60.0, focal=1.0), 4),
             (demo_scene, dict(origin=(0.9, 0.7, 1.4), look=(0.3, -0.2, 1.0), fov=100.0, focal=0.5), 8)]       # inside glass sphere #2
    lights = demo_scene.lights.copy()
    lights["origin"][0] = (4.5, 1.12, -1.0)          # light 0 a hair above the red plastic sphere (c = (4.5, 0.5, -1), r = 0.5)
    lights["origin"][1] = (0.8, 0.8, 1.5)            # light 1 INSIDE glass sphere #2
    lights["origin"][2] = (-1.0, 1.0, 6.5)           # light 2 behind the blue sphere, next to the mirror wall
    lights["radius"][2] = 0.45                       # ... and big
    cases.append((Scene(demo_scene.spheres, demo_scene.planes, lights), CAM, 4))
    lights = demo_scene.lights.copy()
    lights["radius"][:] = (0.0, 1.5, 0.02)           # a point light, a huge one, a tiny one
    cases.append((Scene(demo_scene.spheres, demo_scene.planes, lights), CAM, 4))
    for seed in (0, 1, 2, 3, 5, 8, 9, 13, 14, 21, 23, 34):
        cases.append(random_scene(seed))
    return cases


def test_light_visibility_classes_are_pure_work_skipping(R, demo_scene, tex, sky):
    """wt_light_vis (strict build, small scenes) decides the shadow samples of a light without tracing them when every ray of the
    light's cone has the same outcome.  (1) Same bits with the classes off (variant 256).  (2) The counting build in verification
    mode (variant 1024) classifies AND traces: not one classified light may have a traced factor that differs from its class's."""
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    classified = 0
    strict = True
    for sc, cam, depth in _vis_cases(demo_scene):
        outs = []
        for variant in (0, 256):
            r = Renderer(sc, tex, sky, 240, 160, depth=depth, strict=strict)
            r.w.set_variant(variant)
            r.look(**cam)
            outs.append(r.render())
            r.release()
        assert np.array_equal(outs[0], outs[1])
        r = Renderer(sc, tex, sky, 240, 160, depth=depth, strict=strict)
        r.w.set_variant(1024)
        r.w.enable_counters(1)
        r.look(**cam)
        chk = r.render()
        c = r.w.read_counters()
        r.release()
        assert np.array_equal(chk, outs[0])
        assert c["vis_mismatches"] == 0, c
        classified += c["lights_classified"]
    assert classified > 100000
    report(dict(test="visibility classes verified", strict=strict, lights_classified=classified, mismatches=0))


def test_light_visibility_classes_verified_at_full_c2_size(R, demo_scene, tex):
    """The same verification on the bench workload itself (1920x1080, depth 4) and on the reference's own 800x600 depth 15."""
    from example_gui_opencl_raytracer_amd import textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sky4k = textures.skybox_cross(4096)
    for (w, h, depth) in ((1920, 1080, 4), (800, 600, 15)):
        for strict in (True,):
            r = Renderer(demo_scene, tex, sky4k, w, h, depth=depth, strict=strict)
            r.w.set_variant(1024)
            r.w.enable_counters(1)
            r.look(**CAM)
            r.render(readback=False)
            c = r.w.read_counters()
            r.release()
            report(dict(test="visibility classes", frame=f"{w}x{h} d{depth}", strict=strict, shadow_rays=c["shadow_rays"],
                        lights_classified=c["lights_classified"], mismatches=c["vis_mismatches"]))
            assert c["vis_mismatches"] == 0 and 2 * c["lights_classified"] > 0.5 * c["shadow_rays"], c


def test_cost_sorted_dispatch_is_pure_scheduling(R, demo_scene, tex, sky):
    """Frame 1 runs in the default tile order and records tile costs; frames 2+ serve each XCD's tiles
    heaviest-first.  Same bits in every case, also after the camera moves and with scheduling off."""
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    w, h = 328, 203                                   # ragged: 41 tile columns, 26 tile rows (25.4)
    cams = [CAM, dict(origin=(1.5, 2.0, -6.0), look=(-0.1, -0.1, 1.0), fov=90.0, focal=1.0)]
    ref = []
    off = Renderer(demo_scene, tex, sky, w, h, depth=15, strict=True)
    off.w.set_tile_sched(0)
    for cam in cams:
        off.look(**cam)
        ref.append(off.render())
    off.release()
    on = Renderer(demo_scene, tex, sky, w, h, depth=15, strict=True)
    for cam, want in zip(cams + cams, ref + ref):
        on.look(**cam)
        for _ in range(3):
            assert np.array_equal(on.render(), want)
    on.release()


# ------------------------------------------------------------ edge cases
@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (33, 9), (150, 101), (257, 8)])
def test_ragged_sizes(R, oracle, demo_scene, tex, sky, w, h):
    want, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), demo_scene, tex, sky, 4)
    got = gpu_frame(R, demo_scene, tex, sky, w, h, 4, True)
    check_exact(got, want, f"ragged {w}x{h}")


def test_empty_primitive_lists(R, oracle, demo_scene, tex, sky):
    from example_gui_opencl_raytracer_amd.scene import Scene
    w, h = 96, 64
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    cases = {
        "no spheres": Scene(demo_scene.spheres[:0], demo_scene.planes, demo_scene.lights),
        "no planes": Scene(demo_scene.spheres, demo_scene.planes[:0], demo_scene.lights),
        "no lights": Scene(demo_scene.spheres, demo_scene.planes, demo_scene.lights[:0]),
        "sky only": Scene(demo_scene.spheres[:0], demo_scene.planes[:0], demo_scene.lights[:0]),
    }
    for name, sc in cases.items():
        want, _, _ = oracle.render(cam, sc, tex, sky, 4)
        got = gpu_frame(R, sc, tex, sky, w, h, 4, True)
        check_exact(got, want, name)


@pytest.mark.parametrize("nl", [1, 2, 4, 7])
def test_light_counts_around_the_chunk_size(R, oracle, demo_scene, tex, sky, nl):
    """Lights are shaded three at a time (six shadow rays together): counts that are not multiples of three
    exercise the partial chunk; the RNG stream must still be drawn light by light in the reference's order."""
    from example_gui_opencl_raytracer_amd.scene import Scene, LIGHT
    lights = np.zeros(nl, LIGHT)
    base = demo_scene.lights
    for i in range(nl):
        lights[i] = base[i % 3]
        lights[i]["origin"] = base[i % 3]["origin"] + np.float32(0.7 * (i // 3)) * np.array([1.0, 0.3, -0.5], np.float32)
        lights[i]["intensity"] = base[i % 3]["intensity"] / np.float32(1 + i // 3)
    sc = Scene(demo_scene.spheres, demo_scene.planes, lights)
    w, h = 128, 96
    want, _, cnt = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), sc, tex, sky, 6)
    assert cnt.shadow_rays == 2 * nl * cnt.shaded_hits
    check_exact(gpu_frame(R, sc, tex, sky, w, h, 6, True), want, f"{nl} lights d6")       # depth 6: deep build, sparse-tail loop too
    check_exact(gpu_frame(R, sc, tex, sky, w, h, 4, True), oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), sc, tex, sky, 4)[0], f"{nl} lights d4")


def test_maximum_one_byte_counts(R, oracle, tex, sky):
    """255 spheres, 255 planes... would be absurdly slow on the oracle; 255 spheres + 3 planes + 255 lights is the
    largest scene the reference's one-byte counts can express in the dimensions that matter (raytracing.cl:17)."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.scene import Scene, LIGHT, PLANE
    big = scene.sphere_grid_scene(17, 15)                       # 255 spheres
    assert len(big.spheres) == 255
    planes = np.zeros(3, PLANE)
    planes[0] = big.planes[0]
    planes[1] = scene.render_map_scene().planes[1]
    planes[2] = scene.render_map_scene().planes[1]
    planes[2]["normal"] = (1.0, 0.0, 0.0); planes[2]["point_in_plane"] = (-12.0, 0.0, 0.0)
    lights = np.zeros(255, LIGHT)
    rng = np.random.default_rng(5)
    lights["origin"] = rng.uniform([-8, 1.5, 0], [8, 6, 14], (255, 3)).astype(np.float32)
    lights["radius"] = 0.1
    lights["intensity"] = 0.6
    lights["rgb"] = rng.uniform(0.2, 1.0, (255, 3)).astype(np.float32)
    sc = Scene(big.spheres, planes, lights)
    cam = dict(origin=(0.0, 6.0, -6.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
    w, h = 48, 32
    want, _, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky, 2)
    assert cnt.shadow_rays == 510 * cnt.shaded_hits
    check_exact(gpu_frame(R, sc, tex, sky, w, h, 2, True, cam=cam, wide_counts=False), want, "255 spheres / 255 lights")


def test_odd_image_sizes(R, oracle, demo_scene):
    """Texture layers that are not 256x256 / not a power of two, a skybox whose width is not a multiple of 4
    (face = width / 4 truncates, primitives.cl:14 via raytracing.cl:62)."""
    from example_gui_opencl_raytracer_amd.scene import Scene
    rng = np.random.default_rng(9)
    tex = rng.integers(0, 256, (3, 60, 100, 4), dtype=np.uint8); tex[..., 3] = 255
    sky = rng.integers(0, 256, (1, 375, 501, 4), dtype=np.uint8); sky[..., 3] = 255
    planes = demo_scene.planes.copy()
    planes["material"]["texture_id"][0] = 1
    planes["material"]["texture_scale"][0] = 23.5
    planes["material"]["texture_id"][1] = 2                   # the mirror wall gets a texture as well
    planes["material"]["texture_scale"][1] = 7.0
    sc = Scene(demo_scene.spheres, planes, demo_scene.lights)
    w, h = 160, 120
    want, _, cnt = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), sc, tex, sky, 4)
    assert cnt.texel_fetches > 0 and cnt.sky_fetches > 0
    got = gpu_frame(R, sc, tex, sky, w, h, 4, True)
    if cnt.oob_reads == 0 and cnt.int_cast_oor == 0:
        check_exact(got, want, "odd image sizes")
    else:                                                       # undefined reads in the reference: both sides clamp to the edge
        check(got, want, 0.995)


def test_count_width_is_irrelevant(R, demo_scene, tex, sky):
    a = gpu_frame(R, demo_scene, tex, sky, 96, 64, 4, True, wide_counts=False)     # uchar counts (reference)
    b = gpu_frame(R, demo_scene, tex, sky, 96, 64, 4, True, wide_counts=True)      # 4-byte counts (extension)
    assert np.array_equal(a, b)


def test_pixel_zero_has_the_stuck_rng(R, oracle, demo_scene, tex, sky):
    """id 0 seeds xorshift with 0, its fixed point (raytracing.cl:33): both sides must agree there."""
    w, h = 64, 48
    want, _, _ = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), demo_scene, tex, sky, 4)
    got = gpu_frame(R, demo_scene, tex, sky, w, h, 4, True)
    assert got[0] == want[0]


# ------------------------------------------------------------ BASELINE.json sizes: properties + counters
def test_full_size_c2_against_the_oracle_and_ray_count(R, oracle, demo_scene, tex):
    """Config C2 (1920x1080, depth 4, 4096x3072 skybox) -- the bench workload itself."""
    from example_gui_opencl_raytracer_amd import textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sky4k = textures.skybox_cross(4096)
    w, h, depth = 1920, 1080, 4
    want, _, cnt = oracle.render(oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), demo_scene, tex, sky4k, depth)
    assert 13.8 < cnt.rays / (w * h) < 13.95                      # 13.88 rays/px
    for strict in (True, False):
        r = Renderer(demo_scene, tex, sky4k, w, h, depth=depth, strict=strict)
        r.look(**CAM)
        got = r.render()
        if strict:
            pin_strict_residual(r.w, got, want, oracle, oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h), demo_scene, tex, sky4k, depth, "C2 strict")
        else:
            ex = check(got, want, 0.999, 0.9995)
            report(dict(test="C2 fast", exact=ex, within_1lsb=float((channel_diff(got, want) <= 1).mean())))
        r.w.enable_counters(1)
        r.render(readback=False)
        c = r.w.read_counters()
        r.release()
        rays = c["segments"] + c["shadow_rays"]
        assert abs(rays - cnt.rays) <= (0 if strict else 2e-4 * cnt.rays)
        # shadow rays really traced: the ones of zero-coefficient (glass) surfaces are drawn from the RNG but elided, and so are
        # (strict build) those of lights whose visibility class decides their samples (wt_light_vis: 97 % of the rest on this scene)
        assert 0 < c["shadow_rays_traced"] <= c["shadow_rays"] - 2 * c["lights_classified"]
        if strict:
            assert 2 * c["lights_classified"] > 0.6 * c["shadow_rays"]
        else:
            assert c["lights_classified"] == 0 and c["shadow_rays_traced"] > 0.5 * c["shadow_rays"]
        if strict:
            assert (c["segments"], c["shadow_rays"], c["light_probes"], c["sky_fetches"]) == \
                   (cnt.segments, cnt.shadow_rays, cnt.light_probes, cnt.sky_fetches)
            # the reference fetches a texel for every plane that improves t (primitives.cl:374-378); the
            # kernel fetches it for the winning plane only -- same image, fewer fetches
            assert 0 < c["texel_fetches"] <= cnt.texel_fetches


def test_large_frame_is_deterministic_and_tile_order_free(R, tex, sky):
    """4096x4096 (config C3's frame size): two renders are identical, and equal the linear-mapping build (variant 2)
    and the low-occupancy flavour of the deep build (variant 64: a launch of this size otherwise takes the flavour with
    two DFS levels in LDS and registers capped for five waves per SIMD)."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sc = scene.dielectric_field_scene(8)
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
    outs = []
    # (2048: the scratch part of the DFS stack sized for CLW_MAX_DEPTH instead of for this depth)
    for variant in (0, 0, 2, 64, 2048):
        r = Renderer(sc, tex, sky, 4096, 4096, depth=8, strict=False)
        r.w.set_variant(variant)
        r.look(**cam)
        outs.append(r.render())
        r.release()
    assert all(np.array_equal(outs[0], o) for o in outs[1:])
    assert len(np.unique(outs[0])) > 1000


@pytest.mark.parametrize("strict", [True, False])
def test_deep_refraction_trees_on_the_glass_field(R, oracle, tex, sky, strict):
    """The DFS stack at work: glass field at depth 15 (302 rays per pixel, stacks 15 deep) and, on a smaller frame, depth 20 (1 750 rays per
    pixel; at depth 32 the refraction trees of this scene explode).  The ordinary launch, the low-occupancy flavour of the deep build
    (variant 64: three stack levels in LDS instead of one) and the full-depth scratch stack (variant 2048) give the same bits; the strict
    build equals the oracle."""
    from example_gui_opencl_raytracer_amd import scene
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sc = scene.dielectric_field_scene(8)
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
    for depth, w, h in ((15, 192, 160), (20, 96, 80)):
        outs = []
        for variant in (0, 64, 2048, 64 | 2048):
            r = Renderer(sc, tex, sky, w, h, depth=depth, strict=strict)
            r.w.set_variant(variant)
            r.look(**cam)
            outs.append(r.render())
            r.release()
        assert all(np.array_equal(outs[0], o) for o in outs[1:])
        if strict:
            want, _, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky, depth)
            assert cnt.max_stack >= 14
            check_exact(outs[0], want, f"glass field depth {depth}")


# ------------------------------------------------------------ the tree-parallel tail of deep launches (csrc/whitted_tpt.inc)
TAIL_KEYS = ("segments", "shadow_rays", "light_probes", "sky_fetches", "texel_fetches", "pushes", "shadow_rays_traced")


def _tail_frame(r, max_lanes, frames=3):
    """One counted frame with the tail entered at <= max_lanes live lanes (0: variant 16, the per-lane loop runs to the end).  Frames before
    it let the dispatch order follow the tiles' costs, so that heavy tiles are served by several wavefronts when the tail is on."""
    r.w.set_tpt(max_lanes, 1 if max_lanes == 64 else -1, -1)
    r.w.set_variant(16 if max_lanes == 0 else 0)
    for _ in range(frames - 1):
        r.render(readback=False)
    r.w.enable_counters(1)
    img = r.render()
    c = r.w.read_counters()
    r.w.enable_counters(0)
    return img, c


@pytest.mark.parametrize("strict", [True, False])
@pytest.mark.parametrize("case", ["render.map d15", "render.map d8", "glass field d8", "glass field d15"])
def test_tree_parallel_tail_equals_the_per_lane_loop(R, oracle, demo_scene, tex, sky, strict, case):
    """Deep launches finish their tiles in the tree-parallel tail: pending paths become a pool of nodes traced by the whole wave, depth-first
    positions and xorshift states come from subtree sizes, the lights' factors are added in a per-pixel replay.  Whatever share of a frame goes
    through it -- entered at <= 8 or <= 24 live lanes (lanes arrive with stacks of continuations), or from the first segment on (64:
    EVERYTHING through the tail), heavy tiles split over several wavefronts or not (variant 4096) -- pixels and ray counters equal the
    per-lane loop's (variant 16), in both builds; the strict frames equal the oracle's."""
    from example_gui_opencl_raytracer_amd import scene
    glass_cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
    sc, w, h, depth, cam = {"render.map d15": (demo_scene, 320, 240, 15, CAM), "render.map d8": (demo_scene, 200, 152, 8, CAM),
                            "glass field d8": (scene.dielectric_field_scene(8), 256, 256, 8, glass_cam),
                            "glass field d15": (scene.dielectric_field_scene(4), 160, 160, 15, glass_cam)}[case]
    r = R(sc, tex, sky, w, h, depth=depth, strict=strict)
    r.look(**cam)
    ref, cref = _tail_frame(r, 0)
    assert cref["tpt_tiles"] == 0
    for max_lanes in (8, 24, 64):
        img, c = _tail_frame(r, max_lanes)
        assert c["tpt_tiles"] > 0 and c["tpt_nodes"] > 0, "the tail did not run"
        assert np.array_equal(img, ref), f"{case}: tail entered at <= {max_lanes} lanes: {int((img != ref).sum())} pixels differ from the per-lane loop"
        assert all(c[k] == cref[k] for k in TAIL_KEYS), (max_lanes, {k: (c[k], cref[k]) for k in TAIL_KEYS})
    r.w.set_variant(4096)                                   # heavy tiles not split
    r.w.set_tpt(24, -1, -1)
    for _ in range(2):
        r.render(readback=False)
    assert np.array_equal(r.render(), ref)
    r.release()
    if strict:
        want, _, _ = oracle.render(oracle.camera(cam["origin"], cam["look"], cam.get("fov", 90.0), cam.get("focal", 1.0), w, h), sc, tex, sky, depth)
        check_exact(ref, want, f"tail / per-lane loop vs oracle: {case}")


@pytest.mark.parametrize("strict", [True, False])
def test_tree_parallel_tail_gives_up_cleanly_when_its_slot_is_full(R, tex, sky, strict):
    """A pool so small that a slot holds 192 nodes: the trees of the glass field at depth 15 outgrow it, the tail gives up on those tiles --
    before it has touched any lane state -- and the per-lane loop finishes them: same pixels, same counters, and both paths were taken."""
    from example_gui_opencl_raytracer_amd import scene
    sc = scene.dielectric_field_scene(4)
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
    r = R(sc, tex, sky, 160, 160, depth=15, strict=strict)
    r.look(**cam)
    ref, cref = _tail_frame(r, 0)
    per_node = (29 + 4 * 3) if strict else (27 + 3)
    words = (25 + 36) * 64 + 15 * 192 + per_node * 192 + 63
    r.w.set_tpt(-1, -1, max(1, (words * 8192 * 4) >> 20))
    for max_lanes in (24, 64):
        img, c = _tail_frame(r, max_lanes)
        assert c["tpt_gave_up"] > 0 and c["tpt_tiles"] > 0, c
        assert np.array_equal(img, ref)
        assert all(c[k] == cref[k] for k in TAIL_KEYS)
    r.release()


def test_tree_parallel_tail_with_many_lights_and_on_the_grid(R, oracle, tex, sky):
    """The tail's node carries one weight per light (strict: four): seven lights (more than the replay preloads), and a 576-sphere scene
    whose segments walk the uniform grid, at depth 8 -- strict == oracle with everything through the tail."""
    from example_gui_opencl_raytracer_amd import scene
    sc = scene.sphere_grid_scene(24, 24)
    cam = dict(origin=(0.0, 6.0, -8.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
    base = scene.render_map_scene()
    lights = np.concatenate([base.lights, base.lights, base.lights[:1]])
    for k in range(len(lights)):
        lights["origin"][k, :3] += np.float32(0.37 * k)
    many = scene.Scene(base.spheres, base.planes, lights)
    for sc_, cam_, w, h in ((many, CAM, 160, 120), (sc, cam, 160, 96)):
        want, _, _ = oracle.render(oracle.camera(cam_["origin"], cam_["look"], cam_.get("fov", 90.0), cam_.get("focal", 1.0), w, h), sc_, tex, sky, 8)
        r = R(sc_, tex, sky, w, h, depth=8, strict=True)
        r.look(**cam_)
        for max_lanes in (16, 64):
            img, c = _tail_frame(r, max_lanes)
            assert c["tpt_tiles"] > 0
            check_exact(img, want, f"tail, {len(sc_.lights)} lights, {len(sc_.spheres)} spheres, entered at <= {max_lanes}")
        r.release()


# ------------------------------------------------------------ the other BASELINE.json configurations at FULL size
def test_full_size_c3_glass_field_against_the_oracle(R, oracle, tex):
    """Config C3: 4096x4096, depth 8, 64 dielectric spheres.  493 M rays: the oracle needs the GPU box's host cores."""
    from example_gui_opencl_raytracer_amd import scene, textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sky4k = textures.skybox_cross(4096)
    sc = scene.dielectric_field_scene(8)
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
    w = h = 4096
    want, _, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky4k, 8)
    assert 29.0 < cnt.rays / (w * h) < 30.0
    r = Renderer(sc, tex, sky4k, w, h, depth=8, strict=True)
    r.look(**cam)
    pin_strict_residual(r.w, r.render(), want, oracle, oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky4k, 8, "C3 strict")
    r.w.enable_counters(1); r.render(readback=False); c = r.w.read_counters(); r.release()
    assert c["segments"] + c["shadow_rays"] == cnt.rays
    assert c["shadow_rays_traced"] < 0.5 * c["shadow_rays"]       # a field of glass: most counted shadow rays are elided
    r = Renderer(sc, tex, sky4k, w, h, depth=8, strict=False)
    r.look(**cam)
    got = r.render()
    report(dict(test="C3 fast", exact=check(got, want, 0.99, 0.995), within_1lsb=float((channel_diff(got, want) <= 1).mean())))
    r.release()


def test_full_size_c4_ten_thousand_spheres_against_the_oracle(R, oracle, tex):
    """Config C4: 100x100 opaque spheres, 1920x1080, depth 4 -- the uniform-grid path against the oracle's
    brute-force scan (2.7e11 sphere tests on the host cores)."""
    from example_gui_opencl_raytracer_amd import scene, textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    sky4k = textures.skybox_cross(4096)
    sc = scene.sphere_grid_scene(100, 100)
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
    w, h = 1920, 1080
    want, _, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky4k, 4)
    r = Renderer(sc, tex, sky4k, w, h, depth=4, strict=True)
    r.look(**cam)
    pin_strict_residual(r.w, r.render(), want, oracle, oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky4k, 4, "C4 strict")
    r.w.enable_counters(1); r.render(readback=False); c = r.w.read_counters(); r.release()
    assert c["segments"] + c["shadow_rays"] == cnt.rays
    # opaque plastic everywhere: only the rays to a light BEHIND the surface (diffuse and specular terms exactly zero) are elided
    assert 0.9 * c["shadow_rays"] < c["shadow_rays_traced"] < c["shadow_rays"]
    r = Renderer(sc, tex, sky4k, w, h, depth=4, strict=False)      # the benchmarked build at the full size
    r.look(**cam)
    got = r.render()
    # This scene's own noise floor is low: shadow and reflection rays start up to 100 units from the spheres they pass, so
    # b*b and 4ac of intersect_sphere (primitives.cl:181) agree to 5-7 digits and the sign of their difference is rounding
    # noise over much of a sphere's cross-section.  The bar is what the scene itself predicts: the oracle built WITH FMA
    # contraction (a second legal build of the same source, liboracle_fma.so) against the oracle -- the fast build must be
    # no further from the oracle than that build is (measured: 96.6 % vs 96.1-96.5 %); the strict build reproduces the oracle's
    # rounding exactly (above).  The discontinuity-mask form of the claim is test_fast_outliers_lie_on_the_discontinuity_mask_c3_c4.
    import os
    from oracle.oracle_py import HERE, Oracle
    fma, _, _ = Oracle(os.path.join(HERE, "liboracle_fma.so")).render(oracle.camera(cam["origin"], cam["look"], 90.0, 1.0, w, h), sc, tex, sky4k, 4)
    dfma = channel_diff(fma, want)
    floor_exact, floor_le1 = float((dfma == 0).mean()), float((dfma <= 1).mean())
    ex = check(got, want, floor_exact - 0.005, floor_le1 - 0.005)
    report(dict(test="C4 fast", exact=ex, within_1lsb=float((channel_diff(got, want) <= 1).mean()),
                oracle_fma_exact=floor_exact, oracle_fma_within_1lsb=floor_le1))
    r.release()


def test_full_size_c5_strips_equal_the_single_gpu_frame(R, oracle, demo_scene, tex):
    """Config C5: 8192x8192, depth 4, eight row strips.  Size-independent property at the full size: every strip is
    bit-identical to its rows of the one-GPU frame; two strips are also checked against the oracle."""
    from example_gui_opencl_raytracer_amd import textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer, strip_rows
    sky4k = textures.skybox_cross(4096)
    w = h = 8192
    full = Renderer(demo_scene, tex, sky4k, w, h, depth=4, strict=True)
    full.look(**CAM)
    want = full.render()
    full.release()
    import zlib
    cam = oracle.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
    for rank in range(8):
        r0, rows = strip_rows(h, 8, rank)
        assert rows == 1024
        s = Renderer(demo_scene, tex, sky4k, w, h, depth=4, strict=True, first_row=r0, rows=rows)
        s.look(**CAM)
        got = s.render()
        s.release()
        assert zlib.crc32(got.tobytes()) == zlib.crc32(want[r0 * w:(r0 + rows) * w].tobytes())
        if rank in (3, 5):
            ref, _, _ = oracle.render(cam, demo_scene, tex, sky4k, 4, id_begin=r0 * w, id_end=(r0 + 64) * w)
            check_exact(got[:64 * w], ref, f"C5 strip {rank}")


@pytest.mark.parametrize("seed", range(40))
def test_random_scenes_against_the_oracle(R, oracle, tex, sky, seed):
    """Fuzz: the scenes of tests/fuzz_scenes.py (0-8 spheres, 0-3 planes, 0-4 lights, any material, depth 1-15)."""
    from fuzz_scenes import random_scene
    sc, cam, depth = random_scene(seed)
    w, h = 72, 48
    want, rgb, cnt = oracle.render(oracle.camera(cam["origin"], cam["look"], cam["fov"], 1.0, w, h), sc, tex, sky, depth, want_rgb=True)
    if cnt.int_cast_oor or cnt.oob_reads:
        pytest.skip("scene hits an undefined float->int conversion / image read in the reference")
    got = gpu_frame(R, sc, tex, sky, w, h, depth, True, cam=cam)
    check_exact(got, want, f"fuzz seed {seed}")
    fast = gpu_frame(R, sc, tex, sky, w, h, depth, False, cam=cam)
    df = channel_diff(fast, want)
    assert (df <= 1).mean() >= 0.97, f"seed {seed}: fast build {(df <= 1).mean():.4f} within 1 LSB"
