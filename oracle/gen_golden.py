"""Generate tests/golden/ from the reference's own kernels compiled for the host.

TEST INFRASTRUCTURE ONLY.  Runs in the build container (needs oracle/_ref/libref_cl.so,
i.e. /root/reference); the GPU box only ever sees the resulting data files.

    python -m oracle.gen_golden

Outputs (inputs and expected outputs only -- no reference source in any form):
  tests/golden/frames.npz   packed 0x00RRGGBB frames of the reference `raygen`+`raytracer`
                            kernels for render.map (regenerated from scene_dump.c's values),
                            camera of raypng.c:17-21, procedural textures (textures.py),
                            skybox_cross(512): 160x120 at depth 1, 4, 15 and 320x240 at depth 4;
                            plus the 64-byte ray records of `raygen` for 160x120.
  tests/golden/masks.npz    per frame, bit-packed pixel masks of where the reference's output is DISCONTINUOUS or
                            ill-conditioned (make_masks below): `fma` = two legal builds of the reference kernels
                            (contraction off / on) disagree, dilated 1 px; `jitter` = the oracle's float radiance is
                            not locally linear under camera shifts of 1/64, 1/256, 1/1024 px; `margin` = a sphere
                            discriminant or a texel-index truncation is decided within rounding error.
  tests/golden/camera.npz   inputs and outputs of the reference's own rinit_camera + rgen_perspective
                            (src/cpu_ray.c:24-35, 42-106, built into oracle/_ref/libref_cpu_ray.so).
  tests/golden/vectors.npz  seeded random inputs and the reference's outputs for
                            intersect_sphere, intersect_plane, reflect, refract,
                            compute_schlick, map_to_cube, xorshift32, euclidean_modulo,
                            plane_texture_pixel, testShadowPath, findLightIntersection,
                            findSolidIntersection (reference src/cl/primitives.cl).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from example_gui_opencl_raytracer_amd import scene as S, textures as T  # noqa: E402
from oracle.oracle_py import (MARGIN_SITES, REF_FMA_SO, Oracle, RefCamera, Reference, _f3, _p, render_margins,  # noqa: E402
                              shifted_camera)

GOLD = os.path.join(ROOT, "tests", "golden")
CAM = dict(origin=(0.8, 2.5, -8.0), look=(0.2, 0.0, 1.0))
FRAMES = [(160, 120, 1), (160, 120, 4), (160, 120, 15), (320, 240, 4), (640, 480, 1)]   # the last one is BASELINE config C1
MASK_ONLY_FRAMES = [(1280, 720, 4)]   # too big to commit as pixels: mask + CRC of the reference frame
N = 2000

JITTERS = (1.0 / 64, 1.0 / 256, 1.0 / 1024)   # camera shifts in pixels
JITTER_TAU = 2e-5                              # |f(p+v) + f(p-v) - 2 f(p)| above this = not locally linear
DISC_MARGIN, CAST_MARGIN = 1e-5, 2e-6          # relative decision margins counted as "within rounding error"

# cameras for the rgen_perspective fixtures: the two drivers' own, then assorted positions / fields of view / sizes
CAMERAS = [((0.8, 2.5, -8.0), (0.2, 0.0, 1.0), 90.0, 1.0, 800, 600),        # raypng.c:17-21
           ((0.8, 2.5, -8.0), (0.0, 0.0, 1.0), 90.0, 1.0, 800, 600),        # rayinteractive.c:111-115
           ((0.8, 2.5, -8.0), (0.2, 0.0, 1.0), 90.0, 1.0, 1920, 1080),
           ((0.8, 2.5, -8.0), (0.2, 0.0, 1.0), 90.0, 1.0, 640, 480),
           ((-3.0, 0.6, 0.5), (1.0, 0.05, 0.3), 70.0, 1.0, 1280, 720),
           ((0.9, 0.7, 1.4), (0.3, -0.2, 1.0), 100.0, 0.5, 8192, 8192),
           ((1.0, 9.0, 1.0), (0.01, -1.0, 0.02), 60.0, 2.0, 4096, 4096),
           ((3.5, 3.0, -6.0), (0.0, -2.5, 9.5), 90.0, 1.0, 333, 77),
           ((0.0, 12.0, -10.0), (0.0, -0.45, 1.0), 120.0, 1.0, 1920, 1080),
           ((5.0, 1.0, 5.0), (-1.0, 0.0, -1.0), 35.0, 1.0, 160, 120),
           ((0.0, 0.0, 0.0), (0.0, 0.999, 0.04), 45.0, 1.0, 64, 48),         # almost straight up
           ((0.0, 0.0, 0.0), (0.0, -1.0, 0.0), 90.0, 1.0, 64, 48)]           # straight down (accepted: only +Y is rejected)


def channel_diff(a, b):
    ca = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], 1).astype(np.int32)
    cb = np.stack([(b >> 16) & 255, (b >> 8) & 255, b & 255], 1).astype(np.int32)
    return np.abs(ca - cb).max(1)


def dilate1(m, w, h):
    """3x3 dilation of a boolean image."""
    m = m.reshape(h, w)
    p = np.pad(m, 1)
    out = np.zeros_like(m)
    for dy in range(3):
        for dx in range(3):
            out |= p[dy:dy + h, dx:dx + w]
    return out.reshape(-1)


def make_masks(ref, ref_fma, orc, cam, sc, tex, sky, depth, through=None):
    """The three discontinuity masks of one frame -> (reference frame, dict of boolean arrays).
    ref / ref_fma: the reference kernels built without / with contraction -- or None for a scene the reference's
    one-byte counts cannot express (C4's 10 000 spheres, raytracing.cl:17): the pair is then the restatement itself,
    liboracle.so / liboracle_fma.so."""
    w, h = cam.width, cam.height
    if ref is not None:
        a, oob = ref.render(cam, sc, tex, sky, depth)
        b, _ = ref_fma.render(cam, sc, tex, sky, depth)
        assert oob == 0
    else:
        a, _, cnt = orc.render(cam, sc, tex, sky, depth)
        ofma = Oracle(os.path.join(os.path.dirname(os.path.abspath(__file__)), "liboracle_fma.so"))
        if through is not None:
            ofma.set_transparent_through(through)
        b, _, _ = ofma.render(cam, sc, tex, sky, depth)
        # (float -> int conversions out of range are undefined in OpenCL C; the restatement and the HIP kernels both convert like
        # AMD hardware does, saturating -- the count is printed, such pixels are the restatement's word against nobody's)
        print(f"  restatement pair: oob_reads {cnt.oob_reads} int_cast_oor {cnt.int_cast_oor}", flush=True)
    make_masks.last_agree = float((a == b).mean())     # the scene's own noise floor: two legal builds of one source
    fma = dilate1(channel_diff(a, b) > 0, w, h)
    o0, r0, _ = orc.render(cam, sc, tex, sky, depth, want_rgb=True)
    assert np.array_equal(o0, a), "the restatement must equal the reference kernels bit for bit"
    jit = np.zeros(w * h, bool)
    for delta in JITTERS:
        for dx, dy in ((delta, 0.0), (0.0, delta)):
            _, rp, _ = orc.render(shifted_camera(cam, dx, dy), sc, tex, sky, depth, want_rgb=True)
            _, rm, _ = orc.render(shifted_camera(cam, -dx, -dy), sc, tex, sky, depth, want_rgb=True)
            with np.errstate(invalid="ignore"):
                jit |= np.nan_to_num(np.abs(rp + rm - 2 * r0).max(1), nan=1e9) > JITTER_TAU
    om, mar = render_margins(cam, sc, tex, sky, depth, through)
    assert np.array_equal(om, a)
    margin = (mar[:, MARGIN_SITES.index("disc")] < DISC_MARGIN) | (mar[:, MARGIN_SITES.index("cast")] < CAST_MARGIN)
    return a, dict(fma=fma, jitter=jit, margin=margin)



# the synthetic BASELINE scenes (configs C3 / C4) at sizes the oracle finishes in minutes: masks + CRC only
SCENE_FRAMES = {
    "c3_1024x1024_d8": dict(scene=lambda: S.dielectric_field_scene(8), origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), w=1024, h=1024, depth=8, ref=True),
    "c4_480x270_d4": dict(scene=lambda: S.sphere_grid_scene(100, 100), origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), w=480, h=270, depth=4, ref=False),
}


def main_scenes():
    """tests/golden/masks_scenes.npz: the same three mask planes + CRC for the C3 / C4 scenes (python -m oracle.gen_golden scenes)."""
    import zlib
    orc = Oracle()
    tex, sky = T.texture_layers(), T.skybox_cross(512)
    masks = {}
    for key, f in SCENE_FRAMES.items():
        sc = f["scene"]()
        cam = orc.camera(f["origin"], f["look"], 90.0, 1.0, f["w"], f["h"])
        ref, ref_fma = (Reference(), Reference(REF_FMA_SO)) if f["ref"] else (None, None)
        img, mk = make_masks(ref, ref_fma, orc, cam, sc, tex, sky, f["depth"])
        for name, m in mk.items():
            masks[f"{key}_{name}"] = np.packbits(m)
        masks[f"{key}_crc32"] = np.array([zlib.crc32(img.tobytes())], np.uint32)
        masks[f"{key}_fma_agree"] = np.array([make_masks.last_agree], np.float64)
        un = mk["fma"] | mk["jitter"] | mk["margin"]
        print(f"{key}: masks fma {mk['fma'].mean():.4f} jitter {mk['jitter'].mean():.4f} margin {mk['margin'].mean():.4f} union {un.mean():.4f}"
              f"; contraction on/off builds agree on {make_masks.last_agree:.4f}", flush=True)
    # the reference's own committed render (tests/golden/reference_scene/): 800x600, depth 15, the real assets, transparent-shadow
    # factor 1.0 (see whitted_oracle.c wo_set_transparent_through); restatement pair, as for C4
    from example_gui_opencl_raytracer_amd import api
    fx = os.path.join(GOLD, "reference_scene")
    rtex = np.stack([api.read_png(os.path.join(fx, n + ".png")) for n in ("cobblestone", "sand", "check", "grass")])   # raypng.c:74-78
    rsky = api.read_png(os.path.join(fx, "stormydays.png"))[None]                                                      # raypng.c:80-81
    orc.set_transparent_through(1.0)
    cam = orc.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 800, 600)
    img, mk = make_masks(None, None, orc, cam, S.Scene.load(os.path.join(fx, "render.map")), rtex, rsky, 15, through=1.0)
    orc.set_transparent_through(0.8)
    key = "scene_png_800x600_d15"
    for name, m in mk.items():
        masks[f"{key}_{name}"] = np.packbits(m)
    masks[f"{key}_crc32"] = np.array([zlib.crc32(img.tobytes())], np.uint32)
    masks[f"{key}_fma_agree"] = np.array([make_masks.last_agree], np.float64)
    un = mk["fma"] | mk["jitter"] | mk["margin"]
    print(f"{key}: masks fma {mk['fma'].mean():.4f} jitter {mk['jitter'].mean():.4f} margin {mk['margin'].mean():.4f} union {un.mean():.4f}", flush=True)
    np.savez_compressed(os.path.join(GOLD, "masks_scenes.npz"), **masks)
    print("masks_scenes.npz", os.path.getsize(os.path.join(GOLD, "masks_scenes.npz")), "bytes")


def unit(rng, n):
    v = rng.normal(size=(n, 3)).astype(np.float32)
    return (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32)


def make_vector_inputs(seed=20240229):
    """Seeded inputs, shared with the tests (they re-create them and compare outputs)."""
    rng = np.random.default_rng(seed)
    d = {}
    d["ray_o"] = rng.uniform(-6, 6, (N, 3)).astype(np.float32)
    d["ray_d"] = unit(rng, N)
    d["sph_c"] = rng.uniform(-5, 5, (N, 3)).astype(np.float32)
    d["sph_r"] = rng.uniform(0.05, 2.5, N).astype(np.float32)
    # sphere tests: half the rays are aimed at (or just past) the sphere, 200 start inside it
    d["sph_o"] = d["ray_o"].copy()
    d["sph_o"][N - 200:] = d["sph_c"][N - 200:] + unit(rng, 200) * (d["sph_r"][N - 200:, None] * np.float32(0.5))
    aim = d["sph_c"] + unit(rng, N) * (d["sph_r"][:, None] * rng.uniform(0, 1.3, (N, 1)).astype(np.float32))
    to = (aim - d["sph_o"]).astype(np.float32)
    d["sph_d"] = d["ray_d"].copy()
    d["sph_d"][: N // 2] = (to / np.linalg.norm(to, axis=1, keepdims=True)).astype(np.float32)[: N // 2]
    d["pl_n"] = unit(rng, N)
    d["pl_n"][: N // 8] = np.eye(3, dtype=np.float32)[rng.integers(0, 3, N // 8)]  # axis-aligned planes
    d["pl_n"][: N // 16, :] *= -1
    d["pl_p"] = rng.uniform(-4, 4, (N, 3)).astype(np.float32)
    d["nrm"] = unit(rng, N)
    d["inc"] = unit(rng, N)
    flip = (np.einsum("ij,ij->i", d["nrm"], d["inc"]) > 0)
    d["inc"][flip] *= -1                                   # incident rays face the surface
    pairs = np.array([(1.0, 1.52), (1.52, 1.0), (1.0, 1.57), (1.57, 1.0), (1.0, 1.0), (1.4, 1.0)], np.float32)
    d["n12"] = pairs[rng.integers(0, len(pairs), N)]
    d["dir"] = unit(rng, N)
    d["dir"][:6] = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], np.float32)
    d["dir"][6:9] = np.array([[1, 1, 0], [1, 0, 1], [0, 1, 1]], np.float32) / np.float32(np.sqrt(2))  # face ties
    d["seed"] = rng.integers(0, 2**32, N, dtype=np.uint64).astype(np.uint32)
    d["seed"][:4] = [0, 1, 2, 0xFFFFFFFF]
    d["mod_a"] = rng.integers(-(2**31), 2**31 - 1, N).astype(np.int32)
    d["mod_a"][:6] = [0, -1, 255, 256, -256, -2147483648]
    d["mod_b"] = np.full(N, 256, np.int32)
    d["tex_p"] = rng.uniform(-30, 30, (N, 3)).astype(np.float32)
    d["sh_from"] = rng.uniform(-5, 5, (N, 3)).astype(np.float32)
    d["sh_from"][:, 1] = np.abs(d["sh_from"][:, 1]) * 0.2 + 0.001
    d["sh_to"] = rng.uniform(-3, 4, (N, 3)).astype(np.float32)
    d["sh_to"][:, 1] = np.abs(d["sh_to"][:, 1]) + 1.0
    return d


def main():
    if not Reference.available():
        raise SystemExit("oracle/_ref/libref_cl.so is missing: run `make -C oracle ref` where /root/reference exists")
    os.makedirs(GOLD, exist_ok=True)
    ref, orc = Reference(), Oracle()
    ref_fma = Reference(REF_FMA_SO)
    sc = S.render_map_scene()
    tex, sky = T.texture_layers(), T.skybox_cross(512)

    import zlib
    frames, masks = {}, {}
    for (w, h, depth) in FRAMES + MASK_ONLY_FRAMES:
        cam = orc.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
        img, mk = make_masks(ref, ref_fma, orc, cam, sc, tex, sky, depth)
        key = f"render_map_{w}x{h}_d{depth}"
        if (w, h, depth) in FRAMES:
            frames[key] = img
        for name, m in mk.items():
            masks[f"{key}_{name}"] = np.packbits(m)
        masks[f"{key}_crc32"] = np.array([zlib.crc32(img.tobytes())], np.uint32)
        un = mk["fma"] | mk["jitter"] | mk["margin"]
        print(f"{key}: masks fma {mk['fma'].mean():.4f} jitter {mk['jitter'].mean():.4f} margin {mk['margin'].mean():.4f} union {un.mean():.4f}")
    np.savez_compressed(os.path.join(GOLD, "masks.npz"), **masks)

    rc = RefCamera()
    cin = np.array([list(o) + list(l) + [fov, focal, w, h] for (o, l, fov, focal, w, h) in CAMERAS], np.float64)
    cout = np.stack([rc.perspective(o, l, fov, focal, w, h) for (o, l, fov, focal, w, h) in CAMERAS])
    np.savez_compressed(os.path.join(GOLD, "camera.npz"), inputs=cin, outputs=cout)
    cam = orc.camera(CAM["origin"], CAM["look"], 90.0, 1.0, 160, 120)
    frames["raygen_160x120"] = ref.raygen(cam)
    frames["camera_160x120"] = np.frombuffer(bytes(cam), np.uint8).copy()
    np.savez_compressed(os.path.join(GOLD, "frames.npz"), **frames)

    d = make_vector_inputs()
    L = ref.lib
    out = {}
    f1 = C.c_float()
    t = np.zeros(N, np.float32); hit = np.zeros(N, np.int32)
    for i in range(N):
        hit[i] = L.ref_intersect_sphere(_f3(d["sph_o"][i]), _f3(d["sph_d"][i]), _f3(d["sph_c"][i]), float(d["sph_r"][i]), C.byref(f1))
        t[i] = f1.value
    out["sphere_hit"], out["sphere_t"] = hit.copy(), t.copy()
    for i in range(N):
        hit[i] = L.ref_intersect_plane(_f3(d["ray_o"][i]), _f3(d["ray_d"][i]), _f3(d["pl_n"][i]), _f3(d["pl_p"][i]), C.byref(f1))
        t[i] = f1.value
    out["plane_hit"], out["plane_t"] = hit.copy(), t.copy()
    v = (C.c_float * 3)()
    refl = np.zeros((N, 3), np.float32); refr = np.zeros((N, 3), np.float32); sch = np.zeros(N, np.float32)
    for i in range(N):
        L.ref_reflect(_f3(d["inc"][i]), _f3(d["nrm"][i]), v); refl[i] = v[:]
        L.ref_refract(float(d["n12"][i, 0]), float(d["n12"][i, 1]), _f3(d["inc"][i]), _f3(d["nrm"][i]), v); refr[i] = v[:]
        sch[i] = L.ref_schlick(float(d["n12"][i, 0]), float(d["n12"][i, 1]), _f3(d["inc"][i]), _f3(d["nrm"][i]))
    out["reflect"], out["refract"], out["schlick"] = refl, refr, sch
    uv = (C.c_int * 2)(); cube = np.zeros((N, 2), np.int32)
    for i in range(N):
        L.ref_map_to_cube(_f3(d["dir"][i]), 1024, uv); cube[i] = uv[:]
    out["cube_uv"] = cube
    xs = np.zeros(N, np.float32); st = np.zeros(N, np.uint32)
    for i in range(N):
        s = C.c_uint(int(d["seed"][i])); xs[i] = L.ref_xorshift32(C.byref(s)); st[i] = s.value
    out["xorshift_val"], out["xorshift_state"] = xs, st
    out["emod"] = np.array([L.ref_euclidean_modulo(int(a), int(b)) for a, b in zip(d["mod_a"], d["mod_b"])], np.int32)
    texc = np.zeros((N, 3), np.float32)
    planes = sc.planes[:1].copy()
    for i in range(N):
        pl = planes.copy()
        pl["normal"][0] = d["pl_n"][i]
        pl["material"]["texture_id"][0] = i % 4
        pl["material"]["texture_scale"][0] = np.float32(1 + (i % 7) * 16.5)
        L.ref_plane_texture_pixel(_p(pl), _f3(d["tex_p"][i]), _p(tex), tex.shape[2], tex.shape[1], tex.shape[0], v)
        texc[i] = v[:]
    out["plane_texel"] = texc
    sh = np.zeros(N, np.float32)
    for i in range(N):
        sh[i] = L.ref_shadow(_f3(d["sh_to"][i]), _f3(d["sh_from"][i]), _p(sc.spheres), len(sc.spheres), _p(sc.planes), len(sc.planes))
    out["shadow"] = sh
    lh = np.zeros(N, np.int32); lc = np.zeros((N, 3), np.float32)
    aim = sc.lights["origin"][np.arange(N) % 3] + (d["dir"] * np.float32(0.08))     # half the rays aim at a light
    ld = (aim - d["ray_o"]); ld = (ld / np.linalg.norm(ld, axis=1, keepdims=True)).astype(np.float32)
    ld[N // 2:] = d["ray_d"][N // 2:]
    for i in range(N):
        lh[i] = L.ref_find_light(_f3(d["ray_o"][i]), _f3(ld[i]), _p(sc.lights), len(sc.lights), _p(sc.spheres), len(sc.spheres),
                                 _p(sc.planes), len(sc.planes), v)
        lc[i] = v[:]
    out["light_dir"], out["light_hit"], out["light_color"] = ld, lh, lc
    sh_ = np.zeros(N, np.int32); sp = np.zeros((N, 3), np.float32); sn = np.zeros((N, 3), np.float32)
    sm = np.zeros((N, 16), np.uint32)
    pt, nm = (C.c_float * 3)(), (C.c_float * 3)()
    mat = np.zeros(16, np.uint32)
    o_cam = np.array(CAM["origin"], np.float32)
    for i in range(N):
        sh_[i] = L.ref_find_solid(_f3(o_cam), _f3(d["ray_d"][i]), _p(sc.spheres), len(sc.spheres), _p(sc.planes), len(sc.planes),
                                  _p(tex), tex.shape[2], tex.shape[1], tex.shape[0], pt, nm, _p(mat))
        if sh_[i]:
            sp[i], sn[i], sm[i] = pt[:], nm[:], mat
    sm[:, 3] = 0; sm[:, 14:] = 0   # struct padding is not part of the contract
    out["solid_hit"], out["solid_point"], out["solid_normal"], out["solid_material"] = sh_, sp, sn, sm
    np.savez_compressed(os.path.join(GOLD, "vectors.npz"), **out)
    for f in ("frames.npz", "vectors.npz", "masks.npz", "camera.npz"):
        print(f, os.path.getsize(os.path.join(GOLD, f)), "bytes")


if __name__ == "__main__":
    main_scenes() if sys.argv[1:] == ["scenes"] else main()
