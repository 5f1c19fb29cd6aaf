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
from oracle.oracle_py import Oracle, Reference, _f3, _p  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CAM = dict(origin=(0.8, 2.5, -8.0), look=(0.2, 0.0, 1.0))
FRAMES = [(160, 120, 1), (160, 120, 4), (160, 120, 15), (320, 240, 4)]
N = 2000


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
    sc = S.render_map_scene()
    tex, sky = T.texture_layers(), T.skybox_cross(512)

    frames = {}
    for (w, h, depth) in FRAMES:
        cam = orc.camera(CAM["origin"], CAM["look"], 90.0, 1.0, w, h)
        img, oob = ref.render(cam, sc, tex, sky, depth)
        assert oob == 0, "golden scene must not read outside the images"
        frames[f"render_map_{w}x{h}_d{depth}"] = img
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
    for f in ("frames.npz", "vectors.npz"):
        print(f, os.path.getsize(os.path.join(GOLD, f)), "bytes")


if __name__ == "__main__":
    main()
