"""Function-level parity on the GPU: each device helper of the trace kernel, run through clw_ext_unit, against the
golden vectors produced by the REFERENCE's own functions (tests/golden/vectors.npz, oracle/gen_golden.py).
strict build: bit-exact.  fast build: booleans equal away from thresholds, continuous outputs within 1e-4 relative
(SURVEY.md 8(c) bar 1); the fast sincos / pow / normalize against float64 within their stated ulp bounds."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

OP = dict(sphere=0, plane=1, reflect=2, refract=3, schlick=4, cube=5, xorshift=6, emod=7, sincos=8, pow=9, normalize=10)


@pytest.fixture(scope="module")
def inputs():
    from oracle.gen_golden import make_vector_inputs
    return make_vector_inputs()


@pytest.fixture(scope="module", params=[True, False], ids=["strict", "fast"])
def dev(request):
    import torch  # noqa: F401
    from example_gui_opencl_raytracer_amd import api
    w = api.ClWrap()
    w.set_strict(request.param)
    yield w, request.param
    w.release()


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def close(got, want, strict, rel=1e-4):
    if strict:
        assert np.array_equal(bits(got), bits(want))
    else:
        assert np.allclose(got, want, rtol=rel, atol=1e-6)


def test_intersect_sphere(dev, inputs, golden_vectors):
    w, strict = dev
    rows = np.concatenate([inputs["sph_o"], inputs["sph_d"], inputs["sph_c"], inputs["sph_r"][:, None]], 1)
    out = w.unit(OP["sphere"], rows, 2)
    hit, t = out[:, 0] != 0, out[:, 1]
    ghit, gt = golden_vectors["sphere_hit"] != 0, golden_vectors["sphere_t"]
    if strict:
        assert np.array_equal(hit, ghit) and np.array_equal(bits(t), bits(gt))
    else:
        assert (hit != ghit).mean() < 2e-3                     # grazing rays may flip
        both = hit & ghit
        assert np.allclose(t[both], gt[both], rtol=1e-4, atol=1e-6)


def test_intersect_plane(dev, inputs, golden_vectors):
    w, strict = dev
    rows = np.concatenate([inputs["ray_o"], inputs["ray_d"], inputs["pl_n"], inputs["pl_p"]], 1)
    out = w.unit(OP["plane"], rows, 2)
    hit, t = out[:, 0] != 0, out[:, 1]
    ghit, gt = golden_vectors["plane_hit"] != 0, golden_vectors["plane_t"]
    assert np.array_equal(hit, ghit)
    close(t, gt, strict)


def test_reflect_refract_schlick(dev, inputs, golden_vectors):
    w, strict = dev
    close(w.unit(OP["reflect"], np.concatenate([inputs["inc"], inputs["nrm"]], 1), 3), golden_vectors["reflect"], strict)
    rows = np.concatenate([inputs["n12"], inputs["inc"], inputs["nrm"]], 1)
    out = w.unit(OP["refract"], rows, 4)
    tir = np.isnan(golden_vectors["refract"][:, 0])                   # the reference returns a NaN vector, the caller drops the child
    assert np.array_equal(out[:, 0] == 0, tir)
    close(out[~tir, 1:], golden_vectors["refract"][~tir], strict)
    close(w.unit(OP["schlick"], rows, 1)[:, 0], golden_vectors["schlick"], strict)


def test_map_to_cube_xorshift_emod(dev, inputs, golden_vectors):
    w, strict = dev
    uv = w.unit(OP["cube"], inputs["dir"], 2, aux=1024).view(np.int32)
    if strict:
        assert np.array_equal(uv, golden_vectors["cube_uv"])
    else:
        assert (np.abs(uv - golden_vectors["cube_uv"]).max(1) <= 1).all() and (uv == golden_vectors["cube_uv"]).all(1).mean() > 0.995
    out = w.unit(OP["xorshift"], inputs["seed"].view(np.float32)[:, None], 2)
    assert np.array_equal(out[:, 0].view(np.uint32), golden_vectors["xorshift_state"])       # integer work: exact in both builds
    assert np.array_equal(bits(out[:, 1]), bits(golden_vectors["xorshift_val"]))
    rows = np.stack([inputs["mod_a"].view(np.float32), inputs["mod_b"].view(np.float32)], 1)
    assert np.array_equal(w.unit(OP["emod"], rows, 1)[:, 0].view(np.int32), golden_vectors["emod"])


def _ulp_err(got, want64):
    want = want64.astype(np.float32)
    ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
    ulp = np.maximum(ulp, np.float64(2.0 ** -149))
    return np.abs(got.astype(np.float64) - want64) / ulp


def test_sincos_over_the_sampling_range(dev):
    """The angles are theta = fl32(2*pi*u), phi = fl32(pi*u) with u in [0,4) and an fp64 product (primitives.cl:116-125,
    raytracing.cl:99-100); the unit op takes u and returns sin/cos of that fp32 angle.  strict: OpenCL's 4 ulp.  fast
    (hardware v_sin/v_cos on u as the exact revolution count): an ABSOLUTE bound of 1.2e-6 = the reference's own
    rounding of the angle (0.95e-6 rad at 8 pi) + the hardware's 1.25e-7 -- the values only scale the light radius
    (0.1) before being added to the light centre, and the shadow ray they define has a binary outcome."""
    w, strict = dev
    rng = np.random.default_rng(11)
    u = np.concatenate([np.linspace(0, 4, 200001, dtype=np.float64)[:-1].astype(np.float32),
                        (rng.integers(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.float32) / np.float32(2 ** 30)).astype(np.float32)])
    u = u[u < 4.0]
    for full, scale in ((1, 2 * np.pi), (0, np.pi)):
        x = (scale * u.astype(np.float64)).astype(np.float32).astype(np.float64)
        out = w.unit(OP["sincos"], u[:, None], 2, aux=full)
        bound = 2e-7 if strict else 1.2e-6
        assert np.abs(out[:, 0] - np.sin(x)).max() < bound and np.abs(out[:, 1] - np.cos(x)).max() < bound
        if not strict:   # ... and against the unrounded angle the error is the hardware's alone
            xe = scale * u.astype(np.float64)
            assert np.abs(out[:, 0] - np.sin(xe)).max() < 2e-7 and np.abs(out[:, 1] - np.cos(xe)).max() < 2e-7
        if strict:
            es, ec = _ulp_err(out[:, 0], np.sin(x)), _ulp_err(out[:, 1], np.cos(x))
            small = np.abs(np.sin(x)) > 1e-3, np.abs(np.cos(x)) > 1e-3       # ulp is ill-defined at the zeros
            assert es[small[0]].max() <= 4.0 and ec[small[1]].max() <= 4.0    # OpenCL's bound for sin / cos


def test_pow_of_the_phong_term(dev):
    """base = max(0, n.h) in [0,1], exponent = (float)uint shininess (raytracing.cl:129)."""
    w, strict = dev
    x = np.linspace(0, 1, 100001, dtype=np.float32)
    for y in (0.0, 1.0, 20.0, 50.0, 100.0, 150.0, 255.0):
        out = w.unit(OP["pow"], np.stack([x, np.full_like(x, y)], 1), 1)[:, 0]
        want = np.power(x.astype(np.float64), y)
        assert np.abs(out - want).max() < 2e-7
        big = want >= 2.0 ** -10
        assert _ulp_err(out[big], want[big]).max() <= 16.0                # OpenCL's bound for pow
    assert w.unit(OP["pow"], np.array([[0.0, 0.0], [0.0, 5.0], [1.0, 150.0]], np.float32), 1)[:, 0].tolist() == [1.0, 0.0, 1.0]


def test_normalize(dev):
    w, strict = dev
    rng = np.random.default_rng(3)
    v = (rng.normal(size=(20000, 3)) * np.exp(rng.uniform(-6, 6, (20000, 1)))).astype(np.float32)
    out = w.unit(OP["normalize"], v, 4)
    n64 = np.linalg.norm(v.astype(np.float64), axis=1)
    assert np.allclose(out[:, 3], n64, rtol=4e-7)
    assert np.allclose(out[:, :3], v / n64[:, None], rtol=0, atol=4e-7)
    if strict:                                                    # = v / sqrtf(dot) with IEEE ops, like the oracle
        d = (v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2]
        ln = np.sqrt(d, dtype=np.float32)
        assert np.array_equal(bits(out[:, 3]), bits(ln)) and np.array_equal(bits(out[:, :3]), bits(v / ln[:, None]))


# ------------------------------------------------------------ the helpers that need the scene (clw_ext_unit_scene)
US = dict(hit=0, shadow=1, texel=2)


@pytest.fixture(scope="module", params=[True, False], ids=["strict", "fast"])
def scene_dev(request, demo_scene, tex, sky):
    """A wrapper with render.map bound through the reference's call protocol; the unit ops run on that scene."""
    import torch  # noqa: F401
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    r = Renderer(demo_scene, tex, sky, 64, 64, depth=4, strict=request.param)
    yield r.w, request.param
    r.release()


def test_find_light_intersection(scene_dev, inputs, golden_vectors):
    """H4, reference primitives.cl:262-318, through the kernel's own hit phase (whitted_hit.inc)."""
    w, strict = scene_dev
    rows = np.concatenate([inputs["ray_o"], golden_vectors["light_dir"]], 1)
    out = w.unit_scene(US["hit"], rows, 22)
    lit, glit = out[:, 0] != 0, golden_vectors["light_hit"] != 0
    assert glit.sum() > 100                                          # half the rays were aimed at a light
    if strict:
        assert np.array_equal(lit, glit)
        assert np.array_equal(bits(out[glit, 1:4]), bits(golden_vectors["light_color"][glit]))
    else:
        assert (lit != glit).mean() < 2e-3                            # rays grazing a light's silhouette may flip
        both = lit & glit
        assert np.allclose(out[both, 1:4], golden_vectors["light_color"][both], rtol=1e-4, atol=1e-6)


def test_find_solid_intersection(scene_dev, inputs, golden_vectors):
    """H5, reference primitives.cl:322-394: nearest hit, offset point, normal and the winner's material (with the
    texel of a textured plane as its colour)."""
    from conftest import CAM
    w, strict = scene_dev
    o = np.broadcast_to(np.array(CAM["origin"], np.float32), inputs["ray_d"].shape)
    out = w.unit_scene(US["hit"], np.concatenate([o, inputs["ray_d"]], 1), 22)
    lit = out[:, 0] != 0                                              # a light in view ends the segment before the search
    hit, ghit = out[:, 4] != 0, golden_vectors["solid_hit"] != 0
    ok = ~lit
    assert ok.mean() > 0.99 and ghit[ok].sum() > 500
    gm = golden_vectors["solid_material"]
    gmat = np.concatenate([gm[:, 0:3].view(np.float32), gm[:, 4:7].view(np.float32), gm[:, 7:8].astype(np.float32),
                           gm[:, 8:10].astype(np.float32), gm[:, 10:12].view(np.float32)], 1)        # rgb, amb, diff, spec, shin, transp, diel, n, refl
    both = ok & hit & ghit
    if strict:
        assert np.array_equal(hit[ok], ghit[ok])
        assert np.array_equal(bits(out[both, 5:8]), bits(golden_vectors["solid_point"][both]))
        assert np.array_equal(bits(out[both, 8:11]), bits(golden_vectors["solid_normal"][both]))
        assert np.array_equal(bits(out[both, 11:22]), bits(gmat[both]))
    else:
        assert (hit[ok] != ghit[ok]).mean() < 2e-3
        assert np.allclose(out[both, 5:8], golden_vectors["solid_point"][both], rtol=1e-4, atol=1e-4)
        assert np.allclose(out[both, 8:11], golden_vectors["solid_normal"][both], rtol=1e-4, atol=1e-5)
        same_mat = (out[both, 14:22] == gmat[both, 3:]).all(1)        # same primitive won (silhouette rays may pick the neighbour)
        assert same_mat.mean() > 0.998
        texel_same = (np.abs(out[both, 11:14] - gmat[both, 0:3]).max(1) < 1e-6)
        assert texel_same.mean() > 0.99                               # a hit point within 1e-7 of a texel edge may take the neighbour texel


def test_shadow_path(scene_dev, inputs, golden_vectors):
    """H6, reference primitives.cl:396-442, one ray through the kernel's batched shadow traversal."""
    w, strict = scene_dev
    out = w.unit_scene(US["shadow"], np.concatenate([inputs["sh_to"], inputs["sh_from"]], 1), 1)[:, 0]
    want = golden_vectors["shadow"]
    assert len(np.unique(want)) >= 3                                  # blocked, clear and through-glass cases all occur
    if strict:
        assert np.array_equal(bits(out), bits(want))
    else:
        assert (out != want).mean() < 2e-3                            # binary outcome: only grazing rays may flip
        assert set(np.unique(out)) <= set(np.unique(want)) | {np.float32(0.8) ** k for k in range(5)}


def test_plane_texture_pixel(scene_dev, inputs, golden_vectors, demo_scene):
    """H9, reference primitives.cl:217-259, with the tangent basis hoisted to scene preparation (scene_prep.c)."""
    import ctypes as C
    from example_gui_opencl_raytracer_amd import api
    w, strict = scene_dev
    n = len(inputs["tex_p"])
    planes = np.repeat(demo_scene.planes[:1], n)
    planes["normal"][:, :3] = inputs["pl_n"]
    planes["material"]["texture_id"] = np.arange(n) % 4
    planes["material"]["texture_scale"] = (1 + (np.arange(n) % 7) * 16.5).astype(np.float32)
    L = api.load_library()
    L.wprep_geom_f4.restype = C.c_size_t
    L.wprep_geom_f4.argtypes = [C.c_uint32] * 3
    geom = np.zeros((L.wprep_geom_f4(0, n, 0), 4), np.float32)
    ptex = np.zeros((2 * n, 4), np.float32)
    L.wprep_build.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.wprep_build(None, 0, planes.ctypes.data, n, None, 0, geom.ctypes.data, ptex.ctypes.data)
    rows = np.concatenate([ptex.reshape(n, 8), inputs["tex_p"], np.zeros((n, 1), np.float32)], 1)     # stride 12 floats
    out = w.unit_scene(US["texel"], rows, 3)
    want = golden_vectors["plane_texel"]
    if strict:
        assert np.array_equal(bits(out), bits(want))
    else:
        assert (np.abs(out - want).max(1) < 1e-6).mean() > 0.995      # a coordinate within 1e-7 of a texel edge may flip
