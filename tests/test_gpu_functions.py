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
