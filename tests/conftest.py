import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE_ROOT = "/root/reference"   # exists only in the build container, never on the GPU box
CAM = dict(origin=(0.8, 2.5, -8.0), look=(0.2, 0.0, 1.0), fov=90.0, focal=1.0)   # raypng.c:17-21


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle.oracle_py import Oracle, build
    build(ref=os.path.isdir(REFERENCE_ROOT))
    return Oracle()


@pytest.fixture(scope="session")
def reference(oracle):
    from oracle.oracle_py import Reference
    if not Reference.available():
        pytest.skip("oracle/_ref/libref_cl.so not built (needs /root/reference)")
    return Reference()


@pytest.fixture(scope="session")
def demo_scene():
    from example_gui_opencl_raytracer_amd import scene
    return scene.render_map_scene()


@pytest.fixture(scope="session")
def tex():
    from example_gui_opencl_raytracer_amd import textures
    return textures.texture_layers()


@pytest.fixture(scope="session")
def sky():
    from example_gui_opencl_raytracer_amd import textures
    return textures.skybox_cross(512)


@pytest.fixture(scope="session")
def golden_frames():
    return dict(np.load(os.path.join(GOLDEN, "frames.npz")))


@pytest.fixture(scope="session")
def golden_masks():
    return dict(np.load(os.path.join(GOLDEN, "masks.npz")))


def frame_mask(golden_masks, key, n, parts=("fma", "jitter", "margin")):
    """Union of the named discontinuity masks of frame `key` as a boolean array of n pixels (oracle/gen_golden.py)."""
    m = np.zeros(n, bool)
    for name in parts:
        m |= np.unpackbits(golden_masks[f"{key}_{name}"])[:n].astype(bool)
    return m


@pytest.fixture(scope="session")
def golden_vectors():
    return dict(np.load(os.path.join(GOLDEN, "vectors.npz")))


def channel_diff(a, b):
    """max per-channel |difference| of two packed 0x00RRGGBB arrays, per pixel."""
    ca = np.stack([(a >> 16) & 255, (a >> 8) & 255, a & 255], 1).astype(np.int32)
    cb = np.stack([(b >> 16) & 255, (b >> 8) & 255, b & 255], 1).astype(np.int32)
    return np.abs(ca - cb).max(1)
