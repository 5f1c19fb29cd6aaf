"""Seeded random scenes shared by the CPU (oracle vs reference kernels) and GPU (HIP vs oracle) fuzz tests."""
import numpy as np

from example_gui_opencl_raytracer_amd import scene as S


def random_scene(seed: int):
    rng = np.random.default_rng(seed)
    ns, npl, nl = int(rng.integers(0, 9)), int(rng.integers(0, 4)), int(rng.integers(0, 5))
    presets = [S.stone, S.plastic, S.mirror, S.glass]
    sph = np.zeros(ns, S.SPHERE)
    for i in range(ns):
        sph[i]["origin"] = rng.uniform([-4, 0.2, -2], [5, 3, 7])
        sph[i]["radius"] = rng.uniform(0.2, 1.2)
        m = presets[int(rng.integers(0, 4))]()
        m["rgb"] = rng.uniform(0, 1, 3)
        m["shininess"] = int(rng.integers(0, 200))
        m["reflectivity"] = rng.choice([0.0, 0.04, 0.1, 0.5, 1.0])
        m["n"] = rng.choice([1.0, 1.33, 1.52, 2.4])
        m["texture_id"] = -1
        sph[i]["material"] = m
    pln = np.zeros(npl, S.PLANE)
    for i in range(npl):
        n = rng.normal(size=3) if i else np.array([0.0, 1.0, 0.0])
        n = (n / np.linalg.norm(n)).astype(np.float32)
        pln[i]["normal"] = n
        pln[i]["point_in_plane"] = (0, 0, 0) if i == 0 else (n * -rng.uniform(4, 9)).astype(np.float32)
        m = presets[int(rng.integers(0, 3))]()
        m["rgb"] = rng.uniform(0, 1, 3)
        m["texture_id"] = int(rng.integers(-1, 4))
        m["texture_scale"] = rng.choice([1.0, 17.5, 100.0])
        pln[i]["material"] = m
    lgt = np.zeros(nl, S.LIGHT)
    for i in range(nl):
        lgt[i]["origin"] = rng.uniform([-4, 1.5, -3], [4, 6, 6])
        lgt[i]["radius"] = rng.uniform(0.05, 0.4)
        lgt[i]["intensity"] = rng.uniform(3, 40)
        lgt[i]["rgb"] = rng.uniform(0, 1, 3)
    cam = dict(origin=tuple(rng.uniform([-3, 0.5, -9], [3, 4, -4]).astype(np.float32).tolist()),
               look=tuple(rng.uniform([-0.4, -0.4, 0.8], [0.4, 0.2, 1.0]).astype(np.float32).tolist()),
               fov=float(rng.choice([60.0, 90.0, 110.0])), focal=1.0)
    depth = int(rng.choice([1, 2, 3, 4, 8, 15]))
    return S.Scene(sph, pln, lgt), cam, depth
