"""Exploratory: cost of ONE wave running alone (8x8 frame aimed at the glass spheres), per loop iteration."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
sc, tex, sky = scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(4096)
for depth in (4, 15):
    for fov in (4.0, 1.0):
        for variant in (0, 16):
            r = Renderer(sc, tex, sky, 8, 8, depth=depth)
            r.w.set_variant(variant)
            r.look(origin=(0.8, 2.5, -8.0), look=(0.0, -0.18, 1.0), fov=fov, focal=1.0)   # towards glass sphere #2
            for _ in range(3):
                r.render(readback=False)
            r.w.enable_counters(1); r.render(readback=False); c = r.w.read_counters(); r.w.enable_counters(0)
            r.w.timing_reset()
            for _ in range(20):
                r.render(readback=False)
            n, ms = r.w.timing_get(1)
            cost = r.w.read_tile_costs()
            iters = c["wave_iters_x64"] // 64
            print(f"depth {depth} fov {fov} variant {variant}: kernel {ms / n * 1e3:.1f} us, wave iterations {iters}, lane-iters {c['lane_iters']}, "
                  f"tile cost {cost.tolist()}, {ms / n * 1e3 / max(iters, 1):.2f} us per wave iteration")
            r.release()
