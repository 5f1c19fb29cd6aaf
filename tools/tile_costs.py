"""Exploratory: distribution of per-tile costs (critical path of the launch) for a config."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
W, H, depth = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (800, 600, 15)
r = Renderer(scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=depth)
r.look(**pkg.CAMERA_RAYPNG)
r.render(readback=False)
c = r.w.read_tile_costs()
print("tiles", len(c), "sum", int(c.sum()), "mean", float(c.mean()), "max", int(c.max()), "p99", float(np.percentile(c, 99)), "p90", float(np.percentile(c, 90)))
print("top 12:", np.sort(c)[-12:])
print("hist:", np.histogram(c, bins=[0, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 1 << 30])[0])
r.release()
