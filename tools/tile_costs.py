"""Distribution of the per-tile costs the scheduler sorts by (grid builds: wave lifetime in 256-tick units).
   python tools/tile_costs.py c4"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
cfg = sys.argv[1] if len(sys.argv) > 1 else "c4"
cam = pkg.CAMERA_RAYPNG
if cfg == "c2":
    sc, W, H, depth = scene.render_map_scene(), 1920, 1080, 4
elif cfg == "c3":
    sc, W, H, depth = scene.dielectric_field_scene(8), 4096, 4096, 8
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
elif cfg == "ref800":
    sc, W, H, depth = scene.render_map_scene(), 800, 600, 15
else:
    sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
r = Renderer(sc, textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=depth)
r.look(**cam)
for _ in range(6):
    r.render(readback=False)
r.w.sync()
r.w.timing_reset(); r.render(readback=False); r.w.sync(); n, ms = r.w.timing_get(1)
c = r.w.read_tile_costs().astype(np.int64).reshape((H + 7) // 8, (W + 7) // 8)
q = np.percentile(c, [50, 90, 99, 99.9, 100])
rows = c.mean(axis=1)
print(json.dumps(dict(config=cfg, kernel_ms=round(ms / n, 3), tiles=int(c.size), sum=int(c.sum()), mean=round(float(c.mean()), 1),
                      p50_p90_p99_p999_max=[float(x) for x in q],
                      ticks_per_ms_if_5120_slots=round(float(c.sum()) * 256 / 5120 / (ms / n), 1),
                      max_tile_ms_at_100MHz=round(float(q[-1]) * 256 / 1e5, 3),
                      row_means=[round(float(x), 1) for x in rows][:: max(1, len(rows) // 32)])))
r.release()
