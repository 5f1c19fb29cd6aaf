"""Time one BASELINE.json configuration through the C-ABI (kernel-only, frames back to back) and print JSON.
   python tools/run_config.py c2|c3|c4|c5strip [--strict 1] [--frames N] [--variant V] [--depth D]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--strict", type=int, default=0)
ap.add_argument("--frames", type=int, default=20)
ap.add_argument("--variant", type=int, default=0)
ap.add_argument("--depth", type=int, default=None)
ap.add_argument("--png", default=None)
a = ap.parse_args()
tex, sky = textures.texture_layers(), textures.skybox_cross(4096)
cam = pkg.CAMERA_RAYPNG
kw = {}
if a.config == "c2":
    sc, W, H, depth = scene.render_map_scene(), 1920, 1080, 4
elif a.config == "c3":      # 4096x4096 depth 8, 64 dielectric spheres (divergence stress)
    sc, W, H, depth = scene.dielectric_field_scene(8), 4096, 4096, 8
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
elif a.config == "c4":      # 10k-sphere grid, 1920x1080 depth 4 (geometry streamed, not LDS/SGPR resident)
    sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
elif a.config == "c5strip":  # one GPU's 8192x1024 strip of the 8192x8192 frame (rank 3 of 8)
    sc, W, H, depth = scene.render_map_scene(), 8192, 8192, 4
    kw = dict(first_row=3 * 1024, rows=1024)
elif a.config == "ref800":  # the reference driver's own configuration: 800x600, depth 15
    sc, W, H, depth = scene.render_map_scene(), 800, 600, 15
else:
    raise SystemExit("unknown config")
depth = a.depth or depth
r = Renderer(sc, tex, sky, W, H, depth=depth, strict=bool(a.strict), **kw)
r.w.set_variant(a.variant)
r.look(**cam)
r.render(readback=False); r.render(readback=False)
r.w.enable_counters(1); r.render(readback=False); c = r.w.read_counters(); r.w.enable_counters(0)
r.w.timing_reset(); r.w.set_async(1)
t = time.perf_counter()
for _ in range(a.frames):
    r.render(readback=False)
r.w.sync()
wall = (time.perf_counter() - t) / a.frames
n, ms = r.w.timing_get(1)
r.w.set_async(0)
img = r.render()
rays = c["segments"] + c["shadow_rays"]
px = r.pixels
print(json.dumps(dict(config=a.config, frame=f"{W}x{H}", pixels=px, depth=depth, strict=a.strict, variant=a.variant, kernel_ms=round(ms / n, 4),
                      wall_ms_per_frame=round(wall * 1e3, 4), rays_per_px=round(rays / px, 3), Mrays_s=round(rays / (ms / n) / 1e3, 1),
                      lane_util=round(c["lane_iters"] / max(c["wave_iters_x64"], 1), 4), counters=c)), flush=True)
if a.png:
    from example_gui_opencl_raytracer_amd import api
    api.write_png(a.png, img, W, r.rows)
r.release()
