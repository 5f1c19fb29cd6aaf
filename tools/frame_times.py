"""Per-frame kernel times of one configuration (is the cost-sorted order stable from frame to frame?).
   python tools/frame_times.py c4 [--frames N] [--variant V]"""
import argparse, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer

ap = argparse.ArgumentParser()
ap.add_argument("config")
ap.add_argument("--frames", type=int, default=40)
ap.add_argument("--variant", type=int, default=0)
a = ap.parse_args()
tex, sky = textures.texture_layers(), textures.skybox_cross(4096)
cam = pkg.CAMERA_RAYPNG
if a.config == "c2":
    sc, W, H, depth = scene.render_map_scene(), 1920, 1080, 4
elif a.config == "ref800":
    sc, W, H, depth = scene.render_map_scene(), 800, 600, 15
elif a.config == "c3":
    sc, W, H, depth = scene.dielectric_field_scene(8), 4096, 4096, 8
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
elif a.config == "c4":
    sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
else:
    raise SystemExit("unknown config")
r = Renderer(sc, tex, sky, W, H, depth=depth)
r.w.set_variant(a.variant)
r.look(**cam)
ts = []
for _ in range(a.frames):
    r.w.timing_reset()
    r.render(readback=False)
    r.w.sync()
    n, ms = r.w.timing_get(1)
    ts.append(round(ms / max(n, 1), 3))
print(json.dumps(dict(config=a.config, variant=a.variant, lib=os.environ.get("CLWRAP_LIB", ""), ms=ts)))
r.release()
