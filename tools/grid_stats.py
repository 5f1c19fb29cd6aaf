"""Grid-walk statistics of C4 (DIAGNOSTIC build with -DWT_GRIDSTATS=1; never the product).
   CLWRAP_LIB=.../libopencl_wrap_hip_gstat.so python tools/grid_stats.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
r = Renderer(sc, textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=depth)
r.look(**cam)
r.render(readback=False)
r.w.enable_counters(1); r.render(readback=False); c = r.w.read_counters(); raw = r.w.last_raw_counters
n = dict(walks=raw[10], lane_steps=raw[11], wave_steps=raw[12], tests=raw[13])
s = dict(walks=raw[14], lane_steps=raw[15], wave_steps=raw[16], tests=raw[17])
for d in (n, s):
    d["steps_per_walk"] = round(d["lane_steps"] / max(d["walks"], 1), 2)
    d["tests_per_walk"] = round(d["tests"] / max(d["walks"], 1), 2)
    d["lane_util_of_steps"] = round(d["lane_steps"] / max(64 * d["wave_steps"], 1), 3)
print(json.dumps(dict(lib=os.environ.get("CLWRAP_LIB", ""), counters=c, nearest=n, shadow=s, pool=dict(shadow_wave_tests=raw[21], wave_iterations=raw[18], lane_takes=raw[19], wave_takes=raw[20]))))
r.release()
