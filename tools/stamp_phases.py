"""Where a wave of the trace kernel spends its cycles (DIAGNOSTIC build with in-kernel stamps; never the product).
   python tools/stamp_phases.py [c2|c3|c4|ref800] [--strict 1]
Builds nothing: expects libopencl_wrap_hip_stamp.so (python -c "from example_gui_opencl_raytracer_amd import build; build.build(tag='_stamp', extra_device_flags=["-DWT_STAMPS=1", "-DWT_SHALLOW_WAVES=4"])")."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["CLWRAP_LIB"] = os.path.join(ROOT, "example_gui_opencl_raytracer_amd", "libopencl_wrap_hip_stamp.so")
os.environ["CLWRAP_STAMPS"] = "1"
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
cfg = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "c2"
strict = "--strict" in sys.argv
variant = int(sys.argv[sys.argv.index("--variant") + 1]) if "--variant" in sys.argv else 0
cam = pkg.CAMERA_RAYPNG
if cfg == "c2":
    sc, W, H, depth = scene.render_map_scene(), 1920, 1080, 4
elif cfg == "c4":
    sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
elif cfg == "ref800":
    sc, W, H, depth = scene.render_map_scene(), 800, 600, 15
else:
    sc, W, H, depth = scene.dielectric_field_scene(8), 4096, 4096, 8
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
r = Renderer(sc, textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=depth, strict=strict)
r.w.set_variant(variant)
r.look(**cam)
for _ in range(3):
    r.render(readback=False)
r.w.read_counters()
frames = 5
for _ in range(frames):
    r.render(readback=False)
r.w.read_counters()
raw = r.w.last_raw_counters[16:]
names = ["loop", "probe", "nearest", "resolve", "lights+samples", "shadow", "light_add", "bounce", "pop", "prolog"]
tot = sum(raw[:10])
waves = raw[10]
print(json.dumps(dict(config=cfg, strict=strict, variant=variant, waves_per_frame=waves // frames, cycles_per_wave=round(tot / max(waves, 1)),
                      shares={n: round(v / max(tot, 1), 4) for n, v in zip(names, raw)})))
r.release()
