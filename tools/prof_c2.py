"""Profiling target: N frames of config C2 (render.map, 1920x1080, depth 4) through the C-ABI."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
strict = int(sys.argv[2]) if len(sys.argv) > 2 else 0
variant = int(sys.argv[3]) if len(sys.argv) > 3 else 0
depth = int(sys.argv[4]) if len(sys.argv) > 4 else 4
r = Renderer(scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(4096), 1920, 1080,
             depth=depth, strict=bool(strict))
r.w.set_variant(variant)
r.look(**pkg.CAMERA_RAYPNG)
r.w.timing_reset()
for _ in range(frames):
    r.render(readback=False)
n, ms = r.w.timing_get(1)
print(f"frames {n} avg kernel ms {ms / n:.4f}")
r.release()
