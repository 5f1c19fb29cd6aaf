"""CRC32 of one rendered frame of a configuration (compare builds / environment settings): python tools/frame_crc.py c2 [--strict]"""
import os, sys, zlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
cfg = sys.argv[1]
strict = "--strict" in sys.argv
cam = pkg.CAMERA_RAYPNG
if cfg == "c2":
    sc, W, H, depth = scene.render_map_scene(), 1920, 1080, 4
elif cfg == "c3":
    sc, W, H, depth = scene.dielectric_field_scene(8), 2048, 2048, 8
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
elif cfg == "ref800":
    sc, W, H, depth = scene.render_map_scene(), 800, 600, 15
else:
    sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
r = Renderer(sc, textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=depth, strict=strict)
r.look(**cam)
r.render(readback=False); r.render(readback=False)
img = r.render()
print(cfg, "strict" if strict else "fast", "%08x" % zlib.crc32(img.tobytes()))
r.release()
