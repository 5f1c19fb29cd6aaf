"""Exploratory: init/release cycles must not leak device memory."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
sc, tex, sky = scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(4096)
big = scene.sphere_grid_scene(40, 40)
torch.cuda.init()
free0, total = torch.cuda.mem_get_info()
for i in range(200):
    r = Renderer(big if i % 2 else sc, tex, sky, 640, 360, depth=15 if i % 3 == 0 else 4, fuse=bool(i % 4))
    r.look(**pkg.CAMERA_RAYPNG)
    r.render()
    if i % 5 == 0:
        r.read_rays()
    r.release()
    if i % 50 == 49:
        free, _ = torch.cuda.mem_get_info()
        print(i + 1, "cycles: free memory changed by", (free0 - free) / 2**20, "MiB", flush=True)
