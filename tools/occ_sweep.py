"""Deep-build flavour sweep: kernel ms of render.map at several sizes / depths with the high-occupancy flavour forced
(CLWRAP_OCC_TILES_PER_DEPTH=0) and forbidden (variant 64)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, json, time
sys.path.insert(0, %r)
import torch
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
W, H, depth, variant = map(int, sys.argv[1:5])
r = Renderer(scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(2048), W, H, depth=depth)
r.w.set_variant(variant); r.look(**pkg.CAMERA_RAYPNG)
for _ in range(3): r.render(readback=False)
r.w.timing_reset(); r.w.set_async(1)
for _ in range(20): r.render(readback=False)
r.w.sync(); n, ms = r.w.timing_get(1)
print(json.dumps(dict(frame=f"{W}x{H}", depth=depth, occ=(variant == 0), kernel_ms=round(ms / n, 4))))
''' % ROOT
for (W, H) in ((1280, 720), (1920, 1080), (2560, 1440), (3840, 2160)):
    for depth in (6, 8, 15):
        for variant in (0, 64):
            env = dict(os.environ, CLWRAP_OCC_TILES_PER_DEPTH="0")
            out = subprocess.run([sys.executable, "-c", code, str(W), str(H), str(depth), str(variant)], env=env, capture_output=True, text=True)
            print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
