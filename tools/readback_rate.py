"""Frames/s of the blocking render + read-back call (what rayinteractive.c does per frame) with the pipelined
read-back on and off, at a few frame sizes / depths."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
sc, tex, sky = scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(4096)
for (W, H, depth) in ((1920, 1080, 4), (3840, 2160, 4), (1280, 1024, 15)):
    for on in (0, 1, 0, 1):
        r = Renderer(sc, tex, sky, W, H, depth=depth)
        r.w.set_pipeline(on)
        r.look(**pkg.CAMERA_RAYPNG)
        for _ in range(5): r.render()
        t = time.perf_counter()
        for _ in range(50): r.render()
        dt = (time.perf_counter() - t) / 50
        print(json.dumps(dict(frame=f"{W}x{H}", depth=depth, pipeline=on, ms_per_frame_with_readback=round(dt * 1e3, 4), fps=round(1 / dt, 1))), flush=True)
        r.release()
