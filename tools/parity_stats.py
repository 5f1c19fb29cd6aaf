"""Fast-build parity statistics vs the oracle on a few frames (exploratory; run once per library via CLWRAP_LIB)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (loads libamdhip64 first)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
from oracle.oracle_py import Oracle

o = Oracle()
tex, sky = textures.texture_layers(), textures.skybox_cross(1024)
CAM = pkg.CAMERA_RAYPNG
cases = [("render.map", scene.render_map_scene(), CAM, 1280, 720, 4),
         ("render.map", scene.render_map_scene(), CAM, 800, 600, 15),
         ("glass field", scene.dielectric_field_scene(8), dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0), 768, 768, 8)]
for name, sc, cam, W, H, depth in cases:
    want, wrgb, cnt = o.render(o.camera(cam["origin"], cam["look"], cam["fov"], cam["focal"], W, H), sc, tex, sky, depth, want_rgb=True)
    r = Renderer(sc, tex, sky, W, H, depth=depth, strict=False)
    r.look(**cam)
    got, rgb = r.render_rgb()
    g = np.stack([(got >> 16) & 255, (got >> 8) & 255, got & 255], 1).astype(int)
    w = np.stack([(want >> 16) & 255, (want >> 8) & 255, want & 255], 1).astype(int)
    d = np.abs(g - w).max(1)
    fd = np.abs(rgb - wrgb).max(1)
    print(json.dumps(dict(lib=os.environ.get("CLWRAP_LIB", "default")[-12:], case=name, frame=f"{W}x{H}", depth=depth,
                          exact=round(float((d == 0).mean()), 6), le1=round(float((d <= 1).mean()), 6), maxd=int(d.max()),
                          rgb_le_1e4=round(float((fd <= 1e-4).mean()), 6), rgb_mean_abs=float(np.nanmean(fd)))), flush=True)
    r.release()
