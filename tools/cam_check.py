"""Exploratory: fast-mode parity statistics for the four test cameras (render.map, 128x96, depth 15)."""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
from oracle.oracle_py import Oracle
o = Oracle(); sc = scene.render_map_scene(); tex = textures.texture_layers(); sky = textures.skybox_cross(512)
cams = [((0.8, 2.5, -8.0), (0.0, 0.0, 1.0), 90.0), ((-3.0, 0.6, 0.5), (1.0, 0.05, 0.3), 70.0),
        ((0.8, 0.8, 1.5), (0.3, -0.2, 1.0), 100.0), ((1.0, 9.0, 1.0), (0.01, -1.0, 0.02), 60.0)]
for (org, look, fov) in cams:
    for depth in (1, 2, 15):
        w, h = 128, 96
        want, wrgb, _ = o.render(o.camera(org, look, fov, 1.0, w, h), sc, tex, sky, depth, want_rgb=True)
        r = Renderer(sc, tex, sky, w, h, depth=depth, strict=False)
        r.look(origin=org, look=look, fov=fov, focal=1.0)
        got, rgb = r.render_rgb(); r.release()
        g = np.stack([(got >> 16) & 255, (got >> 8) & 255, got & 255], 1).astype(int)
        x = np.stack([(want >> 16) & 255, (want >> 8) & 255, want & 255], 1).astype(int)
        d = np.abs(g - x).max(1)
        fe = np.abs(rgb - wrgb).max(1)
        bad = np.nonzero(d > 0)[0]
        print(org, depth, dict(exact=float((d == 0).mean()), le1=float((d <= 1).mean()), maxd=int(d.max()), rgb_le_1e4=float((fe <= 1e-4).mean()),
                               rgb_maxerr=float(fe.max())), "first bad:", [(int(i), g[i].tolist(), x[i].tolist(), rgb[i].tolist(), wrgb[i].tolist()) for i in bad[:2]], flush=True)
