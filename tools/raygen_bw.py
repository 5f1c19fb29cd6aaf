"""The API-faithful two-kernel path (CLWRAP_FUSE=0): raygen kernel bandwidth + trace-from-buffer time at C2."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
for (W, H) in ((1920, 1080), (4096, 4096)):
    r = Renderer(scene.render_map_scene(), textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=4, fuse=False)
    r.look(**pkg.CAMERA_RAYPNG)
    for _ in range(3):
        r.render(readback=False)
    r.w.timing_reset(); r.w.set_async(1)
    for _ in range(50):
        r.render(readback=False)
    r.w.sync()
    n0, ms0 = r.w.timing_get(0); n1, ms1 = r.w.timing_get(1)
    px = W * H
    print(json.dumps(dict(frame=f"{W}x{H}", raygen_ms=round(ms0 / n0, 4), raygen_GBps=round(64 * px / (ms0 / n0) / 1e6, 1),
                          raygen_frac_of_8TBps=round(64 * px / (ms0 / n0) / 1e6 / 8000, 3), trace_from_buffer_ms=round(ms1 / n1, 4))))
    r.release()
