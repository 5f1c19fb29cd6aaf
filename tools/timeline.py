"""When does each tile's wave run inside one launch?  (DIAGNOSTIC build -DWT_TIMELINE=1, variant bit 512.)
   CLWRAP_LIB=.../libopencl_wrap_hip_tl.so python tools/timeline.py c2  -> busy wave slots per 5 % of the launch"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
shift = int(os.environ.get("CLWRAP_TIMELINE_SHIFT", "0"))      # ticks of 10 ns << shift (16 bits per stamp: 655 us at shift 0)
tick_us = 0.01 * (1 << shift)
strict = "--strict" in sys.argv
cam = pkg.CAMERA_RAYPNG
if cfg == "c2":
    sc, W, H, depth = scene.render_map_scene(), 1920, 1080, 4
elif cfg == "c3":
    sc, W, H, depth = scene.dielectric_field_scene(8), 4096, 4096, 8
    cam = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)
elif cfg == "ref800":
    sc, W, H, depth = scene.render_map_scene(), 800, 600, 15
else:
    sc, W, H, depth = scene.sphere_grid_scene(100, 100), 1920, 1080, 4
    cam = dict(origin=(0.0, 12.0, -10.0), look=(0.0, -0.45, 1.0), fov=90.0, focal=1.0)
r = Renderer(sc, textures.texture_layers(), textures.skybox_cross(4096), W, H, depth=depth, strict=strict)
r.look(**cam)
if "--tpt" in sys.argv: r.w.set_tpt(int(sys.argv[sys.argv.index("--tpt") + 1]), -1, -1)
for _ in range(6):
    r.render(readback=False)
r.w.sync()
r.w.set_variant(512)
r.w.timing_reset(); r.render(readback=False); r.w.sync(); n, ms = r.w.timing_get(1)
c = r.w.read_tile_costs().astype(np.int64)
if os.environ.get("CLWRAP_TIMELINE_EDGES") == "2":   # the prologue in three parts: staging | tile and pixel mapping | primary ray + loop set-up
    a, b_, c_ = (c >> 22) * 0.01, ((c >> 11) & 2047) * 0.01, (c & 2047) * 0.01
    print(json.dumps(dict(config=cfg, strict=strict, kernel_ms=round(ms / n, 4), staging_us=round(float(a.mean()), 2), mapping_us=round(float(b_.mean()), 2), primary_us=round(float(c_.mean()), 2))))
    r.release()
    sys.exit(0)
if os.environ.get("CLWRAP_TIMELINE_EDGES"):      # (prologue << 16 | epilogue) per tile instead of (start, end)
    pro, epi = (c >> 16) * 0.01, (c & 0xFFFF) * 0.01
    print(json.dumps(dict(config=cfg, strict=strict, kernel_ms=round(ms / n, 4), prologue_us=dict(mean=round(float(pro.mean()), 2), p50=round(float(np.median(pro)), 2), p99=round(float(np.percentile(pro, 99)), 2)),
                          epilogue_us=dict(mean=round(float(epi.mean()), 2), p50=round(float(np.median(epi)), 2), p99=round(float(np.percentile(epi, 99)), 2)))))
    r.release()
    sys.exit(0)
start, end = c >> 16, c & 0xFFFF
t0 = start.min() if (start.max() - start.min()) < 32768 else ((start + 32768) & 0xFFFF).min() - 32768
s = (start - t0) & 0xFFFF
e = (end - t0) & 0xFFFF
span = int(e.max())
edges = np.linspace(0, span, 21)
busy = [int(((s <= x) & (e > x)).sum()) for x in (edges[:-1] + edges[1:]) / 2]
dur = e - s
print(json.dumps(dict(config=cfg, strict=strict, kernel_ms=round(ms / n, 4), span_us=round(span * tick_us, 2), tiles=int(c.size),
                      mean_wave_us=round(float(dur.mean()) * tick_us, 2), max_wave_us=round(float(dur.max()) * tick_us, 2),
                      longest_us=[round(float(x) * tick_us, 1) for x in np.sort(dur)[-8:][::-1]], waves_over_50us=int((dur * tick_us > 50).sum()),
                      last_start_us=round(float(s.max()) * tick_us, 2), busy_waves_per_5pct=busy,
                      mean_busy=round(float(dur.sum()) / max(span, 1), 1))))
r.release()
