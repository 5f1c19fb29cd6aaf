"""The tree-parallel tail of deep launches (csrc/whitted_tpt.inc) against the per-lane loop: same pixels, same counters, and what it costs.
   python tools/tpt_check.py [quick|time]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import example_gui_opencl_raytracer_amd as pkg
from example_gui_opencl_raytracer_amd import scene, textures
from example_gui_opencl_raytracer_amd.renderer import Renderer

mode = sys.argv[1] if len(sys.argv) > 1 else "quick"
tex, sky = textures.texture_layers(), textures.skybox_cross(1024)
C3CAM = dict(origin=(3.5, 3.0, -6.0), look=(0.0, -2.5, 9.5), fov=90.0, focal=1.0)

def frame(r, tpt, counters=True):
    r.w.set_tpt(tpt, 1 if tpt == 64 else -1, -1)
    r.w.set_variant(16 if tpt == 0 else 0)
    for _ in range(2): r.render(readback=False)      # the dispatch order (with its split tiles) follows the costs of the frames before
    if counters: r.w.enable_counters(1)
    img = r.render()
    c = r.w.read_counters() if counters else None
    if counters: r.w.enable_counters(0)
    return img, c

if mode == "quick":
    cases = [("render.map 320x240 d15", scene.render_map_scene(), 320, 240, 15, pkg.CAMERA_RAYPNG),
             ("render.map 200x152 d8", scene.render_map_scene(), 200, 152, 8, pkg.CAMERA_RAYPNG),
             ("glass field 256x256 d8", scene.dielectric_field_scene(8), 256, 256, 8, C3CAM),
             ("glass field 160x160 d15", scene.dielectric_field_scene(4), 160, 160, 15, C3CAM)]
    bad = 0
    for name, sc, W, H, depth, cam in cases:
        for strict in (1, 0):
            r = Renderer(sc, tex, sky, W, H, depth=depth, strict=bool(strict))
            r.look(**cam)
            ref, cref = frame(r, 0)
            for tpt in (8, 24, 64):
                img, c = frame(r, tpt)
                same = bool(np.array_equal(img, ref))
                keys = ("segments", "shadow_rays", "light_probes", "sky_fetches", "texel_fetches", "pushes", "shadow_rays_traced")
                csame = all(c[k] == cref[k] for k in keys)
                print(f"{name} strict={strict} tpt_max={tpt}: pixels {'==' if same else '!= (%d differ)' % int((img != ref).sum())}, counters {'==' if csame else '!='}, "
                      f"tail tiles {c['tpt_tiles']} gave up {c['tpt_gave_up']} nodes {c['tpt_nodes']} lane_util {c['lane_iters'] / max(c['wave_iters_x64'], 1):.3f} (loop only {cref['lane_iters'] / max(cref['wave_iters_x64'], 1):.3f})", flush=True)
                bad += (not same) + (not csame)
            r.release()
    print("FAILED" if bad else "OK", bad)
    sys.exit(1 if bad else 0)

# timing
cfgs = {"ref800": (scene.render_map_scene(), 800, 600, 15, pkg.CAMERA_RAYPNG), "c3": (scene.dielectric_field_scene(8), 4096, 4096, 8, C3CAM),
        "hd15": (scene.render_map_scene(), 1920, 1080, 15, pkg.CAMERA_RAYPNG), "c3s": (scene.dielectric_field_scene(8), 2048, 2048, 8, C3CAM)}
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["ref800", "c3s"]
tpts = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 8, 16, 24, 32, 48, 64]
for cfg in which:
    sc, W, H, depth, cam = cfgs[cfg]
    for strict in (0, 1) if "--strict" in sys.argv else (0,):
        r = Renderer(sc, tex, sky, W, H, depth=depth, strict=bool(strict))
        r.look(**cam)
        ref = None
        for tpt in tpts:
            img, c = frame(r, tpt)
            if ref is None: ref = img
            for _ in range(3): r.render(readback=False)
            r.w.read_counters()                          # (clears the phase clock's words: CLWRAP_TPT_CLOCK=1 books them in every build)
            r.w.timing_reset(); r.w.set_async(1)
            nfr = 30 if cfg != "c3" else 8
            for _ in range(nfr): r.render(readback=False)
            r.w.sync(); n, ms = r.w.timing_get(1); r.w.set_async(0)
            pc = r.w.read_counters()
            if os.environ.get("CLWRAP_TPT_CLOCK"):
                c = dict(c, tpt_phase_us=[round(x / nfr, 1) for x in pc["tpt_phase_us"]], tpt_phase_max_us=pc["tpt_phase_max_us"], tpt_longest_us=pc["tpt_longest_us"])
            print(json.dumps(dict(config=cfg, strict=strict, tpt_max=tpt, kernel_ms=round(ms / n, 4), same=bool(np.array_equal(img, ref)), tail_tiles=c["tpt_tiles"],
                                  gave_up=c["tpt_gave_up"], nodes=c["tpt_nodes"], phase_us=c["tpt_phase_us"], phase_max_us=c["tpt_phase_max_us"], batches=c["tpt_batches"], max_batches=c["tpt_max_batches"], max_nodes=c["tpt_max_nodes"], longest_us=c["tpt_longest_us"], lane_util=round(c["lane_iters"] / max(c["wave_iters_x64"], 1), 4))), flush=True)
        r.release()
