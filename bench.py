#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-pixel Whitted trace at 1920x1080, depth 4.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): the reference's demo scene (scenes/render.map,
regenerated from scene_dump.c's values), camera of raypng.c:17-21, 1920x1080, depth 4, textures
on (4 x 256^2 layers + 4096x3072 cube-cross skybox; procedural stand-ins for the PNG assets, which
cannot travel to the GPU box).  One step = one frame through the reference's call protocol
(`cl_wrap_output(raygen)` + `cl_wrap_output(raytracer)`, raypng.c:86-89) with every input
resident in HBM and the framebuffer left in HBM.

N > 1 (weak scaling): every GPU keeps a 1920x1080 share of ONE 1920 x (1080*N) frame -- rank r
owns every N-th 8-row band (interleaved so the shares cost the same) with GLOBAL work-item ids,
so the assembled frame is bit-identical to a single-GPU render.  No data-path collective while
tracing; each step ends with one gather of the bands (packed to RGB888) to rank 0 over RCCL/xGMI on
a side stream, double-buffered so it overlaps the next frame's trace.

rays = path segments + shadow rays, counted by the counting build of the kernel
outside the timed region.  rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import example_gui_opencl_raytracer_amd as pkg  # noqa: E402
from example_gui_opencl_raytracer_amd import distributed as D, scene, textures  # noqa: E402
from example_gui_opencl_raytracer_amd.renderer import Renderer  # noqa: E402

W, H_PER_GPU, DEPTH = 1920, 1080, 4
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak fp32 vector
VALU_SLOT_NS = 1.04             # measured: one plain wave64 VALU instruction per SIMD (tools/ubench/valu_rates.hip)
TRANS_WEIGHT = 3.1              # measured: a transcendental occupies 3.1 such slots
TIMING_EVERY = 8                # hipEvents around every 8th trace launch of a frame slot (an event record between two
                                # kernels delays the second: 5 us per frame if every launch is timed)
# fp32 operations of the reference's expression trees (DESIGN.md section "flop model")
FLOP = dict(sphere_test=34, plane_test=14, shadow_ray=40, light_shade=94, shaded_hit=73, sky=20, texel=25)


def profiled_traffic():
    """HBM bytes per trace launch from the committed rocprofv3 PMC passes of this same command
    (profiles/*_rocprof_summary.md: FETCH_SIZE and WRITE_SIZE, KiB per dispatch, separate passes).  The
    guide's x2 FETCH_SIZE correction is for wide coalesced streams; this kernel's reads are 4-byte texel
    gathers, so the raw counter is used (uncalibrated for that width)."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary.md"))):
        txt = open(f).read()
        fe, wr = re.search(r"^FETCH_SIZE,([0-9.]+)", txt, re.M), re.search(r"^WRITE_SIZE,([0-9.]+)", txt, re.M)
        if fe and wr:
            best = (int((float(fe.group(1)) + float(wr.group(1))) * 1024), os.path.basename(f))
    return best


def profiled_valu_instructions():
    """(SQ_INSTS_VALU, SQ_INSTS_VALU_TRANS_F32) per trace launch (wave-level instructions) from the committed PMC passes."""
    import glob
    import re
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary.md"))):
        txt = open(f).read()
        m = re.search(r"^SQ_INSTS_VALU,([0-9.]+)", txt, re.M)
        t = re.search(r"^SQ_INSTS_VALU_TRANS_F32,([0-9.]+)", txt, re.M)
        if m:
            best = (float(m.group(1)), float(t.group(1)) if t else 0.0, os.path.basename(f))
    return best


def cpu_baseline(sc, tex, sky):
    """The oracle (plain-C restatement, OpenMP over pixels) on this box's host cores: ONE full C2
    frame per run, 3 runs.  Returns (dict for the JSON line, oracle counters of the frame)."""
    from oracle.oracle_py import Oracle
    o = Oracle()
    cam = o.camera(pkg.CAMERA_RAYPNG["origin"], pkg.CAMERA_RAYPNG["look"], 90.0, 1.0, W, H_PER_GPU)
    threads = o.num_threads()
    times, cnt = [], None
    for _ in range(3):
        t = time.perf_counter()
        _, _, cnt = o.render(cam, sc, tex, sky, DEPTH, threads=0)
        times.append(time.perf_counter() - t)
    rows = H_PER_GPU // 8
    t = time.perf_counter()
    _, _, c1 = o.render(cam, sc, tex, sky, DEPTH, id_begin=0, id_end=rows * W * 1, threads=1)
    t1 = time.perf_counter() - t
    med = statistics.median(times)
    return dict(value=round(cnt.rays / med / 1e6, 3), unit="Mrays/s", cores=threads, kind="port",
                sample=f"3 x one full frame 1920x1080 depth 4 (median {med:.3f} s, {cnt.rays} rays)",
                single_thread_Mrays_s=round(c1.rays / t1 / 1e6, 3),
                frames_per_s=round(1.0 / med, 3)), cnt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--strict", type=int, default=0, help="1: strict arithmetic build (parity build)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--own-streams", action="store_true",
                    help="experiment: each frame slot launches on its own stream, so consecutive frames overlap")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a ONE-GPU box: gloo backend, all ranks on cuda:0 (checks the sharded path, not RCCL)")
    ap.add_argument("--rehearse-device-tensors", action="store_true",
                    help="with --rehearse: hand the GPU tensors to gloo directly (exercises the side-stream packing path)")
    ap.add_argument("--dump-png", default=None, help="rank 0 writes the assembled frame here")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the trace path has no CPU fallback")
    _, _, local_rank = D.env_rank_world()
    if args.rehearse:
        local_rank = 0          # every rank on cuda:0, gathers staged through the CPU (no RCCL on one GPU)
    torch.cuda.set_device(local_rank)          # before the process group: RCCL binds to the current device
    os.environ["CLWRAP_DEVICE"] = str(local_rank)   # the shim binds to the same GPU explicitly
    rank, world, _ = D.init_process_group("gloo" if args.rehearse else None)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    dev = torch.device("cuda", local_rank)

    sc = scene.render_map_scene()
    tex, sky = textures.texture_layers(), textures.skybox_cross(4096)
    H = H_PER_GPU * world
    px_rank = W * H_PER_GPU
    # every launch of this rank goes to ONE explicit HIP stream that torch also treats as current, so the
    # shim's hipEvents, torch's collectives and the synchronisation around the timed region agree
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)

    # two frame slots: the gather of frame k overlaps the trace of frame k+1
    fbs = [torch.zeros(px_rank, dtype=torch.int32, device=dev) for _ in range(2)]
    rr = []
    for fb in fbs:
        r = Renderer(sc, tex, sky, W, H, depth=DEPTH, strict=bool(args.strict),
                     bands=(world, rank) if world > 1 else None, framebuffer_ptr=fb.data_ptr())
        if not args.own_streams:
            r.w.set_stream(stream.cuda_stream)
        r.look(**pkg.CAMERA_RAYPNG)
        r.w.set_async(True)
        rr.append(r)
    gat = D.BandGatherer(W, H, rank, world, dev, staged_on_cpu=args.rehearse and not args.rehearse_device_tensors)

    def step(k):
        s = k & 1
        gat.before_render(s)                       # frame k-2's pixels have been packed for their gather
        rr[s].render(readback=False)               # raygen latch + trace launch (async, torch's stream)
        if world > 1:
            if args.rehearse:
                torch.cuda.current_stream().synchronize()
            gat.submit(s, fbs[s])                  # pack to RGB888 + gather on the side stream

    def drain():
        gat.drain()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # ---- ray count of one frame share (counting build, outside the timed region)
    rr[0].w.enable_counters(1)
    rr[0].render(readback=False)
    cnt = rr[0].w.read_counters()
    rr[0].w.enable_counters(0)
    rays_rank = cnt["segments"] + cnt["shadow_rays"]
    tot = torch.tensor([rays_rank, cnt["texel_fetches"] + cnt["sky_fetches"], cnt["lane_iters"], cnt["wave_iters_x64"]],
                       dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tot)
    rays_step, fetch_step = int(tot[0].item()), int(tot[1].item())

    for k in range(args.warmup):
        step(k)
    drain()
    for r in rr:
        r.w.timing_reset()
        r.w.set_timing_every(TIMING_EVERY)
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(t.item())

    launches, kms = 0, 0.0
    for r in rr:
        n, ms = r.w.timing_get(1)
        launches += n
        kms += ms
    kernel_ms = kms / max(launches, 1)

    # PCIe-inclusive frame rate (what rayinteractive's loop sees: launch + wait + blocking read-back)
    host = np.empty(px_rank, np.uint32)
    rb = []
    if rank == 0:
        rr[0].w.set_async(False)
        for _ in range(5):
            tt = time.perf_counter()
            rr[0].w.output(px_rank, 0, 0, 0, 0, None)
            rr[0].w.output(px_rank, host.nbytes, 1, 1, 10, host)
            rb.append(time.perf_counter() - tt)

    if rank == 0 and args.dump_png:
        from example_gui_opencl_raytracer_amd import api
        if world == 1:
            full = fbs[(args.steps - 1) & 1].cpu().numpy().view(np.uint32)
        else:
            full = gat.assemble((args.steps - 1) & 1).cpu().numpy().view(np.uint32)
        api.write_png(args.dump_png, full, W, H)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = rays_step * args.steps / elapsed / 1e6
        # algorithmic HBM bytes of ONE trace launch on this rank: the 4-byte packed pixel per
        # work-item + one 4-byte texel per texture / skybox fetch + the prepared geometry once
        bytes_launch = 4 * px_rank + 4 * (cnt["texel_fetches"] + cnt["sky_fetches"]) + 16 * (4 + 2 * 2 + 2 * 3)
        ach = bytes_launch / (kernel_ms * 1e-3) / 1e9
        traffic = profiled_traffic() if world == 1 else None
        line = {
            "metric": "Mrays/s (path segments + shadow rays) at 1920x1080 depth 4 per GPU",
            "value": round(value, 1), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C2: scenes/render.map (regenerated), camera raypng.c:17-21, 1920x1080 per GPU, "
                                   "depth 4, 2 soft-shadow samples, 4x256^2 textures + 4096x3072 skybox (procedural)",
                       "frame": f"{W}x{H}", "sharding": "single GPU" if world == 1 else f"interleaved 8-row bands x{world} + gather to rank 0",
                       "arithmetic": "strict" if args.strict else "fast", "rays_per_pixel": round(rays_step / (W * H), 4)},
            "frames_per_s": round(args.steps / elapsed, 1),
            "frames_per_s_with_readback": round(1.0 / statistics.median(rb), 1) if rb else None,
            "trace_kernel_ms": round(kernel_ms, 4), "trace_launches_timed": launches,
            "lane_utilisation": round(float(tot[2].item() / max(tot[3].item(), 1.0)), 4),
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": traffic[0] if traffic else None,
                         "algorithmic_bytes": bytes_launch, "traffic_source": traffic[1] if traffic else None,
                         "note": "algorithmic bytes/launch = 4 B x pixels + 4 B x texel fetches + geometry; the path is "
                                 "VALU-bound (723-byte scene), see roofline_valu"},
        }
        vi = profiled_valu_instructions() if world == 1 else None
        if vi:
            # issue slots: a plain wave64 VALU instruction takes 1.04 ns of a SIMD, a transcendental (v_rcp/rsq/sqrt/
            # sin/cos/log/exp) 3.1x that (tools/ubench/valu_rates.hip on this chip, profiles/*_valu_rates.txt); 1 024 SIMDs
            slots = vi[0] + (TRANS_WEIGHT - 1.0) * vi[1]
            peak = 1024 / VALU_SLOT_NS / 1e3                     # T issue slots / s
            got = slots / (kernel_ms * 1e-3) / 1e12
            line["roofline_valu_issue"] = {"bound": "valu_issue", "achieved": round(got, 4), "peak": round(peak, 4),
                                           "unit": "T issue slots/s", "frac": round(got / peak, 4),
                                           "valu_instructions_per_launch": int(vi[0]), "transcendental_per_launch": int(vi[1]),
                                           "slot_ns": VALU_SLOT_NS, "transcendental_weight": TRANS_WEIGHT, "source": vi[2]}
        if world == 1 and not args.no_cpu_baseline:
            base, oc = cpu_baseline(sc, tex, sky)
            line["cpu_baseline"] = base
            flops = (FLOP["sphere_test"] * oc.sphere_tests + FLOP["plane_test"] * oc.plane_tests + FLOP["shadow_ray"] * oc.shadow_rays
                     + FLOP["light_shade"] * oc.shaded_hits * len(sc.lights) + FLOP["shaded_hit"] * oc.shaded_hits
                     + FLOP["sky"] * oc.sky_fetches + FLOP["texel"] * oc.texel_fetches)
            tf = flops / (kernel_ms * 1e-3) / 1e12
            line["roofline_valu"] = {"bound": "valu_fp32", "achieved": round(tf, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(tf / VALU_PEAK_TFLOPS, 4), "flops_per_launch": int(flops),
                                     "note": "fp32 operations of the reference's expression trees, counted by the oracle"}
            line["gpu_rays_vs_oracle_rays"] = [rays_step, oc.rays]
        print(json.dumps(line), flush=True)

    for r in rr:
        r.release()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
