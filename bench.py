#!/usr/bin/env python3
"""Headline benchmark: Mrays/s of the per-pixel Whitted trace at 1920x1080, depth 4.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1], "C2"): the reference's demo scene (scenes/render.map,
regenerated from scene_dump.c's values), camera of raypng.c:17-21, 1920x1080, depth 4, textures
on (4 x 256^2 layers + 4096x3072 cube-cross skybox; procedural stand-ins for the PNG assets).
One step = one frame through the reference's call protocol (`cl_wrap_output(raygen)` +
`cl_wrap_output(raytracer)`, raypng.c:86-89) with every input resident in HBM and the framebuffer left in HBM.

What `--gpus N` measures (one process per GPU, no data-path collective while tracing, GLOBAL work-item ids, so the
assembled frame is bit-identical to a single-GPU render).  The HEADLINE is BASELINE.json's metric:
  strong  the FIXED 1920x1080 frame cut into N contiguous row strips (north_star's row-strip split)   <- value / ms_per_step
and the same JSON line carries, as named legs under "legs" (each with value, value_per_gpu, ms_per_step, transport):
  weak    every GPU keeps a 1920x1080 share of ONE 1920 x (1080*N) frame -- rank r owns every N-th 8-row band;
  c5      BASELINE config 5: 8192x8192, depth 4, N row strips, the single PNG written by rank 0 after the timed
          region (its time is reported separately).
At N = 1 strong and weak are the same single-GPU frame.  `--scaling weak` / `--config c5` make that leg the headline instead;
`--legs` picks which legs run.  Each step ends with ONE gather of the rows (packed to RGB888) into rank 0's frame buffer:
`--transport rccl-gather` (default: torch.distributed.gather on the nccl backend = RCCL's gather) or `--transport peer`
(every rank stores its share straight into rank 0's peer-mapped buffer; distributed.FrameGatherer), on a side stream and
double-buffered so it overlaps the next frame's trace.

Timing.  W warm-up steps, then EXACTLY K steps between barrier + synchronize on both sides, max over ranks -- and, because K
frames of 0.1 ms are a 2 ms region, that K-step loop is REPEATED until at least 50 ms have been timed: `ms_per_step` is the
median repeat's time / K (min / max of the repeats beside it).  The kernel's own duration is hipEvent-timed around EVERY
launch of one extra, untimed pass of K steps.

rays = path segments + shadow rays as the REFERENCE casts them, counted by the counting build of the kernel
outside the timed region; `rays_traced` / `value_traced` leave out the shadow rays the kernel draws but does not trace
(surfaces whose light terms are exactly zero).  rank 0 prints ONE JSON line; at N = 1 it also carries the strict (bit-exact
parity) build timed on the same workload, the kernel's roofline figures and the CPU baseline (the oracle on the host cores).
"""
from __future__ import annotations

import argparse
import json
import math
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
from example_gui_opencl_raytracer_amd import api, distributed as D, scene, textures  # noqa: E402
from example_gui_opencl_raytracer_amd.renderer import Renderer, strip_rows  # noqa: E402

DEPTH = 4
HBM_PEAK_GBPS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak fp32 vector
VALU_CYCLES, TRANS_CYCLES = 2.0, 8.0   # MI355X_MICROARCH.md cycle constants: v_fma_f32 wave64 2 cycles per SIMD, transcendentals 8
CLOCK_GHZ, SIMDS = 2.4, 1024
MIN_TIMED_S = 0.05              # the K-step loop is repeated until this much has been timed
MAX_REPEATS = 200
# fp32 operations of the reference's expression trees (DESIGN.md section "flop model")
FLOP = dict(sphere_test=34, plane_test=14, shadow_ray=40, light_shade=94, shaded_hit=73, sky=20, texel=25)


def profiled(counter, kernel="wt_fast::wt_trace<4>"):
    """Per-launch mean of a rocprofv3 PMC counter for the C2 trace kernel, from the newest committed summary of this same
    command whose `kernel_source_sha256_16` line names the kernels of the LOADED library (profiles/*_c2_rocprof_summary.md: one
    `### kernel ...` section per kernel, lines `COUNTER,value,dispatches`) -> (value, file) or None.  A summary taken from
    other kernel sources is never quoted: the figure would look current and not be."""
    import glob
    import re
    want = api.kernel_source_hash()
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_rocprof_summary.md"))):
        txt = open(f).read()
        m = re.search(r"kernel_source_sha256_16:\s*(\w+)", txt)
        if not m or m.group(1) != want:
            continue
        if "### kernel" in txt:
            secs = [sec for sec in txt.split("### kernel")[1:] if kernel in sec.split("\n", 1)[0]]
            if not secs:
                continue
            txt = secs[0]
        m = re.search(rf"^{counter},([0-9.]+)", txt, re.M)
        if m:
            best = (float(m.group(1)), os.path.basename(f))
    return best


def cpu_baseline(sc, tex, sky, W, H):
    """The oracle (plain-C restatement, OpenMP over pixels) on this box's host cores: ONE full C2 frame per run,
    3 runs, plus 1/8 frame on one thread; and BASELINE config C1 (render.map 640x480 depth 1, the reference's own
    CPU-runnable case) the same way.  Returns (dict for the JSON line, oracle counters of the C2 frame)."""
    from oracle.oracle_py import Oracle
    o = Oracle()
    threads = o.num_threads()

    def run(w, h, depth, reps):
        cam = o.camera(pkg.CAMERA_RAYPNG["origin"], pkg.CAMERA_RAYPNG["look"], 90.0, 1.0, w, h)
        times, cnt = [], None
        for _ in range(reps):
            t = time.perf_counter()
            _, _, cnt = o.render(cam, sc, tex, sky, depth, threads=0)
            times.append(time.perf_counter() - t)
        rows = max(h // 8, 1)
        t = time.perf_counter()
        _, _, c1 = o.render(cam, sc, tex, sky, depth, id_begin=0, id_end=rows * w, threads=1)
        t1 = time.perf_counter() - t
        return statistics.median(times), cnt, c1.rays / t1 / 1e6

    med, cnt, st = run(W, H, DEPTH, 3)
    base = dict(value=round(cnt.rays / med / 1e6, 3), unit="Mrays/s", cores=threads, kind="port",
                sample=f"3 x one full frame {W}x{H} depth {DEPTH} (median {med:.3f} s, {cnt.rays} rays)",
                single_thread_Mrays_s=round(st, 3), frames_per_s=round(1.0 / med, 3))
    m1, c1, st1 = run(640, 480, 1, 5)
    base["c1"] = dict(workload="C1: scenes/render.map 640x480 depth 1 (BASELINE configs[0], CPU path)", value=round(c1.rays / m1 / 1e6, 3),
                      unit="Mrays/s", cores=threads, single_thread_Mrays_s=round(st1, 3), frames_per_s=round(1.0 / m1, 2),
                      rays_per_pixel=round(c1.rays / (640 * 480), 4), sample=f"5 x one full frame (median {m1 * 1e3:.2f} ms)")
    return base, cnt


class Ctx:
    """What every leg shares: the process group, the device, the stream torch and the shim both launch on, scene and images."""


def frame_of(mode, world):
    """-> (W, H, layout) of a leg"""
    if mode == "c5":
        return 8192, 8192, "strips"
    if mode == "strong":
        return 1920, 1080, "strips"
    return 1920, 1080 * world, "bands"


class Leg:
    """One sharding mode of the benchmark: its renderers (two frame slots), framebuffers and gatherer."""

    def __init__(self, ctx, mode, strict=False):
        self.ctx, self.mode, self.strict = ctx, mode, strict
        world, rank = ctx.world, ctx.rank
        self.W, self.H, self.layout = frame_of(mode, world)
        if self.layout == "strips":
            first_row, rows = strip_rows(self.H, world, rank)
            shard = dict(first_row=first_row, rows=rows) if world > 1 else {}
        else:
            rows = self.H // world
            shard = dict(bands=(world, rank)) if world > 1 else {}
        self.rows, self.px_rank = rows, self.W * rows
        # two frame slots: the gather of frame k overlaps the trace of frame k+1
        self.fbs = [torch.zeros(max(self.px_rank, 1), dtype=torch.int32, device=ctx.dev) for _ in range(2)]
        self.rr = []
        for fb in self.fbs:
            r = Renderer(ctx.sc, ctx.tex, ctx.sky, self.W, self.H, depth=DEPTH, strict=strict, framebuffer_ptr=fb.data_ptr(), **shard)
            if not ctx.args.own_streams:
                r.w.set_stream(ctx.stream.cuda_stream)
            r.look(**pkg.CAMERA_RAYPNG)
            r.w.set_async(True)
            self.rr.append(r)
        self.gat = D.FrameGatherer(self.W, self.H, rank, world, ctx.dev, layout=self.layout, transport=ctx.args.transport,
                                   staged_on_cpu=ctx.args.rehearse and not ctx.args.rehearse_device_tensors)
        self.cnt = None

    def release(self):
        for r in self.rr:
            r.release()
        self.rr, self.fbs, self.gat = [], [], None

    def step(self, k):
        s = k & 1
        self.gat.before_render(s)                  # frame k-2's pixels have been packed for their gather
        self.rr[s].render(readback=False)          # raygen latch + trace launch (async, torch's stream)
        if self.ctx.world > 1:
            if self.ctx.args.rehearse:
                torch.cuda.current_stream().synchronize()
            self.gat.submit(s, self.fbs[s][:self.px_rank])      # pack to RGB888 + one gather into rank 0 on the side stream

    def timed(self, n):
        ctx = self.ctx
        ctx.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            self.step(k)
        self.gat.drain()
        torch.cuda.synchronize()
        ctx.barrier()
        el = time.perf_counter() - t0
        t = torch.tensor([el], dtype=torch.float64, device=ctx.dev)
        if ctx.world > 1:
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        return float(t.item())

    def kernel_pass(self, n):
        """One extra, UNTIMED pass of n steps with hipEvents around every trace launch -> (mean kernel ms, launches)."""
        for r in self.rr:
            r.w.timing_reset()
            r.w.set_timing_every(1)
        for k in range(n):
            self.step(k)
        self.gat.drain()
        torch.cuda.synchronize()
        launches, kms = 0, 0.0
        for r in self.rr:
            m, ms = r.w.timing_get(1)
            launches += m
            kms += ms
            r.w.set_timing_every(0)                # (no events in the timed loops)
        return kms / max(launches, 1), launches

    def count(self):
        """Ray count of one frame share (counting build, outside the timed region) -> summed over the ranks."""
        ctx = self.ctx
        self.rr[0].w.enable_counters(1)
        self.rr[0].render(readback=False)
        cnt = self.rr[0].w.read_counters()
        self.rr[0].w.enable_counters(0)
        tot = torch.tensor([cnt["segments"] + cnt["shadow_rays"], cnt["texel_fetches"] + cnt["sky_fetches"], cnt["lane_iters"],
                            cnt["wave_iters_x64"], cnt["segments"] + cnt["shadow_rays_traced"]], dtype=torch.float64, device=ctx.dev)
        if ctx.world > 1:
            torch.distributed.all_reduce(tot)
        self.cnt = cnt
        self.rays, self.rays_traced = int(tot[0].item()), int(tot[4].item())
        self.lane_util = float(tot[2].item() / max(tot[3].item(), 1.0))

    def run(self, steps, warmup):
        """-> dict of this leg's results (the same on every rank)."""
        ctx = self.ctx
        self.count()
        for r in self.rr:
            r.w.set_timing_every(0)
        for k in range(warmup):
            self.step(k)
        self.gat.drain()
        times = [self.timed(steps)]
        # repeat the K-step loop until MIN_TIMED_S have been timed; every rank takes the same count (the first time is a max over ranks)
        reps = min(MAX_REPEATS, max(1, math.ceil(MIN_TIMED_S / max(times[0], 1e-6))))
        for _ in range(reps - 1):
            times.append(self.timed(steps))
        el = statistics.median(times)
        k_ms, launches = self.kernel_pass(steps)
        world = ctx.world
        sharding = ("single GPU" if world == 1 else
                    (f"{world} contiguous row strips" if self.layout == "strips" else f"interleaved 8-row bands x{world}") +
                    f" + one gather of RGB888 rows into rank 0 per frame ({'rccl-gather' if self.gat.transport == 'gather' else self.gat.transport})")
        return dict(mode=self.mode, frame=f"{self.W}x{self.H}", value=round(self.rays * steps / el / 1e6, 1), unit="Mrays/s",
                    value_per_gpu=round(self.rays * steps / el / 1e6 / world, 1), ms_per_step=round(el / steps * 1e3, 4),
                    ms_per_step_min=round(min(times) / steps * 1e3, 4), ms_per_step_max=round(max(times) / steps * 1e3, 4),
                    repeats=len(times), steps=steps, timed_s=round(sum(times), 4),
                    frames_per_s=round(steps / el, 1), rays=self.rays, rays_traced=self.rays_traced,
                    value_traced=round(self.rays_traced * steps / el / 1e6, 1), rays_per_pixel=round(self.rays / (self.W * self.H), 4),
                    trace_kernel_ms=round(k_ms, 4), trace_launches_timed=launches, lane_utilisation=round(self.lane_util, 4),
                    transport="none" if world == 1 else ("rccl-gather" if self.gat.transport == "gather" else self.gat.transport),
                    sharding=sharding, scaling="weak" if self.mode == "weak" else "strong", elapsed_median_s=el)

    def last_frame(self, steps):
        """rank 0: the frame of the last step as uint32 [H * W]"""
        s = (steps - 1) & 1
        if self.ctx.world == 1:
            return self.fbs[s][:self.px_rank].cpu().numpy().view(np.uint32)
        return self.gat.assemble(s).cpu().numpy().view(np.uint32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="strong",
                    help="which sharding the HEADLINE is: strong = the fixed 1920x1080 frame in row strips (default, BASELINE.json's metric); "
                         "weak = 1920x1080 per GPU of a taller frame.  The other one is reported as a leg")
    ap.add_argument("--config", choices=("c2", "c5"), default="c2", help="c5: the 8192x8192 depth-4 row-strip leg (+ the single PNG) is the headline")
    ap.add_argument("--legs", default="all", help="comma list of strong,weak,c5 (or all / none): the legs reported beside the headline")
    ap.add_argument("--strict", type=int, default=0, help="1: the headline itself runs the strict arithmetic build")
    ap.add_argument("--transport", choices=("rccl-gather", "gather", "peer", "auto"), default=os.environ.get("BENCH_TRANSPORT", "rccl-gather"),
                    help="how shares reach rank 0 (N > 1): ONE torch.distributed.gather per frame (RCCL's gather on the nccl backend; the default), "
                         "or a store into rank 0's peer-mapped buffer (raises when the mapping cannot be set up), or try the second and fall back")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-strict-leg", action="store_true", help="skip timing the strict build next to the fast headline")
    ap.add_argument("--own-streams", action="store_true",
                    help="experiment: each frame slot launches on its own stream, so consecutive frames overlap")
    ap.add_argument("--rehearse", action="store_true",
                    help="N > 1 on a ONE-GPU box: gloo backend, all ranks on cuda:0 (checks the sharded path, not RCCL)")
    ap.add_argument("--rehearse-device-tensors", action="store_true",
                    help="with --rehearse: hand the GPU tensors to gloo directly (exercises the side-stream packing path)")
    ap.add_argument("--dump-png", default=None, help="rank 0 writes the headline leg's assembled frame here")
    args = ap.parse_args()
    if args.transport == "gather":
        args.transport = "rccl-gather"

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

    ctx = Ctx()
    ctx.args, ctx.rank, ctx.world = args, rank, world
    ctx.dev = torch.device("cuda", local_rank)
    ctx.sc = scene.render_map_scene()
    ctx.tex, ctx.sky = textures.texture_layers(), textures.skybox_cross(4096)
    # every launch of this rank goes to ONE explicit HIP stream that torch also treats as current, so the
    # shim's hipEvents, torch's collectives and the synchronisation around the timed region agree
    ctx.stream = torch.cuda.Stream(device=ctx.dev)
    torch.cuda.set_stream(ctx.stream)
    ctx.barrier = (lambda: torch.distributed.barrier()) if world > 1 else (lambda: None)

    head_mode = "c5" if args.config == "c5" else args.scaling
    if world == 1 and head_mode == "weak":
        head_mode = "strong"                    # the same single-GPU frame
    want = ["strong", "weak", "c5"] if args.legs == "all" else ([] if args.legs in ("none", "") else args.legs.split(","))
    want = [m for m in want if m in ("strong", "weak", "c5")]

    # ---- the headline leg
    head = Leg(ctx, head_mode, strict=bool(args.strict))
    res = head.run(args.steps, args.warmup)
    legs = {head_mode: res}
    c5 = head_mode == "c5"

    png = None
    if rank == 0 and (args.dump_png or c5):
        path = args.dump_png or os.path.join(ROOT, "gpurun_out", "c5.png")
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        t = time.perf_counter()
        full = head.last_frame(args.steps)
        t_host = time.perf_counter() - t
        t = time.perf_counter()
        api.write_png(path, full, head.W, head.H)
        t_png = time.perf_counter() - t
        if c5:   # C5: the single PNG, written by rank 0 from the gathered frame; timed separately
            png = dict(path=os.path.relpath(path, ROOT), bytes=os.path.getsize(path), assemble_and_readback_s=round(t_host, 4), write_s=round(t_png, 3),
                       note="outside the timed region: the 192 MB RGB image deflated (level 1) band by band on the host cores (png_codec.c)")
    ctx.barrier()

    # ---- N = 1 extras on the headline leg: PCIe-inclusive frame rates, a moving camera, the strict build on the same workload
    rr, px_rank, W, H = head.rr, head.px_rank, head.W, head.H
    host = np.empty(px_rank, np.uint32)
    rb, moving, strict_leg = [], None, None
    if rank == 0 and world == 1:
        rr[0].w.set_async(False)
        for _ in range(5):
            tt = time.perf_counter()
            rr[0].w.output(px_rank, 0, 0, 0, 0, None)
            rr[0].w.output(px_rank, host.nbytes, 1, 1, 10, host)
            rb.append(time.perf_counter() - tt)
        rr[0].w.set_async(True)
        if not c5:
            # the interactive loop's shape (rayinteractive.c:183-197): the camera turns a little EVERY frame, so every
            # frame's tile costs are re-sorted (on the shim's side stream, behind the frame)
            n = min(args.steps, 200)
            cams = [api.perspective(pkg.CAMERA_RAYPNG["origin"], (0.2 + 0.002 * np.sin(0.1 * i), 0.0005 * i, 1.0), 90.0, 1.0, W, H)
                    for i in range(n + 8)]
            for i in range(8):
                rr[i & 1].set_camera(cams[i])
                rr[i & 1].render(readback=False)
            torch.cuda.synchronize()
            for r in rr:
                r.w.timing_reset()
                r.w.set_timing_every(1)
            tt = time.perf_counter()
            for i in range(n):
                rr[i & 1].set_camera(cams[8 + i])
                rr[i & 1].render(readback=False)
            torch.cuda.synchronize()
            wall = time.perf_counter() - tt
            launches, kms = 0, 0.0
            for r in rr:
                m, ms = r.w.timing_get(1)
                launches += m
                kms += ms
            mk = kms / max(launches, 1)
            moving = dict(trace_kernel_ms=round(mk, 4), frames_per_s_kernel=round(1e3 / mk, 1), frames_per_s_wall=round(n / wall, 1), frames=n,
                          note="camera re-set before every frame (no read-back); the tile order is re-sorted for every frame on a side "
                               "stream, off the critical path; the wall rate is bound by this Python loop's eight API calls per frame")
            for r in rr:
                r.look(**pkg.CAMERA_RAYPNG)
    cnt = head.cnt
    head.release()
    if world == 1 and not args.strict and not args.no_strict_leg and not c5:
        sl = Leg(ctx, head_mode, strict=True)
        n = max(min(args.steps, 100), 2)
        sres = sl.run(n, 6)
        strict_leg = dict(ms_per_step=sres["ms_per_step"], ms_per_step_min=sres["ms_per_step_min"], ms_per_step_max=sres["ms_per_step_max"],
                          value=sres["value"], value_traced=sres["value_traced"], unit="Mrays/s", steps=n, repeats=sres["repeats"],
                          trace_kernel_ms=sres["trace_kernel_ms"], rays_traced=sres["rays_traced"],
                          note="the bit-exact parity build (clw_ext_set_strict) on the same workload")
        sl.release()

    # ---- the reference driver's OWN configuration (800x600, MAX_DEPTH 15: raypng.c:6-7, raytracing.cl:8): a deep launch, bound by the refraction
    #      trees of a few tiles -- with the tree-parallel tail (csrc/whitted_tpt.inc) and, beside it, with the per-lane loop alone (variant 16)
    ref_leg = None
    if rank == 0 and world == 1 and not args.strict and not c5 and not args.no_strict_leg:
        from example_gui_opencl_raytracer_amd.renderer import Renderer
        r = Renderer(ctx.sc, ctx.tex, ctx.sky, 800, 600, depth=15, strict=False)
        r.look(**pkg.CAMERA_RAYPNG)
        ref_leg = dict(frame="800x600", depth=15)
        for key, variant in (("ms_per_frame", 0), ("ms_per_frame_per_lane_loop_only", 16)):
            r.w.set_variant(variant)
            for _ in range(4):
                r.render(readback=False)
            if variant == 0:
                r.w.enable_counters(1); r.render(readback=False); c = r.w.read_counters(); r.w.enable_counters(0)
                ref_leg.update(rays=c["segments"] + c["shadow_rays"], tiles_finished_by_the_tail=c["tpt_tiles"], nodes_in_the_tail=c["tpt_nodes"])
                for _ in range(2):
                    r.render(readback=False)
            r.w.timing_reset(); r.w.set_timing_every(1); r.w.set_async(1)
            for _ in range(40):
                r.render(readback=False)
            r.w.sync(); nn, ms = r.w.timing_get(1); r.w.set_async(0)
            ref_leg[key] = round(ms / max(nn, 1), 4)
        ref_leg["value"] = round(ref_leg["rays"] / ref_leg["ms_per_frame"] / 1e3, 1)
        ref_leg["unit"] = "Mrays/s"
        ref_leg["frames_per_s"] = round(1e3 / ref_leg["ms_per_frame"], 1)
        r.release()

    # ---- the other legs (fewer steps: they are reported, not the headline)
    for m in want:
        if m in legs:
            continue
        if world == 1 and m == "weak":
            legs["weak"] = dict(legs["strong"], mode="weak", scaling="weak", note="N = 1: the same single-GPU frame as the strong leg") if "strong" in legs else None
            if legs["weak"] is None:
                del legs["weak"]
            continue
        n = max(2, min(args.steps, 20 if m == "c5" else 100))
        leg = Leg(ctx, m, strict=bool(args.strict))
        legs[m] = leg.run(n, min(args.warmup, 5))
        leg.release()
        ctx.barrier()

    if rank == 0:
        k_ms = res["trace_kernel_ms"]
        # algorithmic HBM bytes of ONE trace launch on this rank: the 4-byte packed pixel per
        # work-item + one 4-byte texel per texture / skybox fetch + the prepared geometry once
        bytes_launch = 4 * px_rank + 4 * (cnt["texel_fetches"] + cnt["sky_fetches"]) + 16 * (4 + 2 * 2 + 2 * 3)
        ach = bytes_launch / (k_ms * 1e-3) / 1e9
        single_c2 = world == 1 and not c5
        khash = api.kernel_source_hash()
        fe, wr = (profiled("FETCH_SIZE"), profiled("WRITE_SIZE")) if single_c2 and not args.strict else (None, None)
        traffic = int((fe[0] + wr[0]) * 1024) if fe and wr else None
        strong = res["scaling"] == "strong"
        workload = ("C5: scenes/render.map 8192x8192 depth 4" if c5 else "C2: scenes/render.map (regenerated), camera raypng.c:17-21, 1920x1080"
                    + (" per GPU" if not strong else "") + ", depth 4, 2 soft-shadow samples, 4x256^2 textures + 4096x3072 skybox (procedural)")
        size = "8192x8192" if c5 else "1920x1080"
        line = {
            "metric": f"Mrays/s at {size} depth 4, aggregate over n_gpus; rays = path segments + shadow rays as the reference casts them "
                      "(value_traced: only those the kernel really traces)",
            "value": res["value"], "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": res["scaling"], "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "frame": res["frame"], "sharding": res["sharding"],
                       "arithmetic": "strict" if args.strict else "fast", "rays_per_pixel": res["rays_per_pixel"]},
            "ms_per_step_min": res["ms_per_step_min"], "ms_per_step_max": res["ms_per_step_max"], "repeats": res["repeats"],
            "timed_s": res["timed_s"],
            "value_per_gpu": res["value_per_gpu"],
            "rays": res["rays"], "rays_traced": res["rays_traced"], "value_traced": res["value_traced"],
            "frames_per_s": res["frames_per_s"],
            "frames_per_s_with_readback": round(1.0 / statistics.median(rb), 1) if rb else None,
            "trace_kernel_ms": k_ms, "trace_launches_timed": res["trace_launches_timed"],
            "lane_utilisation": res["lane_utilisation"],
            "kernel_source_sha256_16": khash,
            "legs": {m: {k: v for k, v in leg.items() if k != "elapsed_median_s"} for m, leg in legs.items()},
            "roofline": {"bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBPS, 5), "traffic": traffic,
                         "algorithmic_bytes": bytes_launch, "traffic_source": fe[1] if traffic else None,
                         "traffic_note": None if traffic or not single_c2 else "no committed rocprofv3 summary was taken from the loaded kernels (kernel_source_sha256_16)",
                         "note": "algorithmic bytes/launch = 4 B x pixels + 4 B x texel fetches + geometry; the path is "
                                 "VALU-bound (723-byte scene), see roofline_valu / roofline_valu_issue"},
        }
        if strict_leg:
            line["strict"] = strict_leg
        if moving:
            line["moving_camera"] = moving
        if ref_leg:
            line["reference_config"] = ref_leg
        if png:
            line["png"] = png
        vi, vt = (profiled("SQ_INSTS_VALU"), profiled("SQ_INSTS_VALU_TRANS_F32")) if single_c2 and not args.strict else (None, None)
        if vi:
            # VALU issue bound from the guide's cycle constants: a wave64 VALU instruction occupies its SIMD for 2 cycles,
            # a transcendental (v_rcp/rsq/sqrt/sin/cos/log/exp) for 8; 1 024 SIMDs at 2.4 GHz
            trans = vt[0] if vt else 0.0
            cycles = (vi[0] - trans) * VALU_CYCLES + trans * TRANS_CYCLES
            floor_ms = cycles / SIMDS / (CLOCK_GHZ * 1e9) * 1e3
            line["roofline_valu_issue"] = {"bound": "valu_issue", "achieved": round(floor_ms / k_ms, 4), "peak": 1.0,
                                           "unit": "fraction of SIMD issue cycles", "frac": round(floor_ms / k_ms, 4),
                                           "issue_floor_ms": round(floor_ms, 4), "valu_instructions_per_launch": int(vi[0]),
                                           "transcendental_per_launch": int(trans), "cycles_plain": VALU_CYCLES, "cycles_transcendental": TRANS_CYCLES,
                                           "source": vi[1]}
        if single_c2 and not args.no_cpu_baseline:
            base, oc = cpu_baseline(ctx.sc, ctx.tex, ctx.sky, W, H)
            line["cpu_baseline"] = base
            flops = (FLOP["sphere_test"] * oc.sphere_tests + FLOP["plane_test"] * oc.plane_tests + FLOP["shadow_ray"] * oc.shadow_rays
                     + FLOP["light_shade"] * oc.shaded_hits * len(ctx.sc.lights) + FLOP["shaded_hit"] * oc.shaded_hits
                     + FLOP["sky"] * oc.sky_fetches + FLOP["texel"] * oc.texel_fetches)
            tf = flops / (k_ms * 1e-3) / 1e12
            line["roofline_valu"] = {"bound": "valu_fp32", "achieved": round(tf, 3), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": round(tf / VALU_PEAK_TFLOPS, 4), "flops_per_launch": int(flops),
                                     "note": "fp32 operations of the reference's expression trees, counted by the oracle"}
            line["gpu_rays_vs_oracle_rays"] = [res["rays"], oc.rays]
        print(json.dumps(line), flush=True)

    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
