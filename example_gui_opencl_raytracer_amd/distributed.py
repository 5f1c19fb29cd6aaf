"""Row-strip sharding of the framebuffer across GPUs (one process per GPU).

The reference is single-device (one platform, one device, one queue; reference
src/opencl_wrap.c:26-34); pixels are independent work-items (raytracing.cl:23-37,194), so
the path shards by contiguous row strips with no data-path collective.  The one exchange is
a single gather of the finished strips to rank 0 -- RCCL over xGMI with backend "nccl",
gloo on CPU for the tests.  Work-item ids stay GLOBAL (renderer.Renderer / clw_ext_set_id_offset)
so the gathered image is bit-identical to a single-GPU render.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .renderer import strip_rows


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend: str | None = None):
    """Rendezvous from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torch.distributed.run sets them)."""
    rank, world, local_rank = env_rank_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def max_strip_rows(height: int, world: int) -> int:
    return max(strip_rows(height, world, r)[1] for r in range(world))


def gather_strips(strip: torch.Tensor, width: int, height: int, rank: int, world: int, dst: int = 0):
    """strip: int32 [rows_of_rank * width] (this rank's rows, row-major).  Returns the full
    int32 [height * width] image on `dst`, None elsewhere.  One gather; strips are padded to
    the tallest strip so every rank contributes the same count."""
    if world == 1:
        return strip
    pad_rows = max_strip_rows(height, world)
    send = strip
    if strip.numel() != pad_rows * width:
        send = torch.zeros(pad_rows * width, dtype=strip.dtype, device=strip.device)
        send[: strip.numel()] = strip
    parts = None
    if rank == dst:
        parts = [torch.empty(pad_rows * width, dtype=strip.dtype, device=strip.device) for _ in range(world)]
    dist.gather(send, gather_list=parts, dst=dst)
    if rank != dst:
        return None
    full = torch.empty(height * width, dtype=strip.dtype, device=strip.device)
    for r in range(world):
        r0, rows = strip_rows(height, world, r)
        full[r0 * width:(r0 + rows) * width] = parts[r][: rows * width]
    return full


class BandGatherer:
    """One gather per frame of every rank's interleaved 8-row bands to rank 0, with two frame slots so the
    gather of frame k overlaps the trace of frame k+1.

    What travels is RGB888: the framebuffer word is 0x00RRGGBB, its top byte is always zero, so each share is
    packed to 3 bytes per pixel first (a strided copy on a side stream) -- 25 % less on the xGMI links, which
    are what bounds a gather into one GPU.  Rank r owns bands r, r+world, ... (renderer `bands=(world, rank)`),
    so rank 0 receives `world` equal blocks and the full frame is their interleave (`assemble`)."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, staged_on_cpu=False):
        assert height % (8 * world) == 0, "interleaved bands need height % (8 * world) == 0"
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.px_rank = width * height // world
        self.staged = staged_on_cpu           # rehearsal with gloo: collectives on CPU copies
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda" and not staged_on_cpu
        cdev = self.device if self.on_gpu else torch.device("cpu")
        self.packed = [torch.empty(self.px_rank * 3, dtype=torch.uint8, device=cdev) for _ in range(2)]
        self.parts = [[torch.empty(self.px_rank * 3, dtype=torch.uint8, device=cdev) for _ in range(world)] if rank == 0 else None
                      for _ in range(2)]
        self.pending = [None, None]
        self.comm = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.ev_packed = [torch.cuda.Event() if self.on_gpu else None for _ in range(2)]
        self.used = [False, False]

    def before_render(self, slot: int) -> None:
        """Call before tracing into the slot's framebuffer again: its previous contents must have been packed."""
        if self.on_gpu and self.used[slot]:
            torch.cuda.current_stream(self.device).wait_event(self.ev_packed[slot])

    def submit(self, slot: int, share: torch.Tensor) -> None:
        """share: int32 [px_rank] framebuffer of this rank (0x00RRGGBB), just rendered on the current stream."""
        if self.world == 1:
            return
        bgr = share.view(torch.uint8).view(-1, 4)[:, :3]         # little-endian: B, G, R, 0
        if self.on_gpu:
            rendered = torch.cuda.Event()
            rendered.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(rendered)
                if self.pending[slot] is not None:
                    self.pending[slot].wait()                     # the gather that still reads packed[slot]
                self.packed[slot].view(-1, 3).copy_(bgr)
                self.ev_packed[slot].record(self.comm)
                self.pending[slot] = dist.gather(self.packed[slot], gather_list=self.parts[slot], dst=0, async_op=True)
            self.used[slot] = True
        else:
            if self.pending[slot] is not None:
                self.pending[slot].wait()
            self.packed[slot].view(-1, 3).copy_(bgr.cpu() if share.is_cuda else bgr)
            self.pending[slot] = dist.gather(self.packed[slot], gather_list=self.parts[slot], dst=0, async_op=True)

    def drain(self) -> None:
        for slot in (0, 1):
            if self.pending[slot] is not None:
                if self.on_gpu:
                    with torch.cuda.stream(self.comm):
                        self.pending[slot].wait()
                else:
                    self.pending[slot].wait()
                self.pending[slot] = None
        if self.on_gpu:
            self.comm.synchronize()

    def assemble(self, slot: int) -> torch.Tensor:
        """Rank 0: the full frame as int32 [height * width] (0x00RRGGBB) of the last frame gathered into `slot`."""
        assert self.rank == 0
        nb = self.height // (8 * self.world)                      # bands per rank
        rgb = torch.stack([p.view(nb, 8 * self.width * 3) for p in self.parts[slot]], 1).reshape(-1, 3).to(torch.int32)
        return (rgb[:, 2] << 16) | (rgb[:, 1] << 8) | rgb[:, 0]
