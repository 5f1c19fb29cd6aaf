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
    """One gather per frame of every rank's interleaved 8-row bands to rank 0, with two frame slots so
    the gather of frame k overlaps the trace of frame k+1 (the work handle is waited on before the slot
    is reused).  Rank r owns bands r, r+world, r+2*world, ... (renderer `bands=(world, rank)`), so rank 0
    receives `world` equal blocks and the full frame is their interleave."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, dtype=torch.int32, staged_on_cpu=False):
        assert height % (8 * world) == 0, "interleaved bands need height % (8 * world) == 0"
        self.width, self.height, self.rank, self.world = width, height, rank, world
        self.px_rank = width * height // world
        self.staged = staged_on_cpu           # rehearsal with gloo: collectives on CPU copies
        dev = torch.device("cpu") if staged_on_cpu else device
        self.parts = [[torch.empty(self.px_rank, dtype=dtype, device=dev) for _ in range(world)] if rank == 0 else None
                      for _ in range(2)]
        self.pending = [None, None]

    def wait(self, slot: int) -> None:
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None

    def gather_async(self, slot: int, share: torch.Tensor) -> None:
        if self.world == 1:
            return
        send = share.cpu() if self.staged else share
        self.pending[slot] = dist.gather(send, gather_list=self.parts[slot], dst=0, async_op=True)

    def drain(self) -> None:
        self.wait(0)
        self.wait(1)

    def assemble(self, slot: int) -> torch.Tensor:
        """Rank 0: the full frame [height * width] of the last frame gathered into `slot`."""
        assert self.rank == 0
        nb = self.height // (8 * self.world)                      # bands per rank
        return torch.stack([p.view(nb, 8 * self.width) for p in self.parts[slot]], 1).reshape(-1)
