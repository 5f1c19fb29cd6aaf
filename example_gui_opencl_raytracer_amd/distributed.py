"""Row-strip sharding of the framebuffer across GPUs (one process per GPU).

The reference is single-device (one platform, one device, one queue; reference
src/opencl_wrap.c:26-34); pixels are independent work-items (raytracing.cl:23-37,194), so
the path shards by rows (contiguous strips, or interleaved 8-row bands for balance) with no data-path
collective.  The one exchange is a single gather of the finished rows (packed to RGB888) to rank 0 over xGMI:
`torch.distributed.gather` -- RCCL's gather on the "nccl" backend, gloo on CPU for the tests -- or, opt-in, every rank
storing its share straight into rank 0's frame buffer (peer-mapped through a HIP IPC handle, one device-to-device copy
per rank and frame, a one-word all-reduce as the completion signal).  Work-item ids stay GLOBAL (renderer.Renderer /
clw_ext_set_id_offset / clw_ext_set_row_bands) so the assembled image is bit-identical to a single-GPU render.
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


class FrameGatherer:
    """One gather per frame of every rank's rows to rank 0, with two frame slots so the gather of frame k overlaps the
    trace of frame k+1.

    layout "bands": rank r owns the 8-row bands r, r + world, ... (renderer `bands=(world, rank)`; equal shares);
    layout "strips": rank r owns the contiguous rows `strip_rows(height, world, r)` (north_star's row strips).

    What travels is RGB888: the framebuffer word is 0x00RRGGBB, its top byte is always zero, so each share is packed
    to 3 bytes per pixel first (a strided copy on a side stream) -- 25 % less on the xGMI links, which are what
    bounds a gather into one GPU.  Rank 0 keeps, per slot, one row of `px_max * 3` bytes per rank (`parts`);
    `assemble` turns that into the 0x00RRGGBB frame.

    transport "gather" (alias "rccl-gather", the default): ONE `torch.distributed.gather` per frame -- on the nccl backend that is
    RCCL's gather, north_star's "single RCCL gather of the final image over xGMI".
    transport "peer" (opt-in; measured with two ranks on one GPU only, never yet on distinct GPUs): rank 0's `parts` are mapped into
    every rank through a HIP IPC handle and each rank writes its row with ONE device-to-device copy over its own xGMI link -- no
    send/recv kernels, no staging; a one-word all-reduce behind the copies is the completion signal.  It needs every process to see
    every GPU (the mapping opens rank 0's allocation on rank 0's device); where it cannot be set up the constructor RAISES for "peer"
    and falls back to "gather" only for "auto".

    Slot reuse.  Frame k+2 reuses the slot of frame k.  A rank must not overwrite rank 0's `parts[slot]` while rank 0 still reads
    frame k out of it: with "gather" rank 0 issues the receiving collective itself, behind the consumer's `release(slot)` event; with "peer"
    the writers wait for the completion all-reduce of frame k+1, and rank 0 joins that all-reduce only behind `release(slot)` of
    frame k -- the event a per-frame consumer records after its last read of `parts[slot]` (`assemble` records it itself).  A consumer
    that never reads intermediate frames (bench.py) calls nothing and nobody waits."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, layout: str = "bands",
                 transport: str = "auto", staged_on_cpu: bool = False):
        assert layout in ("bands", "strips")
        if layout == "bands":
            assert height % (8 * world) == 0, "interleaved bands need height % (8 * world) == 0"
        self.width, self.height, self.rank, self.world, self.layout = width, height, rank, world, layout
        self.rows = [height // world] * world if layout == "bands" else [strip_rows(height, world, r)[1] for r in range(world)]
        self.px = [r * width for r in self.rows]
        self.px_rank, self.px_max = self.px[rank], max(self.px)
        self.staged = staged_on_cpu           # rehearsal with gloo: collectives on CPU copies
        self.device = torch.device(device)
        self.on_gpu = self.device.type == "cuda" and not staged_on_cpu
        cdev = self.device if self.on_gpu else torch.device("cpu")
        self.packed = [torch.zeros(self.px_max * 3, dtype=torch.uint8, device=cdev) for _ in range(2)]
        self.pending = [None, None]
        self.comm = torch.cuda.Stream(device=self.device) if self.on_gpu else None
        self.ev_packed = [torch.cuda.Event() if self.on_gpu else None for _ in range(2)]
        self.used = [False, False]
        self.flag = torch.zeros(1, dtype=torch.int32, device=cdev)
        self.parts = [None, None]             # rank 0: uint8 [world, px_max * 3] per slot
        self.peer = [None, None]              # every rank (peer transport): rank 0's `parts`, mapped here
        self.consumed = [None, None]          # rank 0, peer transport: recorded after the consumer's last read of parts[slot]
        self.transport = "gather"
        if transport == "rccl-gather":
            transport = "gather"
        if world > 1:
            if transport in ("auto", "peer") and self.device.type == "cuda":
                self.transport = "peer" if self._map_peer_buffers() else "gather"
                if transport == "peer" and self.transport != "peer":
                    raise RuntimeError("peer-mapped gather buffers could not be set up")
            if self.transport == "gather" and rank == 0:
                self.parts = [torch.zeros(world, self.px_max * 3, dtype=torch.uint8, device=cdev) for _ in range(2)]

    # ---- peer transport: rank 0 allocates, everybody maps -----------------------------------------------------
    def _map_peer_buffers(self) -> bool:
        """Rank 0 shares its two `parts` buffers through HIP IPC handles (what torch.multiprocessing uses for CUDA
        tensors); all ranks agree on success through an all-reduce, so either everybody uses the mapping or nobody."""
        from torch.multiprocessing.reductions import reduce_tensor
        ok, handles = 1, [None, None]
        try:
            if self.rank == 0:
                self.parts = [torch.zeros(self.world, self.px_max * 3, dtype=torch.uint8, device=self.device) for _ in range(2)]
                handles = [reduce_tensor(t) for t in self.parts]
        except Exception:
            ok = 0
        box = [handles]
        dist.broadcast_object_list(box, src=0)
        try:
            if self.rank == 0:
                self.peer = list(self.parts)
            elif box[0][0] is not None:
                self.peer = [fn(*args) for (fn, args) in box[0]]
                self.peer[0][self.rank, :1].copy_(self.packed[0][:1])        # touch it: enables peer access now, not in the timed loop
                torch.cuda.synchronize(self.device)
            else:
                ok = 0
        except Exception:
            ok = 0
        cdev = self.device if dist.get_backend() == "nccl" else "cpu"

        def agreed(flag: int) -> bool:
            t = torch.tensor([flag], dtype=torch.int32, device=cdev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1

        if agreed(ok):
            # prove the mapping before trusting it: every rank writes a pattern into its row of both slots through ITS
            # mapping, rank 0 reads the rows back through its own pointer
            try:
                n = min(4096, self.px_max * 3)
                pat = ((torch.arange(n, device=self.device, dtype=torch.int32) * 7 + self.rank * 31 + 5) % 251).to(torch.uint8)
                for slot in (0, 1):
                    self.peer[slot][self.rank, :n].copy_(pat)
                torch.cuda.synchronize(self.device)
            except Exception:
                ok = 0
            dist.barrier()
            if self.rank == 0 and ok:
                n = min(4096, self.px_max * 3)
                for r in range(self.world):
                    want = ((torch.arange(n, device=self.device, dtype=torch.int32) * 7 + r * 31 + 5) % 251).to(torch.uint8)
                    for slot in (0, 1):
                        if not torch.equal(self.parts[slot][r, :n], want):
                            ok = 0
                for slot in (0, 1):
                    self.parts[slot].zero_()
                torch.cuda.synchronize(self.device)
            if agreed(ok):
                return True
        self.peer = [None, None]
        if self.rank == 0:
            self.parts = [None, None]
        return False

    def before_render(self, slot: int) -> None:
        """Call before tracing into the slot's framebuffer again: its previous contents must have been packed."""
        if self.on_gpu and self.used[slot]:
            torch.cuda.current_stream(self.device).wait_event(self.ev_packed[slot])

    def _wait(self, slot: int) -> None:
        if self.pending[slot] is not None:
            self.pending[slot].wait()
            self.pending[slot] = None

    def submit(self, slot: int, share: torch.Tensor) -> None:
        """share: int32 [px_rank] framebuffer of this rank (0x00RRGGBB), just rendered on the current stream."""
        if self.world == 1:
            return
        n3 = self.px_rank * 3
        bgr = share.view(torch.uint8).view(-1, 4)[:, :3]         # little-endian: B, G, R, 0
        if self.on_gpu:
            rendered = torch.cuda.Event()
            rendered.record(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(self.comm):
                self.comm.wait_event(rendered)
                self._wait(slot)                                  # the transfer that still reads packed[slot]
                self.packed[slot][:n3].view(-1, 3).copy_(bgr)
                self.ev_packed[slot].record(self.comm)
                if self.transport == "peer":
                    self._wait(slot ^ 1)                          # frame k-1 complete everywhere => rank 0 has released frame k-2's slot (this one)
                    if self.rank == 0 and self.consumed[slot ^ 1] is not None:
                        self.comm.wait_event(self.consumed[slot ^ 1])   # rank 0 joins frame k's signal only behind its reads of frame k-1... see release()
                    self.peer[slot][self.rank, :n3].copy_(self.packed[slot][:n3], non_blocking=True)   # one copy over this rank's link
                    self.pending[slot] = dist.all_reduce(self.flag, async_op=True)                     # "frame complete" for rank 0
                else:
                    if self.rank == 0 and self.consumed[slot] is not None:
                        self.comm.wait_event(self.consumed[slot])     # the consumer's reads of frame k-2 (same slot) come first
                    self.pending[slot] = dist.gather(self.packed[slot], gather_list=list(self.parts[slot]) if self.rank == 0 else None,
                                                     dst=0, async_op=True)
            self.used[slot] = True
        else:
            self._wait(slot)
            self.packed[slot][:n3].view(-1, 3).copy_(bgr.cpu() if share.is_cuda else bgr)
            if self.transport == "peer":                          # rehearsal on one GPU: the mapping is real, the signal is gloo's
                self._wait(slot ^ 1)                              # see the GPU branch: the other slot's frame is complete everywhere
                if self.rank == 0 and self.consumed[slot ^ 1] is not None:
                    self.consumed[slot ^ 1].synchronize()
                self.peer[slot][self.rank, :n3].copy_(self.packed[slot][:n3].to(self.device))
                torch.cuda.synchronize(self.device)
                self.pending[slot] = dist.all_reduce(self.flag, async_op=True)
            else:
                self.pending[slot] = dist.gather(self.packed[slot], gather_list=list(self.parts[slot]) if self.rank == 0 else None,
                                                 dst=0, async_op=True)

    def drain(self) -> None:
        for slot in (0, 1):
            if self.pending[slot] is not None:
                if self.on_gpu:
                    with torch.cuda.stream(self.comm):
                        self._wait(slot)
                else:
                    self._wait(slot)
        if self.on_gpu:
            self.comm.synchronize()

    def complete(self, slot: int) -> None:
        """Rank 0, per-frame consumer: block until the frame submitted into `slot` has arrived in `parts[slot]`."""
        if self.on_gpu:
            with torch.cuda.stream(self.comm):
                self._wait(slot)
            self.comm.synchronize()
        else:
            self._wait(slot)

    def release(self, slot: int) -> None:
        """Rank 0: the consumer's reads of `parts[slot]` (queued on the current stream) are the last ones: the ranks may overwrite
        it from the frame after next on.  With the peer transport rank 0 signals "frame k+1 complete" only behind this point, and
        the writers of frame k+2 wait for that signal (submit)."""
        if self.rank == 0 and self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self.consumed[slot] = ev

    def assemble_rgb(self, slot: int) -> torch.Tensor:
        """Rank 0: the last frame gathered into `slot` as uint8 [height * width, 3] in B, G, R byte order."""
        assert self.rank == 0
        parts = self.parts[slot]
        if self.layout == "bands":
            nb = self.height // (8 * self.world)                  # bands per rank
            return torch.stack([parts[r].view(nb, 8 * self.width * 3) for r in range(self.world)], 1).reshape(-1, 3)
        return torch.cat([parts[r][: self.px[r] * 3] for r in range(self.world)]).view(-1, 3)

    def assemble(self, slot: int) -> torch.Tensor:
        """Rank 0: the full frame as int32 [height * width] (0x00RRGGBB) of the last frame gathered into `slot`."""
        rgb = self.assemble_rgb(slot).to(torch.int32)
        out = (rgb[:, 2] << 16) | (rgb[:, 1] << 8) | rgb[:, 0]
        self.release(slot)
        return out


class BandGatherer(FrameGatherer):
    """FrameGatherer with interleaved 8-row bands (bench.py's weak-scaling mode)."""

    def __init__(self, width: int, height: int, rank: int, world: int, device, staged_on_cpu=False, transport="auto"):
        super().__init__(width, height, rank, world, device, layout="bands", transport=transport, staged_on_cpu=staged_on_cpu)
