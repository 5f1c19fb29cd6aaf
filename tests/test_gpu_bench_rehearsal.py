"""bench.py's N > 1 paths on the ONE-GPU box: two ranks share cuda:0 and talk through gloo (RCCL needs distinct GPUs);
everything else -- global ids, interleaved bands / row strips, the side-stream RGB888 packing, the double-buffered
transfer into rank 0's frame buffer (peer-mapped through a HIP IPC handle, or torch.distributed.gather), the assembly
on rank 0 and, for config C5, the single PNG -- is the code the 8-GPU run uses.  The assembled frame must equal a
single-GPU render."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import CAM, ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(extra, png):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse", "--dump-png", png] + extra
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    return json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])


def _single_gpu_frame(demo_scene, tex, w, h):
    import torch  # noqa: F401
    from example_gui_opencl_raytracer_amd import textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    r = Renderer(demo_scene, tex, textures.skybox_cross(4096), w, h, depth=4)
    r.look(**CAM)
    want = r.render()
    r.release()
    return want


def _png_frame(path):
    from example_gui_opencl_raytracer_amd import api
    img = api.read_png(path)
    return (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)


@pytest.mark.parametrize("extra", [["--transport", "rccl-gather"], ["--transport", "rccl-gather", "--rehearse-device-tensors"],
                                   ["--transport", "peer"], ["--transport", "peer", "--rehearse-device-tensors"]],
                         ids=["gather-staged", "gather-device", "peer-staged", "peer-device"])
def test_two_rank_weak_scaling_assembles_the_single_gpu_frame(demo_scene, tex, tmp_path, extra):
    png = str(tmp_path / "bands.png")
    line = _run_bench(["--steps", "6", "--warmup", "2", "--scaling", "weak", "--legs", "none"] + extra, png)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frame"] == "1920x2160" and line["value"] > 0
    assert extra[1] in line["config"]["sharding"] and line["value_per_gpu"] == pytest.approx(line["value"] / 2, rel=1e-3)
    assert line["legs"]["weak"]["transport"] == extra[1]
    assert np.array_equal(_png_frame(png), _single_gpu_frame(demo_scene, tex, 1920, 2160))


def test_two_rank_default_is_the_fixed_frame_in_row_strips_with_all_three_legs(demo_scene, tex, tmp_path):
    """`bench.py --gpus N` with no further flags measures BASELINE.json's metric: the FIXED 1920x1080 frame in contiguous row
    strips (536 + 544 rows here), ONE torch.distributed.gather per frame, and carries the weak-scaling and C5 results as named legs."""
    png = str(tmp_path / "strips.png")
    line = _run_bench(["--steps", "4", "--warmup", "2"], png)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["frame"] == "1920x1080"
    assert "2 contiguous row strips" in line["config"]["sharding"] and "rccl-gather" in line["config"]["sharding"]
    assert 13.8 < line["config"]["rays_per_pixel"] < 13.95
    assert line["repeats"] >= 1 and line["ms_per_step_min"] <= line["ms_per_step"] <= line["ms_per_step_max"]
    assert set(line["legs"]) == {"strong", "weak", "c5"}
    for name, frame, scaling in (("strong", "1920x1080", "strong"), ("weak", "1920x2160", "weak"), ("c5", "8192x8192", "strong")):
        leg = line["legs"][name]
        assert leg["frame"] == frame and leg["scaling"] == scaling and leg["transport"] == "rccl-gather"
        assert leg["value"] > 0 and leg["value_per_gpu"] == pytest.approx(leg["value"] / 2, rel=1e-3) and leg["ms_per_step"] > 0
    assert line["legs"]["strong"]["value"] == line["value"]
    assert np.array_equal(_png_frame(png), _single_gpu_frame(demo_scene, tex, 1920, 1080))


def test_two_rank_strong_scaling_over_the_peer_mapped_buffer(demo_scene, tex, tmp_path):
    png = str(tmp_path / "strips_peer.png")
    line = _run_bench(["--steps", "6", "--warmup", "2", "--transport", "peer", "--legs", "none"], png)
    assert line["scaling"] == "strong" and line["legs"]["strong"]["transport"] == "peer"
    assert np.array_equal(_png_frame(png), _single_gpu_frame(demo_scene, tex, 1920, 1080))


def _peer_consumer(rank, world, port, out_path):
    """Two ranks on cuda:0, the peer-mapped transport, every frame DIFFERENT, rank 0 consuming every frame while the ranks keep
    submitting: a rank may only overwrite rank 0's slot behind the consumer's release (FrameGatherer, "Slot reuse")."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    from example_gui_opencl_raytracer_amd import distributed as D
    from example_gui_opencl_raytracer_amd.renderer import strip_rows
    torch.cuda.set_device(0)
    D.init_process_group("gloo")
    W, H = 1920, 1080
    dev = torch.device("cuda", 0)
    gat = D.FrameGatherer(W, H, rank, world, dev, layout="strips", transport="peer")
    assert gat.transport == "peer"
    r0, rows = strip_rows(H, world, rank)
    idx = torch.arange(W * H, dtype=torch.int64, device=dev)

    def frame_pixels(k):
        return ((idx * 2654435761 + k * 40503) & 0xFFFFFF).to(torch.int32)

    bad = 0
    for k in range(9):
        s = k & 1
        gat.before_render(s)
        gat.submit(s, frame_pixels(k)[r0 * W:(r0 + rows) * W].contiguous())
        if rank == 0:
            gat.complete(s)
            bad += int(not torch.equal(gat.assemble(s), frame_pixels(k)))
    gat.drain()
    dist.barrier()
    if rank == 0:
        np.save(out_path, np.array([bad]))
    dist.destroy_process_group()


def test_per_frame_consumer_over_the_peer_mapped_buffer(tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "peer_consumer.npy")
    mp.spawn(_peer_consumer, args=(2, _free_port(), out), nprocs=2, join=True)
    assert int(np.load(out)[0]) == 0


def test_two_rank_c5_writes_the_single_png(demo_scene, tex, tmp_path):
    """BASELINE config 5 with two strips: 8192x8192, depth 4, the single PNG written by rank 0."""
    png = str(tmp_path / "c5.png")
    line = _run_bench(["--steps", "2", "--warmup", "1", "--config", "c5", "--legs", "none"], png)
    assert line["scaling"] == "strong" and line["config"]["frame"] == "8192x8192" and line["png"]["write_s"] > 0
    assert line["png"]["bytes"] == os.path.getsize(png)
    got = _png_frame(png)
    want = _single_gpu_frame(demo_scene, tex, 8192, 8192)
    assert np.array_equal(got, want)
