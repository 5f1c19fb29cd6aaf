"""bench.py's N > 1 path on the ONE-GPU box: two ranks share cuda:0 and gather through gloo (RCCL needs distinct
GPUs), everything else -- interleaved bands with global ids, the side-stream RGB888 packing, the double-buffered
gather, the assembly on rank 0 -- is the code the 8-GPU run uses.  The assembled frame must equal a single-GPU render."""
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


@pytest.mark.parametrize("device_tensors", [False, True])
def test_two_rank_bench_assembles_the_single_gpu_frame(demo_scene, tex, tmp_path, device_tensors):
    import torch  # noqa: F401
    from example_gui_opencl_raytracer_amd import api, textures
    from example_gui_opencl_raytracer_amd.renderer import Renderer
    png = str(tmp_path / "bands.png")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--rehearse", "--dump-png", png] + (["--rehearse-device-tensors"] if device_tensors else [])
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frame"] == "1920x2160" and line["value"] > 0
    img = api.read_png(png)
    got = (img[..., 0].astype(np.uint32) << 16 | img[..., 1].astype(np.uint32) << 8 | img[..., 2]).reshape(-1)
    r = Renderer(demo_scene, tex, textures.skybox_cross(4096), 1920, 2160, depth=4)
    r.look(**CAM)
    want = r.render()
    r.release()
    assert np.array_equal(got, want)
