"""Multi-GPU parity on real devices: RCCL gather of block-cyclic row partitions == the single-GPU frame, byte for byte.
Needs at least two GPUs on the node; on a one-GPU box it is skipped (the same path runs over gloo with two CPU processes
in tests/test_sharding_gloo.py and tests/test_bench_launcher.py, and lt_render_multi's partitioning is exercised on one
device in tests/test_gpu_parity.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gpus():
    import torch
    return torch.cuda.device_count()          # does not initialise a GPU context in this process


@pytest.mark.parametrize("world", [2, 4, 8])
def test_rccl_gather_equals_single_gpu_frame(world):
    if _gpus() < world:
        pytest.skip(f"needs {world} GPUs on the node, {_gpus()} visible")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "nccl_gather_check.py")], env=env))
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world, codes


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_rank_path_rehearsed_on_one_gpu(world):
    """The N-rank path with the ranks sharing GPU 0 and the gather staged through the host (gloo): every rank renders
    its partition with the HIP library, sharding.FrameGather reassembles, rank 0 compares with its own whole frame."""
    if _gpus() < 1:
        pytest.skip("needs a GPU")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LT_CHECK_DEVICES=",".join(["0"] * world), LT_CHECK_BACKEND="gloo")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "nccl_gather_check.py")], env=env))
    codes = [p.wait(timeout=600) for p in procs]
    assert codes == [0] * world, codes


def test_render_multi_on_real_devices():
    """lt_render_multi with one partition per real device (single process, no RCCL) == the one-device frame."""
    n = _gpus()
    if n < 2:
        pytest.skip(f"needs 2 GPUs on the node, {n} visible")
    import numpy as np
    import ltrace
    W, H = 640, 500
    fov_v = np.radians(40.0)
    cam = ltrace.Camera(W, H, 2 * np.arctan(np.tan(fov_v / 2) * W / H), fov_v, 0.0, 0.0, 50.0, np.pi / 2)
    met = ltrace.Metric(1, 0, 1.0, 0.9)
    one = ltrace.render(cam, met, ltrace.default_opts(precision=32), want=("fa", "status", "rgba"))
    many = ltrace.render_multi(cam, met, ltrace.default_opts(precision=32), min(n, 8), want=("fa", "status", "rgba"))
    for k in ("fa", "status", "rgba"):
        assert np.array_equal(one[k], many[k], equal_nan=True), k


def test_row_scatter_kernels_against_numpy():
    """lt_scatter_rows_dev / lt_scatter_rows_indexed_dev byte for byte against numpy indexing (tests/scatter_rows_check.py,
    in a process of its own: it holds device buffers through torch, and torch's HIP runtime must be the first one a
    process initialises)."""
    if _gpus() < 1:
        pytest.skip("needs a GPU")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "scatter_rows_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
