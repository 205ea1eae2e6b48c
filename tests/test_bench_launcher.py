"""bench.py as a launcher: `--gpus N` must either run N ranks or fail loudly -- never print a
1-GPU number labelled N.  Rehearsed on CPU: the stub render path uses the same process-group setup, FrameGather,
barriers and JSON plumbing as the GPU path, over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, env=e, timeout=timeout)


def test_plain_start_launches_its_own_ranks():
    r = _run(["--gpus", "2", "--stub-render", "--size", "72", "--row-block", "8", "--steps", "2", "--warmup", "1"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                       # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_world"] == 2 and d["frame_ok"] is True


def test_three_ranks_ragged_partition():
    r = _run(["--gpus", "3", "--stub-render", "--size", "50", "--row-block", "16", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert d["n_gpus"] == 3 and d["frame_ok"] is True


def test_world_size_mismatch_is_an_error():
    r = _run(["--gpus", "4", "--stub-render", "--size", "32"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


def test_more_gpus_than_the_node_has_fails_loudly():
    import torch
    have = torch.cuda.device_count()
    r = _run(["--gpus", str(max(have, 1) + 1), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"])
    assert r.returncode != 0 and f"has {have} GPU" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]       # no JSON line: nothing to misread


def test_a_failing_rank_fails_the_launch():
    # rank 1 cannot initialise (bad backend name): the launcher must return non-zero, not hang on rank 0's barrier
    r = _run(["--gpus", "2", "--stub-render", "--size", "32", "--backend", "no_such_backend"], timeout=120)
    assert r.returncode != 0
