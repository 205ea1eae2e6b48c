"""Ghost lanes of the integrate kernels (k_kerr_direct / k_kerr_queue) change no result: the same scenes with every
wavefront in ghost mode from its first iteration and with no wavefront ever in it give byte-identical outputs."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(long_iters, long_steps, **extra):
    env = dict(os.environ, LT_D_LONG=str(long_iters), LT_Q_LONG=str(long_steps), **extra)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ghost_lanes_check.py")], env=env, check=True,
                         capture_output=True, text=True, timeout=600).stdout
    lines = [ln for ln in out.splitlines() if ln.startswith("digest ")]
    assert len(lines) == 1, out
    return lines[0].split()[1]


@pytest.mark.gpu
def test_ghost_lanes_change_no_output():
    never = digest(1 << 30, 1 << 30)
    assert digest(1, 0) == never       # every wave, from its first iteration
    assert digest(37, 50) == never     # switch in mid-flight, after some lanes have finished and some have not


@pytest.mark.gpu
def test_handing_tiles_out_changes_no_output():
    """k_kerr_direct with tiles taken from a queue head by a grid that fills the chip once (default) against one
    workgroup per tile (LT_D_PERSIST=0), on frames with more tiles than the chip has wavefront slots."""
    assert digest(1024, 600, LT_CHECK_BIG="1", LT_D_PERSIST="1") == digest(1024, 600, LT_CHECK_BIG="1", LT_D_PERSIST="0")
