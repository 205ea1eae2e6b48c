"""pytest configuration: registers the `gpu` marker and puts the product package
directory (flat modules, like the reference's script directory) and the repo
root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "light-path-tracer_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MPLBACKEND", "Agg")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # built artefacts are git-ignored: a fresh checkout has no libltrace_hip.so / oracle yet.
    # hipcc cross-compiles without a GPU, so build them once here if they are missing.
    lib = os.path.join(PKG, "lib", "libltrace_hip.so")
    ora = os.path.join(ROOT, "oracle", "liblt_oracle.so")
    if not (os.path.exists(lib) and os.path.exists(ora)):
        import __graft_entry__
        __graft_entry__.build()


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip: only skip gpu
    # tests when they were not asked for explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
