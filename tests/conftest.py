import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "news-recommendation_amd")
for p in (ROOT, SRC, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def _gpu_visible():
    """counting devices does not initialise the GPU on this image (torch.cuda.is_available() does)"""
    import torch
    return torch.cuda.device_count() > 0


def pytest_sessionstart(session):
    """`-m gpu` runs on a GPU box: the two-rank rehearsal of `main.twotower` (tests/rehearse_two_ranks.py) needs its ranks
    started by a process that has not touched the GPU, so it runs HERE, to completion, before collection imports
    anything; tests/test_gpu_two_rank_rehearsal.py checks what it wrote.  NR_SKIP_REHEARSAL=1 skips it (quick local runs
    of single tests; the rehearsal test then fails rather than passing vacuously)."""
    config = session.config
    config._nr_rehearsal = None
    expr = config.getoption("markexpr", "") or ""
    if "not gpu" in expr or os.environ.get("NR_SKIP_REHEARSAL") == "1" or not _gpu_visible():
        return
    out = tempfile.mkdtemp(prefix="nr_rehearse_")
    try:
        r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "tests", "rehearse_two_ranks.py"), out],
                           capture_output=True, text=True, timeout=900)
        config._nr_rehearsal = {"dir": out, "rc": r.returncode, "log": (r.stdout + "\n" + r.stderr)[-6000:]}
    except subprocess.TimeoutExpired as e:
        config._nr_rehearsal = {"dir": out, "rc": 124, "log": "timed out: " + str(e)[-2000:]}


def pytest_collection_modifyitems(config, items):
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
