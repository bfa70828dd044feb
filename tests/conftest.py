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
    config._nr_rccl1 = None
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
    # the N > 1 code path of bench.py on RCCL itself, with the one rank a one-GPU box can give it (NR_BENCH_FORCE_DIST=1:
    # "nccl" process group with device_id, FlatGradSync's two all-reduces, barriers, the all-gathered news cache, the typed
    # gather of _predict); tests/test_gpu_rccl_world1.py checks the line it prints
    env = dict(os.environ, NR_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        r = subprocess.run([sys.executable, "-u", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "2",
                            "--repeats", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
        config._nr_rccl1 = {"rc": r.returncode, "out": r.stdout[-20000:], "err": r.stderr[-4000:]}
    except subprocess.TimeoutExpired as e:
        config._nr_rccl1 = {"rc": 124, "out": "", "err": "timed out: " + str(e)[-2000:]}


def pytest_collection_modifyitems(config, items):
    if _gpu_visible():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
