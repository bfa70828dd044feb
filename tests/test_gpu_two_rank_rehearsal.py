"""`python -m main.twotower -ws 2` rehearsed on one GPU (reference utils/manager.py:196-208,
main/twotower.py:38-39,55-61): two fresh ranks run `main.twotower.main` through Manager.setup (process group,
HSA_ENABLE_IPC_MODE_LEGACY exported for the children), DistributedSampler, FlatGradSync (dense-parameter all-reduce
started at the token table's scatter-add, table all-reduce after it), at least 5 train steps and the sharded evaluation, over
gloo with both ranks on device 0.  The run itself happens in tests/conftest.py::pytest_sessionstart — before this
pytest process touches the GPU — and this test checks its records."""
import json
import os

import pytest

pytestmark = pytest.mark.gpu


def test_main_twotower_two_ranks_stay_bit_identical(request):
    rec = getattr(request.config, "_nr_rehearsal", None)
    assert rec is not None, "the rehearsal did not run at session start (NR_SKIP_REHEARSAL=1 or `not gpu` in -m?)"
    assert rec["rc"] == 0, rec["log"]
    ranks = [json.load(open(os.path.join(rec["dir"], f"rank{r}.json"))) for r in range(2)]
    for r, res in enumerate(ranks):
        assert res["rank"] == r and res["world"] == 2 and res["backend"] == "gloo"
        assert res["wrapper"] == "FlatGradSync" and res["steps"] >= 5            # >= 80 train items / 2 ranks / batch 8
        assert res["replicas_identical"], "replicas diverged after 5 steps"
        assert res["ipc_env"] == "0"                                             # what RCCL needs, set by Manager for its children
        assert res["so_loaded"]                                                  # the HIP library did the work
        assert 0 < res["n_early"] < res["n_flat"]                                # two collectives: dense | token table
    assert ranks[0]["param_abs_sum"] == ranks[1]["param_abs_sum"]
    assert set(ranks[0]["best"]) >= {"auc", "mean_mrr", "ndcg@5", "ndcg@10"}     # rank 0 evaluated the gathered shards
