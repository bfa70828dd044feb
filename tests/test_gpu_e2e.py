"""End-to-end on the MI355X: `main.twotower.main(rank, manager)` — Manager flags -> loaders over
MIND-shaped synthetic caches -> train steps (HIP fwd/bwd/Adam) -> cached evaluation -> metrics ->
checkpoint, for the reference's default pair (cnn + rnn, BASELINE config 1 family) and the NRMS pair."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(tmp_path, monkeypatch, flags, n_train=40):
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    from main import twotower
    monkeypatch.chdir(tmp_path)
    m = Manager({"enable_fields": ["title"], "newsEncoder": "cnn", "userEncoder": "rnn"},
                command=(flags + f" -s demo --synthetic-plm -cr {tmp_path}/cache -dr {tmp_path}/data").split())
    m.news_nums = {"MINDdemo_train": 700, "MINDdemo_dev": 500}
    write_synthetic_caches(m, n_train_impr=n_train, n_dev_impr=30, seed=1)
    return m, twotower.main(m.device, m)


def test_main_twotower_default_pair_trains_and_evaluates(tmp_path, monkeypatch):
    m, best = _run(tmp_path, monkeypatch, "-bs 8 -e 1 -hd 64 -vs 4 -lr 1e-3")
    assert set(best) >= {"auc", "mean_mrr", "ndcg@5", "ndcg@10", "main", "step"}
    assert 0.0 <= best["auc"] <= 1.0
    assert os.path.exists("data/ckpts/TwoTower-Cnn-Rnn/demo/best.model")
    assert "TwoTower-Cnn-Rnn" in open("performance.log").read()


def test_main_twotower_nrms_pair_loss_decreases(tmp_path, monkeypatch):
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    from models.TwoTower import TwoTowerModel
    from models.modules import encoder as E
    monkeypatch.chdir(tmp_path)
    m = Manager({"enable_fields": ["title"]}, command=f"-bs 16 -e 1 -ne tfm -ue tfm -hd 48 -hn 4 -lr 3e-3 -s demo --synthetic-plm "
                                                      f"-cr {tmp_path}/cache -dr {tmp_path}/data".split())
    m.news_nums = {"MINDdemo_train": 700, "MINDdemo_dev": 500}
    write_synthetic_caches(m, n_train_impr=64, n_dev_impr=30, seed=2)
    m.setup(m.device)
    loaders = m.prepare()
    model = TwoTowerModel(m, E.TfmNewsEncoder(m), E.TfmUserEncoder(m)).to(m.device).train()
    opt, _ = model.get_optimizer(m, len(loaders["train"]))
    x = next(iter(loaders["train"]))
    losses = []
    for _ in range(30):                       # overfit one batch: the optimisation loop must drive the loss down
        opt.zero_grad(set_to_none=True)
        loss = model(x)
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert losses[-1] < 0.5 * losses[0], losses[::5]
    metrics = model.dev(m, loaders)
    assert metrics["main"] == metrics["auc"]
    # dev mode through the entry point, loading the checkpoint just written
    m.save(model, 1, best=True)
    m2 = Manager({"enable_fields": ["title"]}, command=f"-m dev -ne tfm -ue tfm -hd 48 -hn 4 -s demo --synthetic-plm "
                                                       f"-cr {tmp_path}/cache -dr {tmp_path}/data".split())
    m2.news_nums = m.news_nums
    from main import twotower
    again = twotower.main(m2.device, m2)
    assert again == metrics
