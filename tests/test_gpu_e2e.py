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
    ds = loaders["train"].dataset
    assert not ds.ship_tokens                  # default on a GPU: batches carry news indices only
    model.attach_news_tokens(ds.token_ids, ds.attn_masks)
    x = next(iter(loaders["train"]))
    assert "cdd_token_id" not in x
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
    m.host_metrics = True                      # the reference's host regrouping + python metric loops
    assert model.dev(m, loaders) == metrics    # ... give exactly what the device kernel gave
    m.host_metrics = False
    # dev mode through the entry point, loading the checkpoint just written
    m.save(model, 1, best=True)
    m2 = Manager({"enable_fields": ["title"]}, command=f"-m dev -ne tfm -ue tfm -hd 48 -hn 4 -s demo --synthetic-plm "
                                                       f"-cr {tmp_path}/cache -dr {tmp_path}/data".split())
    m2.news_nums = m.news_nums
    from main import twotower
    again = twotower.main(m2.device, m2)
    assert again == metrics


def test_main_twotower_title_plus_abstract(tmp_path, monkeypatch):
    """`-ef title abs` of the reference (scripts: sequence_length = 32 + 64 = 96 tokens): the attention and pooling
    kernels run two waves per title; training, cached evaluation and metrics go through end to end."""
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    from main import twotower
    monkeypatch.chdir(tmp_path)
    m = Manager({"enable_fields": ["title", "abs"]}, command=f"-bs 8 -e 1 -ne tfm -ue attn -hd 48 -hn 4 -vs 4 -lr 1e-3 -s demo "
                                                             f"--synthetic-plm -cr {tmp_path}/cache -dr {tmp_path}/data".split())
    assert m.sequence_length == 96
    m.news_nums = {"MINDdemo_train": 300, "MINDdemo_dev": 200}
    write_synthetic_caches(m, n_train_impr=24, n_dev_impr=20, seed=3)
    best = twotower.main(m.device, m)
    assert 0.0 <= best["auc"] <= 1.0 and best["step"] > 0


def test_device_side_batch_assembly_is_bit_exact(tmp_path, monkeypatch):
    """token rows gathered on the device from the resident [Nn,T] tables equal the rows the
    reference's __getitem__ ships (--host-assembly), and so does the loss."""
    import random
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    from utils.dataset import MIND_Train
    from models.TwoTower import TwoTowerModel
    from models.modules import encoder as E
    from newsrec_hip import ops
    monkeypatch.chdir(tmp_path)
    base = f"-bs 8 -ne cnn -ue attn -hd 32 -s demo --synthetic-plm -cr {tmp_path}/cache -dr {tmp_path}/data"
    m_dev = Manager({"enable_fields": ["title"]}, command=base.split())
    m_host = Manager({"enable_fields": ["title"]}, command=(base + " --host-assembly").split())
    for m in (m_dev, m_host):
        m.news_nums = {"MINDdemo_train": 700, "MINDdemo_dev": 500}
    write_synthetic_caches(m_dev, n_train_impr=24, n_dev_impr=10, seed=4)
    d_dev, d_host = MIND_Train(m_dev), MIND_Train(m_host)
    assert d_host.ship_tokens and not d_dev.ship_tokens
    collate = torch.utils.data.default_collate
    random.seed(9)
    xb_host = collate([d_host[i] for i in range(8)])
    random.seed(9)
    xb_dev = collate([d_dev[i] for i in range(8)])
    assert torch.equal(xb_host["cdd_idx"], xb_dev["cdd_idx"]) and "his_token_id" not in xb_dev
    tab = torch.from_numpy(d_dev.token_ids).cuda()
    assert torch.equal(ops.gather_rows_i64(tab, xb_dev["his_idx"].cuda()).cpu(), xb_host["his_token_id"])
    m_dev.setup(m_dev.device)
    torch.manual_seed(1)
    model = TwoTowerModel(m_dev, E.CnnNewsEncoder(m_dev), E.AttnUserEncoder(m_dev)).to(m_dev.device).eval()
    model.attach_news_tokens(d_dev.token_ids, d_dev.attn_masks)
    assert float(model(xb_dev)) == float(model(xb_host))


def test_main_twotower_test_mode_writes_prediction_file(tmp_path, monkeypatch):
    """`-m test` through main.twotower.main (reference BaseModel.py:126-152,292-329): the corpus is encoded, the test
    impressions are scored from the cache and prediction.txt holds, per impression, the 1-based ranks of its
    candidates — equal to scipy's rankdata(1 - score, "min") (the reference's expression) on the scores `_test` returns."""
    import scipy.stats as ss
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    from main import twotower
    from models.TwoTower import TwoTowerModel
    from models.modules import encoder as E
    monkeypatch.chdir(tmp_path)
    base = f"-ne tfm -ue tfm -hd 48 -hn 4 -is 6 -s demo --synthetic-plm -cr {tmp_path}/cache -dr {tmp_path}/data"
    m = Manager({"enable_fields": ["title"]}, command=("-m test " + base).split())
    m.news_nums = {"MINDdemo_train": 300, "MINDdemo_dev": 200, "MINDdemo_test": 250}
    write_synthetic_caches(m, n_train_impr=8, n_dev_impr=8, seed=6, splits=("test",))
    # a checkpoint to load (-m test defaults to -ck best)
    m.setup(m.device)
    torch.manual_seed(5)
    model = TwoTowerModel(m, E.TfmNewsEncoder(m), E.TfmUserEncoder(m)).to(m.device)
    m.save(model, 1, best=True)
    twotower.main(m.device, m)
    path = os.path.join("data/cache/results", model.name, "demo", "best", "prediction.txt")
    lines = open(path).read().splitlines()
    loaders = m.prepare()
    with torch.no_grad():
        preds = model.eval()._test(m, loaders)
    assert len(lines) == len(preds) == 8 and any(len(p) > 6 for p in preds)        # impressions span several 6-candidate rows
    for i, (line, p) in enumerate(zip(lines, preds), 1):
        want = ss.rankdata(1 - np.asarray(p), method="min")
        assert line == "{} [{}]".format(i, ",".join(str(int(r)) for r in want))
