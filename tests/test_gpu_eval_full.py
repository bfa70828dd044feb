"""BASELINE configs[3], single-GPU leg, at its real size ("missing" item 2): the MINDlarge_dev corpus of
72 024 news through `TwoTowerBaseModel.encode_news`, >= 100 000 impressions through `_predict`, per-impression
metrics on the device.  Mirrors reference models/BaseModel.py:221-289 and utils/util.py:37-57,197-250.

  * the packed-title cache equals the padded-layout cache on ALL 72 024 rows, and the oracle on 512 sampled rows (1e-5);
  * `_predict` logits equal the oracle's `infer` on 384 sampled rows (1e-5), using the same cache;
  * `device_compute_metrics` == `compute_metrics(pack_results(...))` (the reference's host formulas) after round(., 4)
    over every impression, exact score ties included.
"""
import types

import numpy as np
import pytest
import torch

import bench
from helpers import relerr
from oracle import twotower_oracle as O

pytestmark = pytest.mark.gpu
FP32_TOL = 1e-5


def test_full_corpus_cache_and_100k_impressions():
    from utils.util import device_compute_metrics, compute_metrics, pack_results
    torch.cuda.set_device(0)
    model, mgr = bench.build_model(0, "nrms")
    P = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().eval()
    Nn, L, Ce, H = bench.NEWS_DEV, bench.L, 20, bench.H
    g = torch.Generator().manual_seed(7)
    ids, msk = bench.synth_table(g, Nn)
    ids[5], msk[5] = 0, 0                      # an all-PAD title inside the corpus (rows without a title are encoded too)
    ds = type("NewsRows", (), {"__len__": lambda self: Nn, "token_ids": ids.numpy(), "attn_masks": msk.numpy()})()
    loader_news = types.SimpleNamespace(dataset=ds, batch_size=200)
    mgr.distributed, mgr.debug, mgr.impr_size = False, False, Ce

    # ---- the cache: packed == padded on every row, oracle on a sample
    with torch.no_grad():
        model.encode_news(mgr, loader_news)                         # host arrays, chunked copies
        cache_packed = model.news_embeddings.clone()
        model.attach_news_corpus(ids, msk)
        model.encode_news(mgr, loader_news)                         # token table resident in HBM
        assert torch.equal(model.news_embeddings, cache_packed)
        model.newsEncoder.no_packing = True
        model.encode_news(mgr, loader_news)
        cache_padded = model.news_embeddings.clone()
        model.newsEncoder.no_packing = False
    assert cache_packed.shape == (Nn, bench.D) and torch.isfinite(cache_packed).all()
    row_err = (cache_packed - cache_padded).abs().amax(1) / cache_padded.abs().amax(1)
    print(f"packed vs padded cache: bit-equal rows {int((row_err == 0).sum())}/{Nn}, worst row relerr {row_err.max().item():.2e}")
    assert row_err.max().item() < 2e-6                                # per ROW, not max-norm over the table
    sample = torch.cat([torch.tensor([0, 5, Nn - 1]), torch.randint(1, Nn, (509,), generator=g)])
    ref = O.build_news_cache(P, "tfm", ids[sample], msk[sample], H)
    got = cache_packed[sample.cuda()].cpu()
    per_row = (got - ref).abs().amax(1) / ref.abs().amax(1)
    assert per_row.max().item() < FP32_TOL, per_row.max().item()

    # ---- >= 1e5 impressions (2 .. 80 candidates, <= 20 per row) through _predict
    n_impr = 100_000
    ncand = torch.randint(2, 81, (n_impr,), generator=g)
    nrow = (ncand + Ce - 1) // Ce
    impr_of_row = torch.repeat_interleave(torch.arange(n_impr), nrow)
    first = torch.cumsum(nrow, 0) - nrow
    within = torch.arange(int(nrow.sum())) - first[impr_of_row]
    valid = (within[:, None] * Ce + torch.arange(Ce)[None, :]) < ncand[impr_of_row][:, None]
    R = impr_of_row.numel()
    cdd = torch.randint(1, Nn, (R, Ce), generator=g) * valid
    dup = torch.randint(0, n_impr, (2000,), generator=g)            # duplicate news inside an impression -> exact score ties
    cdd[first[dup], 1] = cdd[first[dup], 0]
    hl = torch.randint(0, 71, (n_impr,), generator=g).clamp(max=L)
    his_u = torch.randint(1, Nn, (n_impr, L), generator=g) * (torch.arange(L)[None, :] < hl[:, None])
    hm_u = (torch.arange(L)[None, :] < hl[:, None]).long()
    hm_u[hl == 0, 0] = 1                                            # empty history: his_mask[0] = 1 (dataset.py:140-141)
    lab = torch.where(valid, (torch.rand(R, Ce, generator=g) < 0.1).long(), torch.tensor(-1))
    lab[first, 0], lab[first, 1] = 1, 0                             # both classes in every impression
    his, hm = his_u[impr_of_row], hm_u[impr_of_row]
    dev = lambda t: t.cuda()
    bs = 8192
    loader = [{"cdd_idx": dev(cdd[s:s + bs]), "his_idx": dev(his[s:s + bs]), "his_mask": dev(hm[s:s + bs]),
               "impr_index": dev(impr_of_row[s:s + bs]), "cdd_mask": dev(valid[s:s + bs]), "label": dev(lab[s:s + bs])}
              for s in range(0, R, bs)]
    model.news_embeddings = cache_packed
    with torch.no_grad():
        impr_d, masks_d, preds_d, labels_d = model._predict(mgr, loader, with_labels=True, to_host=False)
    assert preds_d.shape == (R, Ce) and torch.equal(impr_d.cpu(), impr_of_row) and torch.equal(masks_d.cpu(), valid)

    rows = torch.randint(0, R, (384,), generator=g)
    ref_lg = O.infer(P, "tfm", cache_packed.cpu(), cdd[rows], his[rows], hm[rows], H)
    got_lg = preds_d[rows.cuda()].cpu()
    v = valid[rows]
    assert ((got_lg - ref_lg).abs()[v].max() / ref_lg.abs()[v].max()).item() < FP32_TOL

    metrics = ["auc", "mean_mrr", "ndcg@5", "ndcg@10"]
    got = device_compute_metrics(impr_d, masks_d, preds_d, labels_d, metrics)
    assert got is not None
    l_, p_ = pack_results(impr_d.cpu().numpy(), masks_d.cpu().numpy().astype(np.bool_), labels_d.cpu().numpy().astype(np.int32),
                          preds_d.cpu().numpy().astype(np.float32))
    assert len(l_) == n_impr and [len(x) for x in l_[:50]] == ncand[:50].tolist()
    want = compute_metrics(l_, p_, metrics)
    print("device metrics", got, "host metrics", want)
    assert got == want
    # and against the oracle's restatement of the reference formulas on a slice (pinned by tests/golden/metrics.npz)
    o = O.compute_metrics(l_[:5000], p_[:5000], tuple(metrics))
    h = compute_metrics(l_[:5000], p_[:5000], metrics)
    assert all(abs(o[k] - h[k]) <= 1e-4 for k in metrics), (o, h)
