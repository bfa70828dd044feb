"""Pin the CPU oracle against vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import load_golden, relerr
from oracle import twotower_oracle as O

PAIRS = [(n, u) for n in ("cnn", "tfm") for u in ("rnn", "sum", "avg", "attn", "tfm")]
TOL = 2e-6      # same op sequence on the same CPU backend; GRU restated as explicit cell loop


def _run(tag, ne, ue):
    meta, P, X, Og, G, extra = load_golden(tag)
    P = {k: v.clone().requires_grad_(v.is_floating_point()) for k, v in P.items()}
    H = meta["H"]
    loss, aux = O.train_forward(P, ne, ue, X, head_num=H)
    assert relerr(aux["cdd"], Og["cdd_news"]) < TOL
    assert relerr(aux["his"], Og["his_news"]) < TOL
    assert relerr(aux["user"], Og["user"]) < TOL
    assert relerr(aux["logits"], Og["logits"]) < TOL
    assert abs(loss.item() - Og["loss"].item()) < TOL * max(1, abs(Og["loss"].item()))
    loss.backward()
    for n, g in G.items():
        got = P[n].grad
        if ("g_step." + n) in extra:
            step = int(extra["g_step." + n])
            # (key.bias grads are analytically 0 — softmax shift invariance — so allow an absolute floor)
            samp = got.reshape(-1)[::step]
            assert (samp - g).abs().max().item() <= 5e-5 * g.abs().max().item() + 1e-8, n
            ref_sum = float(extra["g_abs_sum." + n])
            assert abs(got.double().abs().sum().item() - ref_sum) <= 1e-4 * ref_sum + 1e-6, n
        else:
            scale = max(g.abs().max().item(), 1e-12)
            assert (got - g).abs().max().item() <= 2e-5 * scale + 1e-9, n
    with torch.no_grad():
        cache = O.build_news_cache(P, ne, X["news_token_ids"], X["news_attn_masks"], H, batch=7)
        assert relerr(cache, Og["news_cache"]) < TOL
        lg = O.infer(P, ue, Og["news_cache"], X["cdd_idx"], X["his_idx"], X["his_mask"], H)
        assert relerr(lg, Og["infer_logits"]) < TOL
    if "cdd_token_emb" in Og and ne == "tfm":
        with torch.no_grad():
            tok, _ = O.tfm_news_encoder(P, X["cdd_token_id"], X["cdd_attn_mask"], H)
        assert relerr(tok, Og["cdd_token_emb"]) < TOL


@pytest.mark.parametrize("ne,ue", PAIRS)
def test_pair_matches_reference(ne, ue):
    _run(f"pair_{ne}_{ue}", ne, ue)


def test_nrms_shape_matches_reference():
    _run("nrms300_tfm_tfm", "tfm", "tfm")


def test_cnn_rnn_mid_matches_reference():
    _run("cnn_rnn_mid", "cnn", "rnn")


def test_primitives_match_reference():
    z = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "primitives.npz"))
    t = lambda k: torch.from_numpy(z[k])
    D, S, N, H = [int(v) for v in z["meta"]]
    out, prob = O.query_pooling(t("q"), t("key"), t("val"), t("mask"), return_prob=True)
    # same op sequence => bit-identical, including the fully masked row's 2^-7 quantisation
    assert torch.equal(prob, t("prob"))
    assert torch.equal(out, t("out"))
    assert not torch.allclose(prob[0], torch.full_like(prob[0], 1.0 / S), rtol=1e-3), \
        "fully masked row must NOT be uniform (SURVEY 2b-iv trap)"
    assert torch.equal(O.query_pooling(t("q"), t("key"), t("val")), t("out_nomask"))
    P = {k[2:]: t(k) for k in z.files if k.startswith("w.")}
    y = O.tfm_layer(P, "", t("x"), t("mask"), H)
    assert relerr(y, t("y")) < 1e-6


def test_metrics_match_reference():
    z = np.load(__import__("os").path.join(__import__("helpers").GOLDEN, "metrics.npz"))
    labels, preds = O.pack_results(z["impr_indices"].tolist(), z["masks"], z["labels"], z["preds"])
    sizes = z["group_sizes"]
    assert [len(l) for l in labels] == sizes.tolist()
    assert np.array_equal(np.concatenate(labels), z["group_labels"])
    assert np.array_equal(np.concatenate(preds).astype(np.float32), z["group_preds"])
    per = np.asarray([[O.auc(np.asarray(l), np.asarray(p)), O.mrr(np.asarray(l), np.asarray(p)),
                       O.ndcg(np.asarray(l), np.asarray(p), 5), O.ndcg(np.asarray(l), np.asarray(p), 10)]
                      for l, p in zip(labels, preds)])
    assert np.abs(per - z["per_impression"]).max() < 1e-12
    res = O.compute_metrics(labels, preds)
    assert [res["auc"], res["mean_mrr"], res["ndcg@5"], res["ndcg@10"]] == z["result"].tolist()
    for ws, r, s, e, n in z["shards"]:
        assert O.sequential_shard(int(n), int(ws), int(r)) == (int(s), int(e))


@pytest.mark.parametrize("ue", ["attn", "tfm"])
def test_bert_pair_matches_reference(ue):
    """AllBertNewsEncoder (HF BertModel, transformers 5.15.0 in the build container): third-party arithmetic,
    pinned only through these vectors (tests/golden/make_golden_bert.py)."""
    _run(f"pair_bert_{ue}", "bert", ue)
