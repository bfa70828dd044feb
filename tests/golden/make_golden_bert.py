"""Golden fixtures for the BERT news encoder (BASELINE config 5), made by importing the REAL reference
(AllBertNewsEncoder, encoder.py:62-77) over a small random-init HF BertModel directory.  Build container
only; see make_golden.py for the rules (outputs are data, nothing of the reference is copied).

    python tests/golden/make_golden_bert.py        # writes tests/golden/pair_bert_*.npz

The arithmetic of this encoder lives in HF transformers (third party, unpinned by the reference): these
vectors pin it to the transformers version of this container (printed into the fixture's meta)."""
import os
import shutil
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG                                   # noqa: E402  (imports the reference)
from make_golden import R_enc, TwoTowerModel               # noqa: E402


def make_bert_dir(root, vocab, dim, heads, layers):
    from transformers import BertConfig, BertModel
    cfg = BertConfig(vocab_size=vocab, hidden_size=dim, num_hidden_layers=layers, num_attention_heads=heads,
                     intermediate_size=4 * dim, max_position_embeddings=64)
    d = os.path.join(root, f"bert_{vocab}_{dim}_{layers}")
    torch.manual_seed(4321)
    BertModel(cfg).save_pretrained(d)
    return d


def dump(tmp, tag, ue, V, D, H, NL, T, L, K, B, Nn, seed):
    import transformers
    plm = make_bert_dir(tmp, V, D, H, NL)
    mgr = MG.fake_manager(plm, D, D, H, L, T, K)
    torch.manual_seed(seed)
    model = TwoTowerModel(mgr, R_enc.AllBertNewsEncoder(mgr), MG.USER[ue](mgr))
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g))
            if "LayerNorm.weight" in n:
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            if "embeddings.weight" in n or "_embeddings.weight" in n:
                p.copy_(0.5 * torch.randn(p.shape, generator=g))        # HF init std 0.02 would hide embedding bugs
    model.eval()
    ids, msk = MG.synth_news_table(g, Nn, T, V)
    x = MG.synth_batch(g, B, K + 1, L, Nn, ids, msk)
    out = {}
    tok, cdd = model._encode_news(x)
    _, his = model._encode_news(x, cdd=False)
    user = model._encode_user(x)
    logits = model._compute_logits(cdd, user)
    loss = model(x)
    model.zero_grad()
    loss.backward()
    out.update(cdd_news=cdd, his_news=his, user=user, logits=logits, loss=loss, cdd_token_emb=tok)
    with torch.no_grad():
        cache = torch.cat([model.newsEncoder(ids[s:s + 7], msk[s:s + 7])[1] for s in range(0, Nn, 7)], 0)
        model.news_embeddings = cache
        out["news_cache"] = cache
        out["infer_logits"] = model.infer(x)
    blob = {}
    for k, v in x.items():
        blob["in." + k] = v.numpy()
    blob["in.news_token_ids"], blob["in.news_attn_masks"] = ids.numpy(), msk.numpy()
    for k, v in model.state_dict().items():
        blob["w." + k] = v.detach().numpy()
    for k, v in out.items():
        blob["out." + k] = v.detach().numpy()
    for n, p in model.named_parameters():
        blob["g." + n] = (p.grad.detach() if p.grad is not None else torch.zeros_like(p)).numpy()
    blob["meta"] = np.asarray([V, D, D, H, T, L, K, B, Nn, NL], dtype=np.int64)
    blob["transformers_version"] = np.asarray(transformers.__version__)
    blob["attn_implementation"] = np.asarray(str(getattr(model.newsEncoder.plm.config, "_attn_implementation", "?")))
    path = os.path.join(HERE, f"{tag}.npz")
    np.savez_compressed(path, **blob)
    print(f"{tag}: loss={loss.item():.6f} {os.path.getsize(path) / 1024:.1f} KiB attn={blob['attn_implementation']}")
    print("  state keys:", [k for k in model.state_dict() if "layer.1" not in k][:40])
    print("  all-PAD cache row:", cache[0, :4].tolist(), " NaN anywhere:", bool(torch.isnan(cache).any()))


if __name__ == "__main__":
    tmp = tempfile.mkdtemp(prefix="golden_bert_")
    try:
        dump(tmp, "pair_bert_attn", "attn", V=131, D=32, H=4, NL=2, T=8, L=5, K=2, B=3, Nn=23, seed=400)
        dump(tmp, "pair_bert_tfm", "tfm", V=131, D=32, H=4, NL=2, T=8, L=5, K=2, B=3, Nn=23, seed=401)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
