"""shared test helpers: golden loading, synthetic MIND-shaped inputs, error norms."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(tag):
    z = np.load(os.path.join(GOLDEN, tag + ".npz"), allow_pickle=False)
    blob = {k: z[k] for k in z.files}
    meta = dict(zip("V E D H T L K B Nn NL".split(), [int(v) for v in blob["meta"]])) if "meta" in blob and len(blob["meta"]) in (9, 10) else {}
    if "w_seed" in blob:
        from golden.seeded import seeded_state
        names = [str(n) for n in blob["w_names"]]
        shapes = [tuple(int(t) for t in str(s).split(";")) if str(s) else () for s in blob["w_shapes"]]
        P = seeded_state(zip(names, shapes), int(blob["w_seed"]))
        for n, ref in zip(names, blob["w_abs_sum"]):
            got = P[n].double().abs().sum().item()
            assert abs(got - float(ref)) <= 1e-9 * max(1.0, abs(float(ref))), f"seeded weight drift in {n}"
    else:
        P = {k[2:]: torch.from_numpy(v) for k, v in blob.items() if k.startswith("w.")}
    X = {k[3:]: torch.from_numpy(v) for k, v in blob.items() if k.startswith("in.")}
    O = {k[4:]: torch.from_numpy(np.asarray(v)) for k, v in blob.items() if k.startswith("out.")}
    G = {k[2:]: torch.from_numpy(np.asarray(v)) for k, v in blob.items() if k.startswith("g.")}
    extra = {k: blob[k] for k in blob if k.startswith("g_step.") or k.startswith("g_abs_sum.")}
    return meta, P, X, O, G, extra


def relerr(a, b):
    """max-norm relative error ‖a-b‖∞ / ‖b‖∞ (the 1e-5 bar of BASELINE.md §4)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / max(denom, 1e-30)


def synth_news_table(g, Nn, T, V, empty_every=11):
    """MIND-shaped token table: row 0 all PAD, [CLS]=101 … [SEP]=102 then PAD
    (mirrors utils/dataset.py:68-103 of the reference)."""
    ids = torch.zeros(Nn, T, dtype=torch.int64)
    lens = torch.randint(5, T + 1, (Nn,), generator=g)
    lens[0] = 0
    if empty_every:
        lens[7::empty_every] = 0
    body = torch.randint(min(1000, V // 2), V, (Nn, T), generator=g)
    pos = torch.arange(T)[None, :]
    msk = (pos < lens[:, None]).long()
    ids = body * msk
    ids[:, 0] = torch.where(lens > 0, torch.tensor(101 % V), torch.tensor(0))
    last = (lens - 1).clamp(min=0)
    ids[torch.arange(Nn), last] = torch.where(lens > 0, torch.tensor(102 % V), torch.tensor(0))
    return ids, msk


def synth_batch(g, B, C, L, Nn, ids, msk, max_hist=70):
    cdd_idx = torch.randint(1, Nn, (B, C), generator=g)
    npad = torch.randint(0, C, (B,), generator=g)
    for b in range(0, B, 5):                                    # some zero-padded negatives
        k = int(npad[b])
        if k:
            cdd_idx[b, C - k:] = 0
    hl = torch.randint(0, max_hist + 1, (B,), generator=g).clamp(max=L)
    hl[0] = 0
    his_idx = torch.randint(1, Nn, (B, L), generator=g)
    pos = torch.arange(L)[None, :]
    live = pos < hl[:, None]
    his_idx = his_idx * live
    his_mask = live.long()
    his_mask[hl == 0, 0] = 1
    return dict(cdd_idx=cdd_idx, his_idx=his_idx, his_mask=his_mask,
                cdd_token_id=ids[cdd_idx], cdd_attn_mask=msk[cdd_idx],
                his_token_id=ids[his_idx], his_attn_mask=msk[his_idx],
                label=torch.zeros(B, dtype=torch.int64))


# ----------------------------------------------------------------------------- dropout stream of the HIP library
def philox4x32_10(seed, idx, site):
    """numpy restatement of csrc/common.h::philox4x32_10 -> uint32 array [..., 4]"""
    idx = np.asarray(idx, dtype=np.uint64)
    M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
    mask = np.uint64(0xFFFFFFFF)
    k0 = np.uint64(seed & 0xFFFFFFFF)
    k1 = np.uint64((seed >> 32) & 0xFFFFFFFF)
    c0 = idx & mask
    c1 = (idx >> np.uint64(32)) & mask
    c2 = np.full_like(idx, np.uint64(site))
    c3 = np.zeros_like(idx)
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & mask
        hi1, lo1 = p1 >> np.uint64(32), p1 & mask
        c0, c1, c2, c3 = (hi1 ^ c1 ^ k0) & mask, lo1, (hi0 ^ c3 ^ k1) & mask, lo0
        k0 = (k0 + np.uint64(0x9E3779B9)) & mask
        k1 = (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.uint32)


def keep_mask(shape, p, seed, site):
    """boolean keep-mask for a tensor of `shape` (element index = C-order linear index)"""
    n = int(np.prod(shape))
    idx = np.arange(n, dtype=np.uint64)
    r = philox4x32_10(seed, idx >> np.uint64(2), site)
    u = r[np.arange(n), (idx & np.uint64(3)).astype(np.int64)]
    thr = np.uint32(min(max(float(p) * 4294967296.0, 0.0), 4294967295.0))
    return torch.from_numpy((u >= thr).reshape(shape))


def make_manager(**kw):
    import types
    base = dict(plm_dim=24, hidden_dim=24, head_num=4, his_size=5, sequence_length=8, device=0, rank=0, world_size=1,
                negative_num=2, verbose=None, enable_fields=["title"], dropout_p=0.1, vocab_size=131, synthetic_plm=True,
                plm_dir="/nonexistent", distributed=False, debug=False, impr_size=20, batch_size_eval=200,
                metrics=["auc", "mean_mrr", "ndcg@5", "ndcg@10"], learning_rate=1e-3, scheduler="none", epochs=1,
                warmup=0.1, scale="demo", checkpoint="none")
    base.update(kw)
    return types.SimpleNamespace(**base)
