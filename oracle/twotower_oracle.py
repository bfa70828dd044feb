"""CPU oracle for the two-tower hot path.  TEST INFRASTRUCTURE ONLY.

This file restates, in plain functional torch-fp32 (CPU), the arithmetic that
namespace-Pt/News-Recommendation performs on its two-tower path.  It is NOT
part of the product: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it, and only as the checker /
reported baseline.  The product path (``news-recommendation_amd/``) never
imports this module and raises if its HIP library is missing.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from ``/root/reference/src`` (possible only in the build container)
and dumps inputs/weights/outputs to ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks every function here against them.
Two third-party boundaries stay "parity unpinned" except through that import
(transformers 5.15.0 BertModel for AllBert, sklearn 1.7.2 roc_auc_score).

Every function cites the reference file:line (relative to /root/reference/src)
whose behaviour it follows.  Parameters are passed as flat dicts keyed by the
reference's state-dict names so golden weights drop straight in.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

MASK_FILL = -1e5  # models/modules/attention.py:24,49 — additive, finite, NOT -inf


# --------------------------------------------------------------------------
# (i) token-embedding gather — models/modules/encoder.py:52,135
# --------------------------------------------------------------------------
PAD_ID = 0  # manager.py:138 "[PAD]": 0 == BertConfig.pad_token_id


def embedding_lookup(table, ids, padding_idx=PAD_ID):
    """rows of `table` [V,E] picked by int64 `ids` [...]; bit-exact copy.

    The table is HF BERT's ``word_embeddings`` module (encoder.py:29-30,
    :119-120), i.e. nn.Embedding(V, E, padding_idx=pad_token_id): the forward
    is a plain gather, the backward accumulates NO gradient into the PAD row.
    """
    return F.embedding(ids, table, padding_idx=padding_idx)


# --------------------------------------------------------------------------
# (iv) single-query additive pooling — models/modules/attention.py:6-34
# --------------------------------------------------------------------------
def additive_mask(mask):
    """(1 - mask) * -1e5 as fp32 — attention.py:24 / :49 (int64 mask times a
    python float promotes to the default float dtype)."""
    return (1 - mask) * MASK_FILL


def query_pooling(query, key, value, mask=None, return_prob=False):
    """softmax(q·keyᵀ/√D + (1-mask)·(-1e5))·value.

    query [1,D]; key,value [...,S,D]; mask [...,S] (0/1) or None.
    attention.py:19-29.  The mask add happens in fp32 BEFORE the softmax's
    max-subtraction, so fully masked rows see scores quantised to 2^-7.
    """
    score = torch.matmul(query, key.transpose(-2, -1)) / math.sqrt(query.shape[-1])
    if mask is not None:
        score = score + additive_mask(mask).unsqueeze(-2)
    prob = torch.softmax(score, -1)
    out = torch.matmul(prob, value)
    return (out, prob) if return_prob else out


# --------------------------------------------------------------------------
# (iii) post-LN transformer layer — models/modules/attention.py:55-182
# --------------------------------------------------------------------------
def _drop(x, keep_mask, p):
    """inverted dropout with an injected keep-mask (None ⇒ eval / identity)."""
    if keep_mask is None:
        return x
    # torch's dropout multiplies by the noise tensor bernoulli(1-p).div_(1-p)
    return x * (keep_mask.to(x.dtype) / (1.0 - p))


def tfm_layer(P, prefix, x, mask, head_num, drop_masks=None, p=0.1, return_all=False, hf_mask=False):
    """One BERT-style layer.  x [N,S,D], mask [N,S] (0/1 ints).
    hf_mask=False: the reference's own TFMLayer — masked keys get an additive -1e5 (attention.py:44-49).
    hf_mask=True: HF BertModel under torch SDPA (bert_news_encoder below) — masked keys are excluded
    (-inf) and a sequence with NO valid key gets all-zero probabilities (torch's safe softmax), i.e. a zero
    context vector; checked against transformers 5.15.0 / torch 2.10 in tests/golden/pair_bert_*.npz.

    P[prefix + name] uses the reference's keys (attention.self.query.weight …).
    drop_masks: None (eval) or dict with boolean keep-masks 'probs' [N,H,S,S],
    'attn_out' [N,S,D], 'ffn_out' [N,S,D] — the three nn.Dropout sites
    (attention.py:92, :111, :156).
    """
    g = lambda n: P[prefix + n]
    N, S, D = x.shape
    dh = int(D / head_num)                                   # attention.py:60
    A = head_num * dh
    dm = drop_masks or {}

    def heads(t):                                            # attention.py:71-74
        return t.view(N, S, head_num, dh).permute(0, 2, 1, 3)

    q = heads(F.linear(x, g("attention.self.query.weight"), g("attention.self.query.bias")))
    k = heads(F.linear(x, g("attention.self.key.weight"), g("attention.self.key.bias")))
    v = heads(F.linear(x, g("attention.self.value.weight"), g("attention.self.value.bias")))
    scores = torch.matmul(q, k.transpose(-1, -2))            # attention.py:87
    scores = scores / math.sqrt(dh)                          # attention.py:89
    if hf_mask:
        scores = scores.masked_fill(mask[:, None, None, :] == 0, float("-inf"))
        probs = torch.nan_to_num(torch.softmax(scores, dim=-1), nan=0.0)
    else:
        ext = additive_mask(mask)[:, None, None, :]          # attention.py:44-49
        probs = torch.softmax(scores + ext, dim=-1)          # attention.py:90
    probs_d = _drop(probs, dm.get("probs"), p)               # attention.py:92
    ctx = torch.matmul(probs_d, v).permute(0, 2, 1, 3).contiguous().view(N, S, A)

    h = F.linear(ctx, g("attention.output.dense.weight"), g("attention.output.dense.bias"))
    h = _drop(h, dm.get("attn_out"), p)
    attn_out = F.layer_norm(h + x, (D,), g("attention.output.LayerNorm.weight"),
                            g("attention.output.LayerNorm.bias"), eps=1e-12)   # attention.py:105-113

    z = F.linear(attn_out, g("intermediate.dense.weight"), g("intermediate.dense.bias"))
    a = F.gelu(z)                                            # exact erf GELU, attention.py:139
    h2 = F.linear(a, g("output.dense.weight"), g("output.dense.bias"))
    h2 = _drop(h2, dm.get("ffn_out"), p)
    out = F.layer_norm(h2 + attn_out, (D,), g("output.LayerNorm.weight"),
                       g("output.LayerNorm.bias"), eps=1e-12)                  # attention.py:150-158
    if return_all:
        return out, dict(q=q, k=k, v=v, probs=probs, ctx=ctx, attn_out=attn_out, z=z, a=a)
    return out


# --------------------------------------------------------------------------
# news encoders — models/modules/encoder.py
# --------------------------------------------------------------------------
def cnn_news_encoder(P, token_id, attn_mask, prefix="newsEncoder.", return_all=False):
    """CnnNewsEncoder.forward, encoder.py:48-58.  token_id/attn_mask [...,T]."""
    shp = token_id.shape
    T = shp[-1]
    emb = embedding_lookup(P[prefix + "embedding.weight"], token_id)
    E = emb.shape[-1]
    x = emb.reshape(-1, T, E).transpose(-2, -1)
    y = F.relu(F.conv1d(x, P[prefix + "cnn.weight"], P[prefix + "cnn.bias"], padding=1))
    tok = y.transpose(-2, -1).reshape(*shp, -1)
    key = torch.tanh(F.linear(tok, P[prefix + "newsProject.weight"], P[prefix + "newsProject.bias"]))
    pooled, prob = query_pooling(P[prefix + "news_query"], key, tok, attn_mask, return_prob=True)
    news = pooled.squeeze(-2)
    if return_all:
        return tok, news, dict(emb=emb, key=key, prob=prob)
    return tok, news


def tfm_news_encoder(P, token_id, attn_mask, head_num, prefix="newsEncoder.",
                     drop_masks=None, return_all=False):
    """TfmNewsEncoder.forward, encoder.py:130-143."""
    shp = token_id.shape
    T = shp[-1]
    ids = token_id.reshape(-1, T)
    m = attn_mask.reshape(-1, T)
    emb = embedding_lookup(P[prefix + "embedding.weight"], ids)
    tok = tfm_layer(P, prefix + "transformer.", emb, m, head_num, drop_masks)
    key = torch.tanh(F.linear(tok, P[prefix + "newsProject.weight"], P[prefix + "newsProject.bias"]))
    pooled, prob = query_pooling(P[prefix + "news_query"], key, tok, m, return_prob=True)
    news = pooled.squeeze(-2).reshape(*shp[:-1], -1)
    tok = tok.reshape(*shp, -1)
    if return_all:
        return tok, news, dict(emb=emb, key=key, prob=prob)
    return tok, news


def bert_news_encoder(P, token_id, attn_mask, head_num, prefix="newsEncoder.", drop_masks=None):
    """AllBertNewsEncoder.forward, encoder.py:69-77: HF ``BertModel`` (pooler removed, encoder.py:66) over
    [M,T] ids with a 2-D attention mask, news vector = last_hidden_state[:,0].

    The arithmetic is third-party (transformers, unpinned by the reference; pinned here to 5.15.0 through
    tests/golden/pair_bert_*.npz).  Restated from BertModel's published structure:
      embeddings = LayerNorm(word[ids] + position[0..T-1] + token_type[0]; eps=1e-12), dropout
      num_hidden_layers x post-LN layer — the same layer as the reference's TFMLayer (attention.py:161-182,
      which copies BERT's parameter names), except for the key mask (tfm_layer, hf_mask=True).
    drop_masks: None (eval) or dict: 'emb' [M,T,D] and per layer i 'L{i}' -> tfm_layer's dict."""
    shp = token_id.shape
    T = shp[-1]
    ids = token_id.reshape(-1, T)
    m = attn_mask.reshape(-1, T)
    e = prefix + "plm.embeddings."
    D = P[e + "word_embeddings.weight"].shape[1]
    x = embedding_lookup(P[e + "word_embeddings.weight"], ids)
    x = x + P[e + "token_type_embeddings.weight"][0] + P[e + "position_embeddings.weight"][:T]
    x = F.layer_norm(x, (D,), P[e + "LayerNorm.weight"], P[e + "LayerNorm.bias"], eps=1e-12)
    dm = drop_masks or {}
    x = _drop(x, dm.get("emb"), 0.1)
    i = 0
    while f"{prefix}plm.encoder.layer.{i}.output.dense.weight" in P:
        x = tfm_layer(P, f"{prefix}plm.encoder.layer.{i}.", x, m, head_num, dm.get(f"L{i}"), hf_mask=True)
        i += 1
    news = x[:, 0].reshape(*shp[:-1], -1)
    return x.reshape(*shp, -1), news


# --------------------------------------------------------------------------
# user encoders — models/modules/encoder.py:204-327
# --------------------------------------------------------------------------
def gru_cell(x_t, h, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.GRU cell equations (gate order r,z,n)."""
    gi = F.linear(x_t, w_ih, b_ih)
    gh = F.linear(h, w_hh, b_hh)
    i_r, i_z, i_n = gi.chunk(3, -1)
    h_r, h_z, h_n = gh.chunk(3, -1)
    r = torch.sigmoid(i_r + h_r)
    z = torch.sigmoid(i_z + h_z)
    n = torch.tanh(i_n + r * h_n)
    return (1 - z) * n + z * h


def rnn_user_encoder(P, news_emb, his_mask, prefix="userEncoder."):
    """RnnUserEncoder.forward, encoder.py:216-233: the GRU state after
    len_b = his_mask[b].sum() steps (pack_padded_sequence semantics)."""
    B, L, D = news_emb.shape
    lens = his_mask.sum(-1)
    w_ih, w_hh = P[prefix + "rnn.weight_ih_l0"], P[prefix + "rnn.weight_hh_l0"]
    b_ih, b_hh = P[prefix + "rnn.bias_ih_l0"], P[prefix + "rnn.bias_hh_l0"]
    h = news_emb.new_zeros(B, D)
    for t in range(L):
        h_new = gru_cell(news_emb[:, t], h, w_ih, w_hh, b_ih, b_hh)
        live = (lens > t).unsqueeze(-1)
        h = torch.where(live, h_new, h)
    return h.unsqueeze(1)


def sum_user_encoder(P, news_emb, his_mask=None, prefix="userEncoder."):
    return news_emb.sum(dim=-2, keepdim=True)                # encoder.py:253 (mask ignored)


def avg_user_encoder(P, news_emb, his_mask=None, prefix="userEncoder."):
    return news_emb.mean(dim=-2, keepdim=True)               # encoder.py:274 (mask ignored)


def attn_user_encoder(P, news_emb, his_mask=None, prefix="userEncoder."):
    # encoder.py:298 — key = value = history, no projection, NO mask
    return query_pooling(P[prefix + "user_query"], news_emb, news_emb)


def tfm_user_encoder(P, news_emb, his_mask, head_num, prefix="userEncoder.", drop_masks=None):
    """TfmUserEncoder.forward, encoder.py:314-327."""
    h = tfm_layer(P, prefix + "transformer.", news_emb, his_mask, head_num, drop_masks)
    key = torch.tanh(F.linear(h, P[prefix + "userProject.weight"], P[prefix + "userProject.bias"]))
    return query_pooling(P[prefix + "user_query"], key, h, his_mask)


def encode_news(P, kind, token_id, attn_mask, head_num=None, drop_masks=None):
    if kind == "cnn":
        return cnn_news_encoder(P, token_id, attn_mask)[1]
    if kind == "tfm":
        return tfm_news_encoder(P, token_id, attn_mask, head_num, drop_masks=drop_masks)[1]
    if kind == "bert":
        return bert_news_encoder(P, token_id, attn_mask, head_num, drop_masks=drop_masks)[1]
    raise ValueError(kind)


def encode_user(P, kind, his_emb, his_mask, head_num=None, drop_masks=None):
    if kind == "rnn":
        return rnn_user_encoder(P, his_emb, his_mask)
    if kind == "sum":
        return sum_user_encoder(P, his_emb)
    if kind == "avg":
        return avg_user_encoder(P, his_emb)
    if kind == "attn":
        return attn_user_encoder(P, his_emb)
    if kind == "tfm":
        return tfm_user_encoder(P, his_emb, his_mask, head_num, drop_masks=drop_masks)
    raise ValueError(kind)


# --------------------------------------------------------------------------
# (vii) scoring + loss — models/BaseModel.py:164-205
# --------------------------------------------------------------------------
def compute_logits(cdd, user):
    """[B,C,D]·[B,1,D]ᵀ / √D → [B,C]   (BaseModel.py:174)."""
    return cdd.matmul(user.transpose(-2, -1)).squeeze(-1) / math.sqrt(cdd.size(-1))


def ce_loss(logits, label):
    """nn.CrossEntropyLoss() default (mean over B) — BaseModel.py:41,204."""
    return F.cross_entropy(logits, label)


def train_forward(P, news_kind, user_kind, batch, head_num=None,
                  news_drop=None, his_drop=None, user_drop=None):
    """TwoTowerBaseModel.forward, BaseModel.py:198-205.  Returns (loss, aux)."""
    cdd = encode_news(P, news_kind, batch["cdd_token_id"], batch["cdd_attn_mask"], head_num, news_drop)
    his = encode_news(P, news_kind, batch["his_token_id"], batch["his_attn_mask"], head_num, his_drop)
    user = encode_user(P, user_kind, his, batch["his_mask"], head_num, user_drop)
    logits = compute_logits(cdd, user)
    loss = ce_loss(logits, batch["label"])
    return loss, dict(cdd=cdd, his=his, user=user, logits=logits)


# --------------------------------------------------------------------------
# (viii) cached evaluation — models/BaseModel.py:208-243
# --------------------------------------------------------------------------
def build_news_cache(P, news_kind, token_ids, attn_masks, head_num=None, batch=200):
    """encode_news over the whole [Nn,T] table in batches of `batch` rows
    (BaseModel.py:224-234); row 0 (all PAD) is encoded like any other."""
    outs = []
    for s in range(0, token_ids.shape[0], batch):
        outs.append(encode_news(P, news_kind, token_ids[s:s + batch], attn_masks[s:s + batch], head_num))
    return torch.cat(outs, 0)


def infer(P, user_kind, cache, cdd_idx, his_idx, his_mask, head_num=None):
    """TwoTowerBaseModel.infer, BaseModel.py:212-218."""
    cdd = cache[cdd_idx]
    his = cache[his_idx]
    user = encode_user(P, user_kind, his, his_mask, head_num)
    return compute_logits(cdd, user)


# --------------------------------------------------------------------------
# metrics & result packing — utils/util.py:37-57,126-158,179-250,269-282
# --------------------------------------------------------------------------
def sequential_shard(n, world_size, rank):
    """Sequential_Sampler's [start,end) — util.py:272-274 (python round())."""
    per = n / world_size
    return round(per * rank), round(per * (rank + 1))


def pack_results(impr_indices, masks, *lists):
    """group per-row arrays by impression id keeping first-seen order and
    dropping masked-out candidates — util.py:37-57."""
    groups = [OrderedDict() for _ in lists]
    for row, key in enumerate(impr_indices):
        m = np.asarray(masks[row], dtype=bool)
        for gi, lst in enumerate(lists):
            groups[gi].setdefault(key, []).extend(np.asarray(lst[row])[m].tolist())
    return [list(g.values()) for g in groups]


def _desc_order(y_score):
    return np.argsort(y_score)[::-1]                          # util.py:137,190 (unstable sort, reversed)


def mrr(y_true, y_score):
    y = np.take(y_true, _desc_order(y_score))                 # util.py:137-143
    return np.sum(y / (np.arange(len(y)) + 1)) / np.sum(y)


def dcg(y_true, y_score, k):
    k = min(np.shape(y_true)[-1], k)                          # util.py:189-194
    y = np.take(y_true, _desc_order(y_score)[:k])
    return np.sum((2 ** y - 1) / np.log2(np.arange(len(y)) + 2))


def ndcg(y_true, y_score, k):
    return dcg(y_true, y_score, k) / dcg(y_true, y_true, k)   # util.py:156-158


def auc(y_true, y_score):
    """ROC AUC with ties counted ½ (Mann-Whitney U) — what
    sklearn.metrics.roc_auc_score (util.py:205) returns for binary labels."""
    y_true = np.asarray(y_true)
    s = np.asarray(y_score, dtype=np.float64)
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), dtype=np.float64)
    ss = s[order]
    i = 0
    while i < len(ss):
        j = i
        while j + 1 < len(ss) and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    pos = y_true == 1
    n_pos, n_neg = pos.sum(), (~pos).sum()
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return (ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def compute_metrics(labels, preds, metrics=("auc", "mean_mrr", "ndcg@5", "ndcg@10")):
    """per-impression metric, mean over impressions, round(…,4) — util.py:197-250."""
    res = {}
    pairs = [(np.asarray(l), np.asarray(p)) for l, p in zip(labels, preds)]
    for m in metrics:
        if m == "auc":
            res["auc"] = round(float(np.mean([auc(l, p) for l, p in pairs])), 4)
        elif m == "mean_mrr":
            res["mean_mrr"] = round(float(np.mean([mrr(l, p) for l, p in pairs])), 4)
        elif m.startswith("ndcg"):
            ks = [1, 2]
            if "@" in m:
                ks = [int(t) for t in m.split("@")[1].split(";")]
            for k in ks:
                res[f"ndcg@{k}"] = round(float(np.mean([ndcg(l, p, k) for l, p in pairs])), 4)
        else:
            raise ValueError(f"not define this metric {m}")
    return res


# --------------------------------------------------------------------------
# (ix) Adam — models/BaseModel.py:46 (torch.optim.Adam defaults)
# --------------------------------------------------------------------------
def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """one dense Adam update, torch.optim.Adam single-tensor formulation."""
    m.lerp_(g, 1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))
