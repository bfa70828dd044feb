"""Host-side helpers of the two-tower path: result packing, ranking metrics, samplers, Adam driver.

Follows the reference's utils/util.py for everything that defines the AUC / MRR / nDCG parity
target (pack_results :37-57, mrr/dcg/ndcg :126-194, compute_metrics :197-266, Sequential_Sampler
:269-282, sample_news :60-76).  Rankings call the same numpy primitive on the same fp32 scores
(np.argsort(y_score)[::-1]) so tie order is identical by construction.
"""
import math
import pickle
from random import sample

import numpy as np
import torch


def load_pickle(path):
    """only used on caches this package (or the user's own preprocessing) wrote"""
    with open(path, "rb") as f:
        return pickle.load(f)


def save_pickle(obj, path):
    with open(path, "wb") as f:
        pickle.dump(obj, f)


def pack_results(impr_indices, masks, *associated_lists):
    """regroup per-row arrays (rows = <=impr_size-candidate chunks of an impression) by impression id,
    first-seen order, dropping masked-out (padded) candidates.  Vectorised; returns, per associated
    array, a list of python lists exactly like the reference."""
    impr = np.asarray(impr_indices).reshape(-1)
    masks = np.asarray(masks, dtype=np.bool_)
    n = impr.shape[0]
    if n == 0:
        return [[] for _ in associated_lists]
    uniq, first, inv = np.unique(impr, return_index=True, return_inverse=True)
    group_rank = np.argsort(np.argsort(first, kind="stable"), kind="stable")[inv]     # first-seen order of each row's group
    order = np.argsort(group_rank, kind="stable")                                   # rows grouped, original order inside
    counts_row = masks.sum(axis=1)
    sizes = np.bincount(group_rank, weights=counts_row, minlength=len(uniq)).astype(np.int64)
    bounds = np.cumsum(sizes)[:-1]
    out = []
    for arr in associated_lists:
        arr = np.asarray(arr)
        flat = arr[order][masks[order]]
        out.append([g.tolist() for g in np.split(flat, bounds)])
    return out


def sample_news(news, k):
    """k negatives without replacement; zero-pad when fewer than k exist -> (list, valid_count)"""
    num = len(news)
    if k > num:
        return news + [0] * (k - num), num
    return sample(news, k), k


def _order(y_score):
    return np.argsort(y_score)[::-1]


def mrr_score(y_true, y_score):
    y = np.take(y_true, _order(y_score))
    return np.sum(y / (np.arange(len(y)) + 1)) / np.sum(y)


def dcg_score(y_true, y_score, k=10):
    k = min(np.shape(y_true)[-1], k)
    y = np.take(y_true, _order(y_score)[:k])
    return np.sum((2 ** y - 1) / np.log2(np.arange(len(y)) + 2))


def ndcg_score(y_true, y_score, k=10):
    return dcg_score(y_true, y_score, k) / dcg_score(y_true, y_true, k)


def hit_score(y_true, y_score, k=10):
    truth = set(np.where(np.asarray(y_true) == 1)[0].tolist())
    return int(any(int(i) in truth for i in _order(y_score)[:k]))


def auc_score(y_true, y_score):
    """binary ROC AUC with ties counted 1/2 (rank-sum form of what sklearn.metrics.roc_auc_score,
    the reference's call at util.py:205, returns)."""
    y_true = np.asarray(y_true)
    s = np.asarray(y_score, dtype=np.float64)
    n = s.shape[0]
    order = np.argsort(s, kind="stable")
    ss = s[order]
    new = np.empty(n, dtype=bool)
    new[0] = True
    new[1:] = ss[1:] != ss[:-1]
    start = np.flatnonzero(new)
    end = np.append(start[1:], n)
    avg = 0.5 * (start + end - 1) + 1.0                  # average 1-based rank of each tie group
    ranks = np.empty(n, dtype=np.float64)
    ranks[order] = np.repeat(avg, end - start)
    pos = y_true == 1
    n_pos = int(pos.sum())
    n_neg = n - n_pos
    if n_pos == 0 or n_neg == 0:
        raise ValueError("Only one class present in y_true. ROC AUC score is not defined in that case.")
    return (ranks[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg)


def compute_metrics(labels, preds, metrics):
    """mean per-impression metric, round(…, 4); keys: auc, mean_mrr, ndcg@k[;k…], hit@k[;k…]"""
    pairs = [(np.asarray(l), np.asarray(p)) for l, p in zip(labels, preds)]
    res = {}
    for metric in metrics:
        if metric == "auc":
            res["auc"] = round(np.mean([auc_score(l, p) for l, p in pairs]), 4)
        elif metric == "mean_mrr":
            res["mean_mrr"] = round(np.mean([mrr_score(l, p) for l, p in pairs]), 4)
        elif metric.startswith("ndcg") or metric.startswith("hit"):
            fn, tag = (ndcg_score, "ndcg") if metric.startswith("ndcg") else (hit_score, "hit")
            ks = [1, 2]
            parts = metric.split("@")
            if len(parts) > 1:
                ks = [int(t) for t in parts[1].split(";")]
            for k in ks:
                res["{}@{}".format(tag, k)] = round(np.mean([fn(l, p, k) for l, p in pairs]), 4)
        else:
            raise ValueError("not define this metric {0}".format(metric))
    return res


def parse_metrics(metrics):
    """-> (want_auc, want_mrr, ndcg cut-offs) or None when a metric the device kernel does not produce is asked for"""
    want_auc = want_mrr = False
    ks = []
    for metric in metrics:
        if metric == "auc":
            want_auc = True
        elif metric == "mean_mrr":
            want_mrr = True
        elif metric.startswith("ndcg"):
            parts = metric.split("@")
            ks += [int(t) for t in parts[1].split(";")] if len(parts) > 1 else [1, 2]
        else:
            return None
    ks = list(dict.fromkeys(ks))
    return (want_auc, want_mrr, ks) if len(ks) <= 4 else None


def device_compute_metrics(impr, masks, preds, labels, metrics):
    """compute_metrics(pack_results(...)) without leaving the GPU: `impr` [R] int64, `masks`
    [R,Ce] bool, `preds` [R,Ce] f32, `labels` [R,Ce] int64 are the device tensors the scoring pass produced.
    Returns the same dict as compute_metrics, or None when the host path has to run (metrics the kernel does
    not know, impressions whose rows are not adjacent, > 1024 candidates, non-binary labels).
    Impressions with an exact score tie get their MRR / nDCG recomputed on the host with the reference's numpy
    expression (np.argsort(...)[::-1]: its tie order is numpy's own), so the result equals the host path's."""
    from newsrec_hip import ops
    spec = parse_metrics(metrics)
    if spec is None or impr.numel() == 0:
        return None
    want_auc, want_mrr, ks = spec
    new = torch.ones_like(impr, dtype=torch.bool)
    new[1:] = impr[1:] != impr[:-1]
    starts = torch.nonzero(new).flatten()
    if torch.unique(impr).numel() != starts.numel():
        return None                                   # an impression's rows are not adjacent: pack_results' regrouping needed
    counts = torch.diff(torch.cat([starts, torch.tensor([impr.numel()], device=impr.device)])).to(torch.int32)
    out, flags = ops.impression_metrics(preds, labels, masks, starts, counts, ks)
    flags = flags.cpu().numpy()
    if (flags & 14).any():
        return None
    out = out.cpu().numpy()
    tied = np.flatnonzero(flags & 1)
    if tied.size and (want_mrr or ks):
        st, ct = starts.cpu().numpy(), counts.cpu().numpy()
        rows = np.concatenate([np.arange(st[g], st[g] + ct[g]) for g in tied])
        ridx = torch.from_numpy(rows).to(impr.device)
        m_h, p_h, l_h = masks[ridx].cpu().numpy().astype(np.bool_), preds[ridx].cpu().numpy(), labels[ridx].cpu().numpy()
        pos = 0
        for g in tied:
            sl = slice(pos, pos + ct[g])
            pos += ct[g]
            y, sc = l_h[sl][m_h[sl]].astype(np.int32), p_h[sl][m_h[sl]].astype(np.float32)
            out[g, 1] = mrr_score(y, sc)
            for c, k in enumerate(ks):
                out[g, 2 + c] = ndcg_score(y, sc, k)
    res = {}
    for metric in metrics:
        if metric == "auc":
            res["auc"] = round(np.mean(out[:, 0]), 4)
        elif metric == "mean_mrr":
            res["mean_mrr"] = round(np.mean(out[:, 1]), 4)
        else:
            parts = metric.split("@")
            for k in ([int(t) for t in parts[1].split(";")] if len(parts) > 1 else [1, 2]):
                res["ndcg@{}".format(k)] = round(np.mean(out[:, 2 + ks.index(k)]), 4)
    return res


class frozen_gc:
    """`with frozen_gc():` around a long loop: what is alive now (datasets, model, optimizer state — millions of Python
    objects) moves to the garbage collector's permanent generation, so the collections that the loop's own garbage
    triggers stay in the microseconds.  A full collection over those objects was measured at 65-80 ms, every few hundred
    steps of a 2.3 ms step.  Re-entrant: the outermost exit thaws."""
    depth = 0

    def __init__(self, collect=True):
        self.collect = collect                # a full collection first (tens of ms): worth it once per training run

    def __enter__(self):
        import gc
        if frozen_gc.depth == 0:
            if self.collect:
                gc.collect()
            gc.freeze()
        frozen_gc.depth += 1
        return self

    def __exit__(self, *exc):
        import gc
        frozen_gc.depth -= 1
        if frozen_gc.depth == 0:
            gc.unfreeze()
        return False


class Sequential_Sampler:
    """rank r reads the contiguous slice [round(n/ws*r), round(n/ws*(r+1)))"""

    def __init__(self, dataset_length, num_replicas, rank) -> None:
        per = dataset_length / num_replicas
        self.start = round(per * rank)
        self.end = round(per * (rank + 1))

    def __iter__(self):
        return iter(range(self.start, self.end))

    def __len__(self):
        return self.end - self.start


def linear_warmup_lambda(num_warmup_steps, num_training_steps):
    """lr multiplier of transformers.get_linear_schedule_with_warmup (reference BaseModel.py:51-53)"""
    def fn(step):
        if step < num_warmup_steps:
            return float(step) / float(max(1, num_warmup_steps))
        return max(0.0, float(num_training_steps - step) / float(max(1, num_training_steps - num_warmup_steps)))
    return fn


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam(params, lr) with its defaults (betas 0.9/0.999, eps 1e-8, no weight decay),
    dense over every parameter incl. the V x E token table (reference BaseModel.py:46); the update
    itself is the fused HIP kernel nr_adam_step_f32.  State keys match torch's Adam."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables = {}                     # the moments are new tensors now

    def __getstate__(self):
        """the cached ctypes pointer tables never leave the process (copy.deepcopy / pickle / torch.save of the optimizer)"""
        st = super().__getstate__()
        st.pop("_tables", None)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        from newsrec_hip import ops
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            b1, b2 = group["betas"]
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            # everything about the step except the gradient pointers is the same from step to step (the same tensors get
            # gradients, parameters and moments do not move): the pointer tables of the fused launch are built once
            # (a parameter moved by .to() / .data = ..., or a moment tensor replaced in self.state by any route, rebuilds them)
            state = self.state
            key = []
            for p in ps:
                st = state.get(p)
                key.append((id(p), p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) if st else (id(p), p.data_ptr(), 0, 0))
            key = tuple(key)
            cache = getattr(self, "_tables", {}).get(gi)
            if cache is None or cache["key"] != key:
                for p in ps:
                    st = self.state[p]
                    if not st:
                        st["step"] = 0
                        st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                        st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    if not p.is_contiguous():     # the kernel walks flat memory
                        raise RuntimeError("HipAdam: parameters must be contiguous")
                states = [self.state[p] for p in ps]
                key = tuple((id(p), p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()) for p, st in zip(ps, states))
                cache = {"key": key, "states": states,
                         "static": ops.adam_tables(ps, [st["exp_avg"] for st in states], [st["exp_avg_sq"] for st in states])}
                if not hasattr(self, "_tables"):
                    self._tables = {}
                self._tables[gi] = cache
            states = cache["states"]
            steps = set()
            for st in states:
                st["step"] += 1
                steps.add(int(st["step"]))
            gs = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in ps]
            if len(steps) == 1:
                ops.adam_multi_cached(cache["static"], gs, steps.pop(), float(group["lr"]), b1, b2, group["eps"])
            else:                             # tensors that share a step number go into one launch
                by_step = {}
                for p, g, st in zip(ps, gs, states):
                    by_step.setdefault(int(st["step"]), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
                for step, items in by_step.items():
                    a, b, c, d = zip(*items)
                    ops.adam_multi(a, b, c, d, step, float(group["lr"]), b1, b2, group["eps"])
        return loss
