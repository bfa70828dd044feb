"""Model shells of the two-tower path on MI355X.

`BaseModel` (optimizer, dev/test drivers) and `TwoTowerBaseModel` (forward / infer /
encode_news / _dev / _test) keep the reference's method names and semantics
(models/BaseModel.py:17-329) with three deliberate, result-preserving changes that
SURVEY §2a/§8e call for:

  * `forward` fuses `_compute_logits` + `nn.CrossEntropyLoss` into one HIP kernel pair;
  * `encode_news` shards the corpus rows over the ranks and all-gathers the [Nn, D] cache
    (every rank ends with the same table the reference's rank-0-encode + broadcast yields);
  * `_dev` / `_test` gather flat typed tensors instead of pickled python lists.

The one-tower / gating helpers of the reference are outside the two-tower path and are not built.
"""
import logging
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn as nn
import torch.distributed as dist
from tqdm import tqdm

from newsrec_hip import ops
from utils.dp import all_gather_into_tensor_, all_gather_list
from utils.util import pack_results, compute_metrics, device_compute_metrics, HipAdam, linear_warmup_lambda


class BaseModel(nn.Module):
    def __init__(self, manager, name):
        super().__init__()
        self.his_size = manager.his_size
        self.sequence_length = manager.sequence_length
        self.hidden_dim = manager.hidden_dim
        self.device = manager.device
        self.rank = manager.rank
        self.world_size = manager.world_size
        for k, v in vars(manager).items():
            if k.startswith("enable"):
                setattr(self, k, v)
        self.negative_num = manager.negative_num

        name = name or type(self).__name__
        self.name = name if manager.verbose is None else "-".join([name, manager.verbose])
        self.crossEntropy = nn.CrossEntropyLoss()        # kept for API parity; forward() uses the fused kernel
        self.logger = logging.getLogger(self.name)

    def get_optimizer(self, manager, dataloader_length):
        """Adam(lr) over every parameter, optional linear warmup/decay (reference BaseModel.py:45-55).
        The update runs in the HIP Adam kernel; state keys match torch.optim.Adam."""
        optimizer = HipAdam(self.parameters(), lr=manager.learning_rate)
        scheduler = None
        if manager.scheduler == "linear":
            total_steps = dataloader_length * manager.epochs
            warmup = round(manager.warmup * total_steps)
            scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, linear_warmup_lambda(warmup, total_steps))
        return optimizer, scheduler

    @torch.no_grad()
    def dev(self, manager, loaders, log=False):
        self.eval()
        labels, preds = self._dev(manager, loaders)
        metrics = None
        if self.rank == 0:
            metrics = compute_metrics(labels, preds, manager.metrics)
            metrics["main"] = metrics["auc"]
            self.logger.info(metrics)
            if log:
                manager._log(self.name, metrics)
        if manager.distributed:
            _barrier(self.device)
        return metrics

    @torch.no_grad()
    def test(self, manager, loaders, log=False):
        """MIND leaderboard file: per impression, 1-based ranks of the candidates by descending
        score, ties sharing the smallest rank (reference BaseModel.py:126-152)."""
        self.eval()
        preds = self._test(manager, loaders)
        if manager.rank == 0:
            save_dir = "data/cache/results/{}/{}/{}".format(self.name, manager.scale, os.path.split(manager.checkpoint)[-1])
            os.makedirs(save_dir, exist_ok=True)
            save_path = os.path.join(save_dir, "prediction.txt")
            with open(save_path, "w") as f:
                for index, pred in enumerate(preds, 1):
                    f.write("{} [{}]\n".format(index, ",".join(str(int(r)) for r in min_ranks(pred))))
            try:
                subprocess.run(["zip", "-j", os.path.join(save_dir, "prediction.zip"), save_path], check=False)
            except OSError:
                self.logger.warning("Zip Command Not Found! Skip zipping.")
            self.logger.info("written to prediction at {}!".format(save_path))
        if manager.distributed:
            _barrier(self.device)


def min_ranks(pred):
    """1-based ranks of the candidates by descending score, ties sharing the smallest rank: what the reference gets from
    scipy `ss.rankdata(1 - np.asarray(pred), method="min")` (models/BaseModel.py:141; its pred is a list of python floats,
    so `1 - score` is formed in float64), restated as a comparison count."""
    a = 1 - np.asarray(pred, dtype=np.float64)
    return 1 + (a[None, :] < a[:, None]).sum(axis=1)


def _barrier(device):
    if isinstance(device, int) and dist.get_backend() == "nccl":
        dist.barrier(device_ids=[device])
    else:
        dist.barrier()


def _gather_to_rank0(t, world_size, rank):
    """variable-length typed gather: returns the list of every rank's tensor on rank 0 (else None)."""
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [int(s.item()) for s in all_gather_list(n, world_size)]
    cap = max(sizes)
    pad = t.new_zeros((cap,) + tuple(t.shape[1:]))
    pad[:t.shape[0]] = t
    bufs = all_gather_list(pad, world_size)
    if rank != 0:
        return None
    return [b[:s] for b, s in zip(bufs, sizes)]


class TwoTowerBaseModel(BaseModel):
    """two-tower models: every news vector can be cached once and reused when scoring."""
    # cached inference runs a Tfm user tower over the clicked history slots only (TfmUserEncoder.from_cache);
    # NR_PACKED_HISTORY=0 keeps the padded [B, L] layout (A/B measurements)
    packed_history = os.environ.get("NR_PACKED_HISTORY", "1") != "0"

    def __init__(self, manager, name=None):
        super().__init__(manager, name)

    # ------------------------------------------------------------------ scoring
    def _compute_logits(self, cdd_news_repr, user_repr):
        """[B,C,D] x [B,1,D] -> [B,C] dot products / sqrt(D) (reference BaseModel.py:164-175)."""
        return ops.score(cdd_news_repr, user_repr)

    # ------------------------------------------------------------------ device-side batch assembly
    def attach_news_tokens(self, token_ids, attn_masks):
        """keep the corpus' [Nn,T] token-id / mask tables in HBM; batches that carry only news indices
        (datasets built without --host-assembly) are then assembled by a row gather on the device — the
        same rows the reference's MIND.__getitem__ picks with numpy (utils/dataset.py:152-155)."""
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        self._news_token_ids = as_t(token_ids).to(self.device).long().contiguous()
        self._news_attn_masks = as_t(attn_masks).to(self.device).long().contiguous()

    def attach_news_corpus(self, token_ids, attn_masks):
        """the same for the rows of the MIND_News set that encode_news walks (the dev / test corpus, not the training
        one): encode_news then slices them on the device instead of copying every chunk from the host."""
        as_t = lambda a: a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        self._corpus_token_ids = as_t(token_ids).to(self.device).long().contiguous()
        self._corpus_attn_masks = as_t(attn_masks).to(self.device).long().contiguous()

    def _tokens(self, x, key):
        """(token_id, attn_mask) of the `key` ("cdd" | "his") titles of a batch, on the device"""
        if key + "_token_id" in x:
            return x[key + "_token_id"].to(self.device), x[key + "_attn_mask"].to(self.device)
        if getattr(self, "_news_token_ids", None) is None:
            raise RuntimeError("the batch carries no token ids and no token table is attached: call "
                               "model.attach_news_tokens(dataset.token_ids, dataset.attn_masks) or use --host-assembly")
        idx = x[key + "_idx"].to(self.device)
        return ops.gather_rows_i64(self._news_token_ids, idx), ops.gather_rows_i64(self._news_attn_masks, idx)

    def _news_vectors(self, token_id, attn_mask, title_keep=None, pending=None):
        """news vectors only: encoders that offer `news_vectors` skip the padded token rows, and the titles
        `title_keep` marks 0 (same results wherever they are used)"""
        fn = getattr(self.newsEncoder, "news_vectors", None)
        return fn(token_id, attn_mask, title_keep, pending) if fn is not None else self.newsEncoder(token_id, attn_mask)[1]

    def _corpus_vectors(self, x):
        """news vectors of one chunk of the corpus (encode_news); token embeddings are not needed there"""
        return self._news_vectors(*self._tokens(x, "cdd"))

    def _encode_news(self, x, cdd=True):
        token_id, attn_mask = self._tokens(x, "cdd" if cdd else "his")
        return self.newsEncoder(token_id, attn_mask)

    def _encode_user(self, x=None, his_news_embedding=None, his_mask=None):
        if x is not None:
            _, his_news_embedding = self._encode_news(x, cdd=False)
            his_mask = x["his_mask"].to(self.device)
        return self.userEncoder(his_news_embedding, his_mask=his_mask)

    def _train_inputs(self, x):
        """the tensors forward() works on: (block, C, ids, masks, his_mask, keep, pending pack plan).
        block = True: block-major title tensors [B*C + B*L, T] — every candidate title, then every history title —
        assembled with their keep flags by ONE launch (ops.assemble_titles) for news encoders whose news_vectors can hand
        the two groups out separately (`split`); otherwise batch-major [B, C+L, T] through torch concatenations."""
        his_mask = x["his_mask"].to(self.device)
        begin = getattr(self.newsEncoder, "plan_begin", None)
        if "cdd_token_id" in x and begin is not None and getattr(self, "block_major", True) and torch.device(self.device).type == "cuda":
            cdd_ids = x["cdd_token_id"].to(self.device)
            C = cdd_ids.shape[1]
            has_keep = hasattr(self.userEncoder, "history_keep")
            mode = getattr(self.userEncoder, "history_keep_mode", None) if has_keep else 0
            if mode is not None:
                ids, masks, keep = ops.assemble_titles(cdd_ids, x["his_token_id"].to(self.device), x["cdd_attn_mask"].to(self.device),
                                                       x["his_attn_mask"].to(self.device), his_mask, mode)
                if not has_keep:
                    keep = None
                return True, C, ids, masks, his_mask, keep, begin(ids, masks, keep)
        if "cdd_token_id" in x:
            cdd_ids = x["cdd_token_id"].to(self.device)
            C = cdd_ids.shape[1]
            ids = torch.cat([cdd_ids, x["his_token_id"].to(self.device)], dim=1)
            masks = torch.cat([x["cdd_attn_mask"].to(self.device), x["his_attn_mask"].to(self.device)], dim=1)
        else:                               # indices only: one gather for candidates + history
            C = x["cdd_idx"].shape[1]
            ids, masks = self._tokens({"all_idx": torch.cat([x["cdd_idx"], x["his_idx"]], dim=1)}, "all")
        keep = None
        if hasattr(self.userEncoder, "history_keep"):
            # history slots the user encoder masks out cannot reach the loss: as masked keys / skipped GRU steps their
            # weight is exactly 0 and so is every gradient into them -> those titles are not encoded at all
            keep = torch.cat([torch.ones_like(his_mask[:, :1]).expand(-1, C), self.userEncoder.history_keep(his_mask)], dim=1)
        pending = begin(ids, masks, keep) if begin is not None else None
        return False, C, ids, masks, his_mask, keep, pending

    def plan_ahead(self, x):
        """optional look-ahead for a training loop that already holds its NEXT batch: assemble that batch's title tensors
        and queue its pack plan now.  forward(x) on the same dict then finds the row count on the host and the step runs
        without a device->host wait (the only one the packed path has).  Same results with or without this call."""
        self._ahead = (x, self._train_inputs(x))

    def forward(self, x):
        """mean softmax-CE of the clicked candidate (label 0) among 1+K candidates (reference
        BaseModel.py:198-205).  Titles are encoded independently of each other, so the candidate and
        history titles of the batch go through the news encoder as ONE call (the reference makes two;
        per-title results are identical) and are split afterwards."""
        ahead, self._ahead = getattr(self, "_ahead", None), None
        block, C, ids, masks, his_mask, keep, pending = ahead[1] if ahead is not None and ahead[0] is x else self._train_inputs(x)
        if block:
            B, L = his_mask.shape
            cdd, his = self.newsEncoder.news_vectors(ids, masks, keep, pending, split=B * C)
            cdd_news_embedding, his_news_embedding = cdd.view(B, C, -1), his.view(B, L, -1)
        else:
            news = self._news_vectors(ids, masks, keep, pending)
            cdd_news_embedding, his_news_embedding = ops.split_titles(news, C)
        user_embedding = self.userEncoder(his_news_embedding, his_mask=his_mask)
        labels = x["label"].to(self.device)
        loss, _ = ops.score_cross_entropy(cdd_news_embedding, user_embedding, labels)
        return loss

    # ------------------------------------------------------------------ cached evaluation
    def infer(self, x):
        """logits for rows of (cdd_idx [R,Ce], his_idx [R,L], his_mask [R,L]) against the HBM-resident
        news-vector cache (reference BaseModel.py:208-218).  Candidates are never materialised:
        the scoring kernel gathers rows of the cache by index (segmented gather-dot)."""
        cdd_idx = x["cdd_idx"].to(self.device, non_blocking=True)
        his_idx = x["his_idx"].to(self.device, non_blocking=True)
        his_mask = x.get("his_mask")
        # the <=Ce-candidate chunks of one impression share the user's history: encode each user once per run of
        # equal impr_index (the reference re-encodes it for every chunk; same vectors, ~2.5x fewer on MIND)
        inv = None
        if "impr_index" in x and his_idx.shape[0] > 1:
            impr = x["impr_index"].to(self.device)
            first = torch.ones_like(impr, dtype=torch.bool)
            first[1:] = impr[1:] != impr[:-1]
            if not bool(first.all()):
                inv = torch.cumsum(first, 0) - 1
                his_idx = his_idx[first]
                his_mask = his_mask.to(self.device)[first] if his_mask is not None else None
        user = self._user_from_cache(his_idx, his_mask)
        if inv is not None:
            user = ops.gather_rows(user.reshape(user.shape[0], -1), inv)
        return ops.score_from_table(self.news_embeddings, cdd_idx, user)

    def _user_from_cache(self, his_idx, his_mask):
        kind = type(self.userEncoder).__name__
        if kind == "SumUserEncoder":
            user = ops.seq_reduce_from_table(self.news_embeddings, his_idx, mean=False)
        elif kind == "AvgUserEncoder":
            user = ops.seq_reduce_from_table(self.news_embeddings, his_idx, mean=True)
        elif kind == "AttnUserEncoder":
            user = ops.pool_from_table(self.news_embeddings, his_idx, self.userEncoder.user_query)
        else:
            his_mask = his_mask.to(self.device)
            user = None
            if hasattr(self.userEncoder, "from_cache") and getattr(self, "packed_history", True):
                user = self.userEncoder.from_cache(self.news_embeddings, his_idx, his_mask)     # clicked slots only
            if user is None:
                his = ops.gather_rows(self.news_embeddings, his_idx)
                user = self._encode_user(his_news_embedding=his, his_mask=his_mask)
        return user

    def _check_indices(self, where):
        """a token id / news index outside its table raises here (the kernels clamp and flag it; ops.check_indices)"""
        if torch.device(self.device).type == "cuda":
            ops.check_indices(where)

    @torch.no_grad()
    def encode_news(self, manager, loader_news):
        """build the [Nn, D] news-vector cache.  Rows are independent, so each rank encodes the
        contiguous slice round(Nn/ws*r) .. round(Nn/ws*(r+1)) and the slices are all-gathered; the
        table every rank ends with equals the reference's rank-0 encode + broadcast
        (BaseModel.py:221-243), which it replaces.  Token ids are read straight from the dataset's
        [Nn, T] arrays in large chunks (the per-title result does not depend on the chunking)."""
        ds = loader_news.dataset
        Nn = len(ds)
        ws, rank = (self.world_size, self.rank) if manager.distributed else (1, 0)
        per = Nn / ws
        lo, hi = round(per * rank), round(per * (rank + 1))
        local = torch.zeros((hi - lo, self.hidden_dim), dtype=torch.float32, device=self.device)
        chunk = int(getattr(manager, "encode_chunk", 8192))
        stop = min(hi, lo + 7 * loader_news.batch_size) if manager.debug else hi      # --debug: first 7 batches only
        tok = getattr(self, "_corpus_token_ids", None)       # attach_news_corpus(): the table is already in HBM
        on_dev = tok is not None and tok.shape[0] == Nn
        for s in tqdm(range(lo, stop, chunk), ncols=80, desc="Encoding News", disable=self.rank != 0):
            e = min(stop, s + chunk)
            if on_dev:
                x = {"cdd_token_id": tok[s:e], "cdd_attn_mask": self._corpus_attn_masks[s:e]}
            else:
                x = {"cdd_token_id": torch.from_numpy(ds.token_ids[s:e]), "cdd_attn_mask": torch.from_numpy(ds.attn_masks[s:e])}
            local[s - lo:e - lo] = self._corpus_vectors(x)
        self._check_indices("encode_news")
        if manager.distributed:
            cap = max(round(per * (r + 1)) - round(per * r) for r in range(ws))
            pad = torch.zeros((cap, self.hidden_dim), dtype=torch.float32, device=self.device)
            pad[:hi - lo] = local
            gathered = torch.empty((ws * cap, self.hidden_dim), dtype=torch.float32, device=self.device)
            all_gather_into_tensor_(gathered, pad)
            parts = [gathered[r * cap: r * cap + round(per * (r + 1)) - round(per * r)] for r in range(ws)]
            self.news_embeddings = torch.cat(parts, 0).contiguous()
        else:
            self.news_embeddings = local

    def _predict(self, manager, loader, with_labels, to_host=True):
        """score this rank's shard; results stay on the device as flat typed tensors until the end."""
        from utils.util import frozen_gc
        impr, masks, labels, preds = [], [], [], []
        with frozen_gc(collect=False):          # thousands of batches: no full garbage collection over the datasets mid-loop
            for x in tqdm(loader, ncols=80, desc="Predicting", disable=self.rank != 0):
                preds.append(self.infer(x))
                impr.append(x["impr_index"])
                masks.append(x["cdd_mask"])
                if with_labels:
                    labels.append(x["label"])
        self._check_indices("infer")
        dev = self.device
        Ce = manager.impr_size
        cat = lambda ts, shape, dt: (torch.cat([t.reshape(shape) for t in ts]).to(dev) if ts else torch.zeros(shape, dtype=dt)[:0].to(dev))
        out = [cat(impr, (-1,), torch.int64).long(), cat(masks, (-1, Ce), torch.bool).bool(), cat(preds, (-1, Ce), torch.float32)]
        if with_labels:
            out.append(cat(labels, (-1, Ce), torch.int64).long())
        if manager.distributed:
            _barrier(self.device)
            gathered = [_gather_to_rank0(t.contiguous() if t.dtype != torch.bool else t.to(torch.uint8), self.world_size, self.rank) for t in out]
            if self.rank != 0:
                return None
            out = [torch.cat(g) for g in gathered]
            out[1] = out[1].bool()
        return [t.cpu().numpy() for t in out] if to_host else out

    def _dev(self, manager, loaders):
        self.encode_news(manager, loaders["news"])
        res = self._predict(manager, loaders["dev"], with_labels=True)
        if res is None:
            return [], []
        impr, masks, preds, labels = res
        labels, preds = pack_results(impr, masks.astype(np.bool_), labels.astype(np.int32), preds.astype(np.float32))
        return labels, preds

    @torch.no_grad()
    def dev(self, manager, loaders, log=False):
        """same contract as BaseModel.dev; on a GPU the per-impression metrics are computed from the result
        tensors where they are (device_compute_metrics), the host regrouping + python loops only as fallback"""
        if self.device == "cpu" or getattr(manager, "host_metrics", False):
            return super().dev(manager, loaders, log)
        self.eval()
        self.encode_news(manager, loaders["news"])
        res = self._predict(manager, loaders["dev"], with_labels=True, to_host=False)
        metrics = None
        if self.rank == 0:
            impr, masks, preds, labels = res
            metrics = device_compute_metrics(impr, masks, preds, labels, manager.metrics)
            if metrics is None:
                l_, p_ = pack_results(impr.cpu().numpy(), masks.cpu().numpy().astype(np.bool_), labels.cpu().numpy().astype(np.int32),
                                      preds.cpu().numpy().astype(np.float32))
                metrics = compute_metrics(l_, p_, manager.metrics)
            metrics["main"] = metrics["auc"]
            self.logger.info(metrics)
            if log:
                manager._log(self.name, metrics)
        if manager.distributed:
            _barrier(self.device)
        return metrics

    def _test(self, manager, loaders):
        self.encode_news(manager, loaders["news"])
        res = self._predict(manager, loaders["test"], with_labels=False)
        if res is None:
            return []
        impr, masks, preds = res
        preds, = pack_results(impr, masks.astype(np.bool_), preds.astype(np.float32))
        return preds
