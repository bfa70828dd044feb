"""MIND datasets over the pre-tokenised caches (fixed-shape int64 batches: the hot path's input
contract).  Cache layout and per-item dict keys are the reference's (utils/dataset.py:14-293):

    {cache_root}/MIND/MIND{scale}_{split}/news/title_token_ids.pkl      list[list[int]]  (news i+1)
    {cache_root}/MIND/MIND{scale}_train/behaviors/behaviors.pkl          imprs, user_indices, histories, negatives
    {cache_root}/MIND/MIND{scale}_{dev,test}/behaviors/{impr_size}/behaviors.pkl   imprs, (labels), histories, user_indices

A missing cache is built on rank 0 from {data_root}/MIND/MIND{scale}_{split}/{news,behaviors}.tsv by
utils/cache.py (reference :47-53, :298-427); utils/synthetic.py writes MIND-shaped caches
for tests/bench.
"""
import logging
import os

import numpy as np
from torch.utils.data import Dataset

from utils.util import load_pickle, sample_news


class MIND(Dataset):
    def __init__(self, manager, data_dir, load_news=True, load_behaviors=True) -> None:
        super().__init__()
        self.logger = logging.getLogger(type(self).__name__)
        self.his_size, self.impr_size, self.negative_num = manager.his_size, manager.impr_size, manager.negative_num
        self.title_length, self.abs_length = manager.title_length, manager.abs_length
        name = os.path.basename(data_dir)
        root = os.path.join(manager.cache_root, "MIND", name)
        self.news_cache_root = os.path.join(root, "news")
        self.behaviors_cache_dir = os.path.join(root, "behaviors") if "train" in name else \
            os.path.join(root, "behaviors", str(self.impr_size))
        self.enable_fields = list(manager.enable_fields)
        # with device-side assembly a batch carries news indices only; the [Nn,T] token / mask
        # tables live in HBM and the rows are gathered there (bit-exact ids, no per-item fancy indexing, no
        # H->D copy of B x (C+L) x T ids).  --host-assembly (or -d cpu) keeps the reference's batch contract.
        self.ship_tokens = bool(getattr(manager, "host_assembly", True)) or getattr(manager, "device", "cpu") == "cpu"
        news_num = manager.news_nums[name] + 1               # row 0 = the all-PAD news
        if manager.rank == 0:
            from utils.cache import ensure_caches
            ensure_caches(manager, data_dir, self.news_cache_root, self.behaviors_cache_dir)
        if getattr(manager, "distributed", False):
            manager.barrier()
        if manager.rank == 0:
            self.logger.info(f"Loading Cache at {name}")
        if load_news:
            self._load_news(manager, news_num)
        if load_behaviors:
            path = os.path.join(self.behaviors_cache_dir, "behaviors.pkl")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} missing: build the behaviour cache first (utils/synthetic.py for synthetic data)")
            for k, v in load_pickle(path).items():
                setattr(self, k, v)

    def _load_news(self, manager, news_num):
        """token_ids / attn_masks [news_num, sequence_length] int64.  Per enabled field: truncate to the
        field length, force the last kept token to [SEP] (fields with <= 2 tokens stay empty), drop the
        leading [CLS] of every field after the first, concatenate, then right-pad with [PAD]."""
        pad, sep = manager.special_token_ids["[PAD]"], manager.special_token_ids["[SEP]"]
        self.sequence_length = S = manager.sequence_length
        ids = np.full((news_num, S), pad, dtype=np.int64)
        used = np.zeros(news_num, dtype=np.int64)
        start = 0
        for field, length in (("title", self.title_length), ("abs", self.abs_length)):
            if field not in self.enable_fields:
                continue
            path = os.path.join(self.news_cache_root, f"{field}_token_ids.pkl")
            if not os.path.exists(path):
                raise FileNotFoundError(f"{path} missing: build the news cache first (utils/synthetic.py for synthetic data)")
            for i, toks in enumerate(load_pickle(path), start=1):
                if i >= news_num:
                    break
                toks = list(toks[start:start + length])
                if len(toks) > 2 - start:
                    toks[-1] = sep
                    n = len(toks)
                    ids[i, used[i]:used[i] + n] = toks
                    used[i] += n
            if start == 0:
                start = 1
        self.token_ids = ids
        self.attn_masks = (np.arange(S)[None, :] < used[:, None]).astype(np.int64)

    def __len__(self):
        return len(self.imprs) if hasattr(self, "imprs") else len(self.token_ids)

    def _history(self, impr_index):
        his = list(self.histories[impr_index][:self.his_size])
        mask = np.zeros(self.his_size, dtype=np.int64)
        mask[:max(len(his), 1)] = 1                      # an empty history still "sees" one (PAD) slot
        his = np.asarray(his + [0] * (self.his_size - len(his)), dtype=np.int64)
        return his, mask

    def _pack(self, impr_index, cdd_idx, cdd_mask, **extra):
        his_idx, his_mask = self._history(impr_index)
        d = {
            "impr_index": impr_index, "user_index": self.user_indices[impr_index],
            "cdd_idx": cdd_idx, "his_idx": his_idx, "cdd_mask": cdd_mask, "his_mask": his_mask,
        }
        if self.ship_tokens:
            d.update({"cdd_token_id": self.token_ids[cdd_idx], "his_token_id": self.token_ids[his_idx],
                      "cdd_attn_mask": self.attn_masks[cdd_idx], "his_attn_mask": self.attn_masks[his_idx]})
        d.update(extra)
        return d


class MIND_Train(MIND):
    def __init__(self, manager) -> None:
        super().__init__(manager, os.path.join(manager.data_root, "MIND", f"MIND{manager.scale}_train"))

    def __getitem__(self, index):
        impr_index, positive = self.imprs[index]
        negatives, valid = sample_news(self.negatives[impr_index], self.negative_num)
        cdd_idx = np.asarray([positive] + negatives, dtype=np.int64)
        cdd_mask = np.zeros(len(cdd_idx), dtype=np.int64)
        cdd_mask[:1 + valid] = 1
        return self._pack(impr_index, cdd_idx, cdd_mask, label=0)      # the clicked news is candidate 0


class _MIND_Eval(MIND):
    def _candidates(self, index):
        impr_index, news = self.imprs[index]
        n = len(news)
        cdd_mask = np.zeros(self.impr_size, dtype=np.bool_)
        cdd_mask[:n] = True
        cdd_idx = np.asarray(list(news) + [0] * (self.impr_size - n), dtype=np.int64)
        return impr_index, n, cdd_idx, cdd_mask


class MIND_Dev(_MIND_Eval):
    def __init__(self, manager) -> None:
        super().__init__(manager, os.path.join(manager.data_root, "MIND", f"MIND{manager.scale}_dev"))

    def __getitem__(self, index):
        impr_index, n, cdd_idx, cdd_mask = self._candidates(index)
        label = np.asarray(list(self.labels[index]) + [-1] * (self.impr_size - n), dtype=np.int64)
        return self._pack(impr_index, cdd_idx, cdd_mask, label=label)


class MIND_Test(_MIND_Eval):
    def __init__(self, manager) -> None:
        super().__init__(manager, os.path.join(manager.data_root, "MIND", f"MIND{manager.scale}_test"))

    def __getitem__(self, index):
        impr_index, n, cdd_idx, cdd_mask = self._candidates(index)
        return self._pack(impr_index, cdd_idx, cdd_mask)


class MIND_News(MIND):
    """the corpus whose vectors are cached for evaluation: dev news (test news in test mode), row 0 included"""

    def __init__(self, manager) -> None:
        split = "test" if manager.mode == "test" else "dev"
        super().__init__(manager, os.path.join(manager.data_root, "MIND", f"MIND{manager.scale}_{split}"),
                         load_news=True, load_behaviors=False)

    def __getitem__(self, index):
        return {"cdd_idx": index, "cdd_token_id": self.token_ids[index], "cdd_attn_mask": self.attn_masks[index]}
