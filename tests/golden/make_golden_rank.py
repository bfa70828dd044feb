"""Ranking fixture from the REAL reference (build container only): cached-inference scores of a two-tower model on
test-mode rows, regrouped by the reference's pack_results and ranked by its own expression
`ss.rankdata(1 - np.asarray(pred), method="min")` (models/BaseModel.py:141), plus the lines it writes to prediction.txt.

    python tests/golden/make_golden_rank.py         # writes tests/golden/rank_tfm_tfm.npz

The candidate sets are drawn so that any two DISTINCT scores of an impression differ by at least MIN_GAP_REL x the
largest |score| (stored as `min_gap`): a 1e-5-accurate implementation must then reproduce every ranking bit for bit.
Exact ties are kept on purpose (the same news twice in one impression): they rank by the tie rule, not by rounding.
"""
import os
import sys
import tempfile
import shutil

import numpy as np

np.bool8 = np.bool_
import scipy.stats as ss
import torch

REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
sys.path.insert(0, OUT)

from models.modules import encoder as R_enc                # noqa: E402
from models.TwoTower import TwoTowerModel                  # noqa: E402
from utils import util as R_util                           # noqa: E402
from make_golden import make_plm_dir, fake_manager, synth_news_table    # noqa: E402

MIN_GAP_REL = 1e-3
torch.set_num_threads(4)


def main():
    V, E, D, H, T, L, Ce, Nn = 499, 48, 48, 4, 16, 12, 10, 150
    tmp = tempfile.mkdtemp(prefix="golden_rank_")
    try:
        plm = make_plm_dir(tmp, V, E, H)
        mgr = fake_manager(plm, E, D, H, L, T, 4)
        torch.manual_seed(777)
        model = TwoTowerModel(mgr, R_enc.TfmNewsEncoder(mgr), R_enc.TfmUserEncoder(mgr))
        g = torch.Generator().manual_seed(778)
        with torch.no_grad():
            for n, p in model.named_parameters():
                if n.endswith("bias"):
                    p.copy_(0.1 * torch.randn(p.shape, generator=g))
                if "LayerNorm.weight" in n:
                    p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
                if n.endswith("embedding.weight"):
                    p.mul_(25.0)                            # spread the scores (the 0.02-std table gives near-ties everywhere)
        model.eval()
        ids, msk = synth_news_table(g, Nn, T, V)
        with torch.no_grad():
            cache = torch.cat([model.newsEncoder(ids[s:s + 7], msk[s:s + 7])[1] for s in range(0, Nn, 7)], 0)
            model.news_embeddings = cache
        rng = np.random.RandomState(9)
        n_impr = 40
        hists = [[int(v) for v in rng.randint(1, Nn, size=rng.choice([0, 2, 7, 12, 20]))][:L] for _ in range(n_impr)]

        def draw(imp):
            n = int(rng.randint(2, 3 * Ce))
            cand = [int(v) for v in rng.randint(1, Nn, size=n)]
            if imp % 4 == 0 and n >= 3:
                cand[min(n, Ce) - 1] = cand[0]              # the same news twice IN ONE ROW: an exact tie (across rows the
                                                            # reference re-encodes the user and the scores differ by an ulp)
            if imp % 9 == 0:
                cand[1] = 7                                 # a news without a title (all-PAD row of the cache)
            return cand
        cands = [draw(imp) for imp in range(n_impr)]
        for attempt in range(100):
            rows = [(imp, cands[imp][s:s + Ce], hists[imp]) for imp in range(n_impr) for s in range(0, len(cands[imp]), Ce)]
            R = len(rows)
            impr_index = np.asarray([r[0] for r in rows], dtype=np.int64)
            cdd_idx = np.zeros((R, Ce), dtype=np.int64)
            cdd_mask = np.zeros((R, Ce), dtype=np.bool_)
            his_idx = np.zeros((R, L), dtype=np.int64)
            his_mask = np.zeros((R, L), dtype=np.int64)
            for i, (_, cand, hist) in enumerate(rows):
                cdd_idx[i, :len(cand)] = cand
                cdd_mask[i, :len(cand)] = True
                his_idx[i, :len(hist)] = hist
                his_mask[i, :max(len(hist), 1)] = 1         # dataset.py:140-141: an empty history keeps slot 0
            # the reference's _test loop (BaseModel.py:292-329), batches of 7 rows
            impr_l, mask_l, pred_l = [], [], []
            with torch.no_grad():
                for s in range(0, R, 7):
                    x = {"cdd_idx": torch.from_numpy(cdd_idx[s:s + 7]), "his_idx": torch.from_numpy(his_idx[s:s + 7]),
                         "his_mask": torch.from_numpy(his_mask[s:s + 7])}
                    logits = model.infer(x)
                    mask_l.extend(torch.from_numpy(cdd_mask[s:s + 7]).tolist())
                    impr_l.extend(impr_index[s:s + 7].tolist())
                    pred_l.extend(logits.tolist())
            masks = np.asarray(mask_l, dtype=np.bool8)
            preds = np.asarray(pred_l, dtype=np.float32)
            grouped, = R_util.pack_results(impr_l, masks, preds)
            top = max(float(np.abs(np.asarray(p)).max()) for p in grouped)
            bad = []
            for imp, pred in enumerate(grouped):
                array = np.asarray(pred)
                d = np.abs(array[:, None] - array[None, :])
                d = d[d > 0]
                if d.size and d.min() < MIN_GAP_REL * top:
                    bad.append(imp)
            if not bad:
                break
            for imp in bad:                                 # redraw the impressions with a near-tie, keep the rest
                cands[imp] = draw(imp)
        else:
            raise RuntimeError("could not draw candidate sets with a safe gap")
        lines, ranks, gaps = [], [], []
        for index, pred in enumerate(grouped, 1):
            array = np.asarray(pred)
            rank_list = ss.rankdata(1 - array, method="min")               # BaseModel.py:141
            lines.append(str(index) + " [" + ",".join([str(i) for i in rank_list]) + "]")
            ranks.append(np.asarray(rank_list, dtype=np.int64))
            d = np.abs(array[:, None] - array[None, :])
            d = d[d > 0]
            if d.size:
                gaps.append(d.min())
        assert min(gaps) >= MIN_GAP_REL * top, (min(gaps), top)            # the fixture's promise
        blob = {"w." + k: v.detach().numpy() for k, v in model.state_dict().items()}
        blob.update(news_token_ids=ids.numpy(), news_attn_masks=msk.numpy(), impr_index=impr_index, cdd_idx=cdd_idx, cdd_mask=cdd_mask,
                    his_idx=his_idx, his_mask=his_mask, news_cache=cache.numpy(), logits=preds,
                    rank_sizes=np.asarray([len(r) for r in ranks]), ranks=np.concatenate(ranks), lines=np.asarray(lines),
                    min_gap=np.asarray(min(gaps)), max_abs=np.asarray(top), min_gap_rel=np.asarray(MIN_GAP_REL),
                    n_ties=np.asarray(sum(int(len(r) - len(set(r.tolist()))) for r in ranks)),
                    meta=np.asarray([V, E, D, H, T, L, 4, R, Nn], dtype=np.int64), Ce=np.asarray(Ce))
        path = os.path.join(OUT, "rank_tfm_tfm.npz")
        np.savez_compressed(path, **blob)
        print(f"rank fixture: {R} rows, {len(ranks)} impressions, min gap {min(gaps):.3e} vs max |score| {top:.3f}, "
              f"{int(blob['n_ties'])} tied ranks, {os.path.getsize(path) / 1024:.1f} KiB")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
