"""where the cached-evaluation leg of bench.py spends its time: _predict (per-batch infer: wall and device time) against
the device metrics, for full and MIND-like histories.   usage: python tools/eval_breakdown.py"""
import os, sys, time, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from utils.util import device_compute_metrics

torch.cuda.set_device(0)
L, Ce, n_impr, Nn = bench.L, 20, 40000, bench.NEWS_DEV
model, mgr = bench.build_model(0, "nrms")
model = model.cuda().eval()
gd = torch.Generator().manual_seed(7)
model.news_embeddings = torch.randn(Nn, 300, generator=gd).cuda()
ncand = torch.randint(2, 81, (n_impr,), generator=gd)
nrow = (ncand + Ce - 1) // Ce
impr_of_row = torch.repeat_interleave(torch.arange(n_impr), nrow)
first = torch.cumsum(nrow, 0) - nrow
within = torch.arange(int(nrow.sum())) - first[impr_of_row]
valid = (within[:, None] * Ce + torch.arange(Ce)[None, :]) < ncand[impr_of_row][:, None]
Rtot = impr_of_row.numel()
s_i = torch.randint(0, Nn - 1, (n_impr,), generator=gd)[impr_of_row]
cdd = ((1 + (s_i[:, None] + 7919 * (within[:, None] * Ce + torch.arange(Ce)[None, :])) % (Nn - 1)) * valid).cuda()     # no duplicates (bench.py)
his_u = torch.randint(1, Nn, (n_impr, L), generator=gd)
hl = torch.randint(0, 71, (n_impr,), generator=gd).clamp(max=L)
live = torch.arange(L)[None, :] < hl[:, None]
hm_u = live.long(); hm_u[hl == 0, 0] = 1
lab = torch.where(valid, (torch.rand(Rtot, Ce, generator=gd) < 0.1).long(), torch.tensor(-1))
lab[first, 0], lab[first, 1] = 1, 0
lab, valid_d, impr_d = lab.cuda(), valid.cuda(), impr_of_row.cuda()
cases = {"full": (his_u[impr_of_row].cuda(), torch.ones(Rtot, L, dtype=torch.int64, device="cuda")),
         "mind": ((his_u * live)[impr_of_row].cuda(), hm_u[impr_of_row].cuda())}
mgr.impr_size, mgr.distributed = Ce, False
for name, (his, hm) in cases.items():
    loader = [{"cdd_idx": cdd[s:s + 8192], "his_idx": his[s:s + 8192], "his_mask": hm[s:s + 8192], "impr_index": impr_d[s:s + 8192],
               "cdd_mask": valid_d[s:s + 8192], "label": lab[s:s + 8192]} for s in range(0, Rtot, 8192)]
    with torch.no_grad():
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            res = model._predict(mgr, loader, with_labels=True, to_host=False)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            met = device_compute_metrics(*res, ["auc", "mean_mrr", "ndcg@5;10"])
            torch.cuda.synchronize(); t2 = time.perf_counter()
        ev = []
        for x in loader:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize(); h0 = time.perf_counter()
            e0.record(); model.infer(x); e1.record()
            torch.cuda.synchronize(); ev.append((e0.elapsed_time(e1), 1e3 * (time.perf_counter() - h0)))
    print(f"{name}: {len(loader)} batches of 8192 rows ({Rtot} rows); _predict {1e3 * (t1 - t0):.1f} ms, metrics {1e3 * (t2 - t1):.1f} ms; "
          f"infer alone per batch (device ms / wall ms): " + " ".join(f"{a:.1f}/{b:.1f}" for a, b in ev))
