"""embedding scatter-add with and without two hot ids ([CLS] / [SEP] of every title): the hot rows serialise their atomics
(GPU box only: python tools/scatter_probe.py)"""
import os, sys
sys.path.insert(0, "/root/repo/news-recommendation_amd")
import torch
from newsrec_hip import ops

from newsrec_hip.ops import _p, _stream, check, lib
V, E, n = 30522, 300, 22560
def timeit(fn, k=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
g = torch.randn(n, E, device="cuda")
gt = torch.zeros(V, E, device="cuda")
for name, ids in (("uniform", torch.randint(1000, V, (n,))),
                  ("cls+sep every 20 rows", torch.where(torch.arange(n) % 20 == 0, torch.tensor(101), torch.where(torch.arange(n) % 20 == 19, torch.tensor(102), torch.randint(1000, V, (n,)))))):
    ids = ids.cuda()
    fn = lambda: check(lib().nr_scatter_add_rows_f32(_p(g), _p(ids), n, E, V, 0, _p(gt), _stream()), "scatter")
    print(name, round(timeit(fn), 1), "us")
