"""A/B of a tuning switch (nr_set_tuning KEY) on ONE transformer layer forward + backward at the headline news-tower
shape (623 titles x 32 = 19 936 rows, 300-d, 12 heads, dropout 0.1), interleaved rounds in one process.
usage: python tools/layer_ab.py KEY [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops
from models.modules.attention import TFMLayer

KEY = int(sys.argv[1]); ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 7
L = ops.lib()
torch.manual_seed(0)
N, S, D, H = 623, 32, 300, 12
layer = TFMLayer(D, H, 0.1).cuda().train()
x = torch.randn(N, S, D, device="cuda", requires_grad=True)
mask = torch.ones(N, S, dtype=torch.int64, device="cuda")
g = torch.randn(N, S, D, device="cuda")
ops.set_side_stream(False)
def run(n=10):
    ef, eb, ee = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    tf = tb = 0.0
    for _ in range(n):
        ef.record()
        y = layer(x, attention_mask=mask)
        eb.record()
        y.backward(g)
        ee.record()
        torch.cuda.synchronize()
        tf += ef.elapsed_time(eb); tb += eb.elapsed_time(ee)
    return tf / n * 1e3, tb / n * 1e3
res = {0: [], 1: []}
for v in (0, 1):
    L.nr_set_tuning(KEY, v); run(3)
for _ in range(ROUNDS):
    for v in (0, 1):
        L.nr_set_tuning(KEY, v)
        res[v].append(run())
L.nr_set_tuning(KEY, 0)
for v in (0, 1):
    f = sorted(r[0] for r in res[v])[ROUNDS // 2]; b = sorted(r[1] for r in res[v])[ROUNDS // 2]
    print(f"key {KEY} = {v}: fwd {f:.1f} us  bwd {b:.1f} us  total {f + b:.1f} us")
