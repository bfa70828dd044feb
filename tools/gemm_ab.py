"""A/B of a tuning switch (nr_set_tuning) on the GEMM launches of the NRMS train step, interleaved rounds in ONE process
(GPU box only).   usage: python tools/gemm_ab.py KEY [rows] [rounds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops

KEY = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R = int(sys.argv[2]) if len(sys.argv) > 2 else 19936
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 5
L = ops.lib()

def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

torch.manual_seed(0)
tot = {0: 0.0, 1: 0.0}
for name, K, N in [("qkv", 300, 900), ("outproj", 300, 300), ("ffn1", 300, 1200), ("ffn2", 1200, 300)]:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(R, N, device="cuda"); aux = torch.randn(R, K, device="cuda"); z = torch.empty(R, N, device="cuda")
    fns = {"fwd": (lambda: ops.linear_fwd(x, w, b, "gelu", z)) if name == "ffn1" else (lambda: ops.linear_fwd(x, w, b)),
           "dX": lambda: ops.linear_bwd_input(dy, w, 1, aux), "dW": lambda: ops.linear_bwd_weight(dy, x)}
    for tag, fn in fns.items():
        ts = {0: [], 1: []}
        for _ in range(ROUNDS):
            for v in (0, 1):
                L.nr_set_tuning(KEY, v)
                ts[v].append(timeit(fn))
        L.nr_set_tuning(KEY, 0)
        m0, m1 = sorted(ts[0])[ROUNDS // 2], sorted(ts[1])[ROUNDS // 2]
        mult = 2 if name == "outproj" and tag != "dW" else 1      # out-proj and pool-proj share the shape
        tot[0] += m0 * mult; tot[1] += m1 * mult
        fl = 2.0 * R * K * N
        print(f"{name:8s} {tag:3s} R={R} K={K:4d} N={N:4d} | off {m0:7.1f}us {fl/m0/1e6:6.1f}TF | on {m1:7.1f}us {fl/m1/1e6:6.1f}TF | {100*(m0-m1)/m0:+5.1f}%", flush=True)
print(f"sum over the step's big GEMM launches (approx): off {tot[0]:.0f} us, on {tot[1]:.0f} us ({100*(tot[0]-tot[1])/tot[0]:+.1f}%)")
