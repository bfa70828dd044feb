"""dW[300,300] (out-proj / pooling projection weight gradients of the news tower) under the env switches of gemm.hip"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops
R = 19936
torch.manual_seed(0)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for N, K in [(300, 300), (900, 300), (300, 1200), (1200, 300)]:
    x = torch.randn(R, K, device="cuda"); dy = torch.randn(R, N, device="cuda")
    t = timeit(lambda: ops.linear_bwd_weight(dy, x))
    print(f"dW[{N},{K}] R={R}: {t:.1f} us {2.0*R*N*K/t/1e6:.1f} TF  env: " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("NR_")), flush=True)
