"""accuracy (against an fp64 product) and rate of the GEMM entry points in the fp32 / fp32x3 modes (GPU box only)
usage: python tools/gemm_x3_check.py [rows]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops

def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

R = int(sys.argv[1]) if len(sys.argv) > 1 else 19936
torch.manual_seed(0)
for name, K, N in [("qkv", 300, 900), ("outproj", 300, 300), ("ffn1", 300, 1200), ("ffn2", 1200, 300), ("d768", 768, 768), ("user_ffn2", 1200, 300), ("big4096", 4096, 4096), ("bigk", 8192, 320)]:
    Rr = 1600 if name.startswith("user") else 4096 if name == "big4096" else R
    x = torch.randn(Rr, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(Rr, N, device="cuda")
    ref = {"fwd": (x.double() @ w.double().t() + b.double()), "dX": dy.double() @ w.double(), "dW": dy.double().t() @ x.double()}
    fns = {"fwd": lambda: ops.linear_fwd(x, w, b), "dX": lambda: ops.linear_bwd_input(dy, w), "dW": lambda: ops.linear_bwd_weight(dy, x)}
    line = f"{name:9s} R={Rr} K={K:4d} N={N:4d}"
    for tag in ("fwd", "dX", "dW"):
        fl = 2.0 * Rr * K * N
        for mode in ("fp32-mfma", "fp32"):
            ops.set_gemm_compute(mode)
            out = fns[tag]()
            out = out[0] if isinstance(out, tuple) else out
            err = ((out.double() - ref[tag]).abs().max() / ref[tag].abs().max()).item()
            ms = timeit(fns[tag])
            line += f" | {tag} {mode} err {err:.1e} {ms*1e3:6.1f}us {fl/ms/1e9:6.1f}TF"
    print(line, flush=True)
