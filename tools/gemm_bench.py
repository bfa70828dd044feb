"""time the f32-MFMA GEMM entry points on the shapes of the NRMS train step (GPU box only)"""
import os, sys, time
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

R = int(sys.argv[1]) if len(sys.argv) > 1 else 56320
shapes = [("qkv", 300, 900), ("outproj", 300, 300), ("ffn1", 300, 1200), ("ffn2", 1200, 300), ("d768", 768, 768), ("ffn1_768", 768, 3072)]
if len(sys.argv) > 2:                      # second argument: fp32 | bf16 (nr_set_gemm_compute); bf16 runs the BERT-base shapes
    ops.set_gemm_compute(sys.argv[2])
    if sys.argv[2] == "bf16":
        shapes = [("qkv_768", 768, 2304), ("d768", 768, 768), ("ffn1_768", 768, 3072), ("ffn2_768", 3072, 768), ("ffn1", 300, 1200)]
for name, K, N in shapes:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(R, N, device="cuda"); aux = torch.randn(R, K, device="cuda")
    z = torch.empty(R, N, device="cuda")
    fl = 2.0 * R * K * N
    res = []
    for tag, fn in (("fwd", lambda: ops.linear_fwd(x, w, b)), ("fwd+gelu", lambda: ops.linear_fwd(x, w, b, "gelu", z)),
                    ("dX", lambda: ops.linear_bwd_input(dy, w)), ("dX+aux", lambda: ops.linear_bwd_input(dy, w, 1, aux)),
                    ("dX*gelu'", lambda: ops.linear_bwd_input(dy, w, 2, aux)), ("dW", lambda: ops.linear_bwd_weight(dy, x)),
                    ("vendor_fwd", lambda: torch.addmm(b, x, w.t())), ("vendor_dW", lambda: dy.t() @ x)):
        ms = timeit(fn)
        res.append(f"{tag} {ms*1e3:7.1f}us {fl/ms/1e9:6.1f}TF")
    print(f"{name:9s} R={R} K={K:4d} N={N:4d} | " + " | ".join(res), flush=True)
