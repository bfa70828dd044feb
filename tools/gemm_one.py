"""run one GEMM shape a few times (for rocprofv3 --pmc passes): python tools/gemm_one.py MODE KIND R K N [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops
mode, kind, R, K, N = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
ops.set_gemm_compute(mode)
torch.manual_seed(0)
x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
dy = torch.randn(R, N, device="cuda")
fn = {"fwd": lambda: ops.linear_fwd(x, w, b), "dX": lambda: ops.linear_bwd_input(dy, w), "dW": lambda: ops.linear_bwd_weight(dy, x)}[kind]
for _ in range(reps):
    fn()
torch.cuda.synchronize()
