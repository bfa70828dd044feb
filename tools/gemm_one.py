"""run ONE GEMM shape repeatedly (for rocprofv3 --pmc passes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops
kind, R, K, N, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
dy = torch.randn(R, N, device="cuda")
for _ in range(reps):
    if kind == "fwd": ops.linear_fwd(x, w, b)
    elif kind == "dx": ops.linear_bwd_input(dy, w)
    else: ops.linear_bwd_weight(dy, x, False)
torch.cuda.synchronize()
