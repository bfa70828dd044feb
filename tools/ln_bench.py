"""LayerNorm forward / backward at the packed row count of the headline step: python tools/ln_bench.py [R D p]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import ops
R, D = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (22560, 300)
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
h = torch.randn(R, D, device="cuda", requires_grad=True); x = torch.randn(R, D, device="cuda", requires_grad=True)
g = torch.ones(D, device="cuda", requires_grad=True); b = torch.zeros(D, device="cuda", requires_grad=True)
dy = torch.randn(R, D, device="cuda")
for _ in range(30):
    y = ops.add_layernorm(h, x, g, b, 1e-12, p, 7, 1)
    y.backward(dy)
torch.cuda.synchronize()
