"""what bounds gemm_x3_kernel on the NRMS shapes: the same launches on diagnostic builds of gemm.hip (-DNR_X3_EXP=1: no
operand split arithmetic, 2: one of the six MFMAs, 3: no epilogue, 5: one MFMA but every fragment read, 6: half of the
fragment reads, 7: no split / LDS stores after the first k-tile, 8: no global loads after the second k-tile), each build in its own process (the variant library is
swapped in by pointing newsrec_hip._lib.LIB_PATH at it before the first call; build it with
`make -C news-recommendation_amd/csrc variant NAME=e1 EXTRA=-DNR_X3_EXP=1`, which keeps gemm.hip's own flags).
usage: python tools/gemm_exp.py libnewsrec_hip_e1.so"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), sys.argv[1])
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
out = []
for name, K, N in [("qkv", 300, 900), ("outproj", 300, 300), ("ffn1", 300, 1200), ("ffn2", 1200, 300)]:
    x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    dy = torch.randn(R, N, device="cuda")
    out.append(f"{name} fwd {timeit(lambda: ops.linear_fwd(x, w, b)):6.1f} dX {timeit(lambda: ops.linear_bwd_input(dy, w)):6.1f} dW {timeit(lambda: ops.linear_bwd_weight(dy, x, False)):6.1f}")
print(f"{os.path.basename(_lib.LIB_PATH):24s} " + " | ".join(out))
