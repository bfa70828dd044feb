"""diagnostic (library built with -DNR_ATTN_STAMP): where a workgroup of mhsa_fwd_mfma2_kernel spends its cycles
usage: python tools/attn_stamps.py N S H dh [p]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import numpy as np, torch
from newsrec_hip import ops, _lib
N, S, H, dh = (int(v) for v in sys.argv[1:5])
p = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
qkv = torch.randn(N * S, 3 * H * dh, device="cuda")
mask = torch.ones(N, S, dtype=torch.int64, device="cuda")
for _ in range(5):
    ops.mhsa(qkv, mask, N, S, H, dh, p, 1234, 3)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 4096 * 16
buf = (ctypes.c_ulonglong * n)()
assert lib.nr_debug_attn_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.int64)
nwg = int((st[:, 0] != 0).sum())
st = st[:nwg]
names = ["entry -> operands in LDS (loads, scatter, zero pad)", "barrier", "two score tiles (K.Q^T)", "softmax + dropout",
         "barrier", "P.V (2 x 16 MFMA per 32 features) + tile store", "barrier", "results out"]
idx = [0, 2, 3, 4, 5, 6, 7, 8, 9]
for k, nm in enumerate(names):
    d = st[:, idx[k + 1]] - st[:, idx[k]]
    print(f"{nm:60s} median {int(np.median(d)):7d}  max {int(d.max()):7d} cycles")
tot = st[:, 9] - st[:, 0]
rt = st[:, 10] - st[:, 1]
print("total cycles median", int(np.median(tot)), " clock GHz", float(np.median(tot / np.maximum(rt, 1))) * 0.1)
first = st[:, 1].min()
print("kernel span us (first entry -> last end)", (st[:, 10].max() - first) / 100.0)
print("workgroup start us: deciles", [round(float(v), 1) for v in np.percentile((st[:, 1] - first) / 100.0, [0, 10, 50, 90, 100])])
