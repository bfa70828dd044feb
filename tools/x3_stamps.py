"""diagnostic (library built with -DNR_X3_EXP=4): where a workgroup of gemm_x3_kernel spends its cycles
usage: [X3_LIB=libnewsrec_hip_exp4.so] python tools/x3_stamps.py R K N [BK]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import numpy as np, torch
from newsrec_hip import _lib
if os.environ.get("X3_LIB"):
    _lib.LIB_PATH = os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ["X3_LIB"])
from newsrec_hip import ops
R, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(R, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
for _ in range(5):
    ops.linear_fwd(x, w, b)
torch.cuda.synchronize()
lib = ctypes.CDLL(_lib.LIB_PATH)
n = 8192 * 32
buf = (ctypes.c_ulonglong * n)()
assert lib.nr_debug_x3_stamps(buf, n) == 0
st = np.frombuffer(buf, dtype=np.uint64).reshape(8192, 32).astype(np.int64)
nwg = int((st[:, 0] != 0).sum())
st = st[:nwg]
BK = int(sys.argv[4]) if len(sys.argv) > 4 else 32          # 64: the small-tile kernel (gemm_x3s_kernel)
nkt = (K + BK - 1) // BK
t0 = st[:, 0]
pro = st[:, 2] - t0
nk = min(nkt, 6)
comp = np.stack([st[:, 3 + 4 * i] - (st[:, 2] if i == 0 else st[:, 2 + 4 * i]) for i in range(nk)], 1)
bar1 = np.stack([st[:, 4 + 4 * i] - st[:, 3 + 4 * i] for i in range(nk)], 1)
stor = np.stack([st[:, 5 + 4 * i] - st[:, 4 + 4 * i] for i in range(nk)], 1)
bar2 = np.stack([st[:, 6 + 4 * i] - st[:, 5 + 4 * i] for i in range(nk)], 1)
stage = bar1 + stor + bar2
print("  barrier 1 (wave 0 waits for the others)", [int(v) for v in np.median(bar1, 0)])
print("  wait for loads + split + LDS stores", [int(v) for v in np.median(stor, 0)])
print("  barrier 2", [int(v) for v in np.median(bar2, 0)])
loop_end = st[:, 28] - t0
end = st[:, 29] - t0
rt = (st[:, 30] - st[:, 1])           # 100 MHz ticks
print(f"workgroups {nwg}, k-tiles {nkt}; medians in shader cycles")
print("prologue (entry -> first tile in LDS)", int(np.median(pro)))
print("compute phase per k-tile (frag reads + 48 MFMAs)", [int(v) for v in np.median(comp, 0)])
print("staging phase per k-tile (barrier + split + LDS stores + barrier)", [int(v) for v in np.median(stage, 0)])
print("k-loop end", int(np.median(loop_end)), " epilogue", int(np.median(end - loop_end)), " total", int(np.median(end)))
print("clock GHz (cycles / realtime)", float(np.median(end / np.maximum(rt, 1))) * 0.1)
first = st[:, 1].min()
print("kernel span us (first entry -> last end, realtime)", (st[:, 30].max() - first) / 100.0)
start_rel = (st[:, 1] - first) / 100.0
print("workgroup start times us: deciles", [round(float(v), 1) for v in np.percentile(start_rel, [0, 10, 25, 50, 75, 90, 100])])
