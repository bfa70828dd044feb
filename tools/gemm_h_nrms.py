"""the bf16-in-HBM DMA-staged GEMM (gemm_h_kernel) on NRMS-like shapes (K padded to 64): how fast is the structure that a
plane-based split-operand kernel would have, at short reductions?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gemm_h_bench as G
for (M, N, K) in [(19936, 896, 320), (19936, 320, 320), (19936, 1216, 320), (19936, 320, 1216), (19936, 896, 960), (19936, 1216, 960)]:
    for out in ("f32", "bf16"):
        G.run(M, N, K, check=False, out=out)
