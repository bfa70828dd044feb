"""bf16 GEMM operand layouts (nr_gemm_bf16) at the BERT-base shapes: NT (both k-contiguous, what round 2 transposed its
operands into) against the reduction-major forms read with ds_read_b64_tr_b16.   usage: python tools/gemm_h_rm_bench.py [R]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import _lib
L = _lib.load()
R = int(sys.argv[1]) if len(sys.argv) > 1 else 14080
st = torch.cuda.current_stream().cuda_stream
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
def run(M, N, K, a_rm, b_rm, split):
    a = torch.randn(M, K, device="cuda").bfloat16(); b = torch.randn(N, K, device="cuda").bfloat16()
    A = a.t().contiguous() if a_rm else a; B = b.t().contiguous() if b_rm else b
    C = torch.empty(M, N, device="cuda")
    ws = torch.empty(L.nr_gemm_nt_bf16_ws_bytes(M, N, K, split) // 4 + 64, device="cuda")
    t = timeit(lambda: _lib.check(L.nr_gemm_bf16(A.data_ptr(), a_rm, M if a_rm else K, B.data_ptr(), b_rm, N if b_rm else K, None, M, N, K, 0,
                                                 C.data_ptr(), None, None, split, ws.data_ptr(), ws.numel() * 4, st), "gemm"))
    return t, 2.0 * M * N * K / t / 1e6
for name, M, N, K, split in [("dX out  ", R, 768, 768, 0), ("dX ffn2 ", R, 3072, 768, 0), ("dX ffn1 ", R, 768, 3072, 0), ("dX qkv  ", R, 768, 2304, 0),
                             ("dW ffn  ", 768, 3072, R, 1), ("dW ffn' ", 3072, 768, R, 1), ("dW qkv  ", 2304, 768, R, 1), ("dW out  ", 768, 768, R, 1)]:
    t0, f0 = run(M, N, K, 0, 0, split)
    if split:
        t1, f1 = run(M, N, K, 1, 1, split)
        print(f"{name} M={M} N={N} K={K}: NT {t0:7.1f} us {f0:6.1f} TF | both RM {t1:7.1f} us {f1:6.1f} TF", flush=True)
    else:
        t1, f1 = run(M, N, K, 0, 1, split)
        print(f"{name} M={M} N={N} K={K}: NT {t0:7.1f} us {f0:6.1f} TF | B RM    {t1:7.1f} us {f1:6.1f} TF", flush=True)
