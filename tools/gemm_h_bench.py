"""bf16-in-HBM GEMM (nr_gemm_nt_bf16): correctness against a torch reference on the same bf16 operands, and TFLOP/s at the
BERT-base shapes of the bf16 build.   python tools/gemm_h_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "news-recommendation_amd"))
import torch
from newsrec_hip import _lib
L = _lib.load()
st = lambda: torch.cuda.current_stream().cuda_stream
def run(M, N, K, act=0, split=0, reps=20, check=True, out="both"):
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    b = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda() if not split else None
    ah = torch.empty(M, K, dtype=torch.bfloat16, device="cuda"); bh = torch.empty(N, K, dtype=torch.bfloat16, device="cuda")
    assert L.nr_cast_bf16(a.data_ptr(), a.numel(), ah.data_ptr(), st()) == 0
    assert L.nr_cast_bf16(b.data_ptr(), b.numel(), bh.data_ptr(), st()) == 0
    assert torch.equal(ah, a.bfloat16()) and torch.equal(bh, b.bfloat16())
    c = torch.empty(M, N, device="cuda"); ch = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(L.nr_gemm_nt_bf16_ws_bytes(M, N, K, split) // 4 + 64, device="cuda")
    call = lambda: L.nr_gemm_nt_bf16(ah.data_ptr(), bh.data_ptr(), None if bias is None else bias.data_ptr(), M, N, K, act,
                                     None if out == "bf16" else c.data_ptr(), None if (split or out == "f32") else ch.data_ptr(), None, split,
                                     ws.data_ptr(), ws.numel() * 4, st())
    check = check and out == "both" or (check and split)
    rc = call(); assert rc == 0, L.nr_last_error()
    torch.cuda.synchronize()
    err = None
    if check:
        ref = ah.float() @ bh.float().t()
        if bias is not None: ref = ref + bias
        if act == 2: ref = torch.nn.functional.gelu(ref)
        err = ((c - ref).abs().max() / ref.abs().max()).item()
        if not split:
            assert (ch.float() - c.bfloat16().float()).abs().max().item() == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps * 1e-3
    print(f"M={M:6d} N={N:5d} K={K:6d} act={act} split={split} out={out}: {2.0*M*N*K/t/1e12:7.1f} TFLOP/s  {t*1e6:7.1f} us  max rel err {err}")
if __name__ == "__main__" and os.environ.get("NR_GEMM_H_KSPLIT"):
    R = 14080
    for (M, N, K) in [(768, 3072, R), (3072, 768, R), (768, 768, R), (2304, 768, R), (768, 3072, 4 * R), (768, 768, 4 * R)]:
        run(M, N, K, split=1, check=False)
    sys.exit(0)
if __name__ == "__main__":
    # transposes
    x = torch.randn(1000, 300, device="cuda"); xt = torch.empty(300, 1024, dtype=torch.bfloat16, device="cuda")
    assert L.nr_transpose_bf16(x.data_ptr(), 0, 1000, 300, 300, xt.data_ptr(), 1024, st()) == 0
    assert torch.equal(xt[:, :1000], x.t().bfloat16()) and (xt[:, 1000:] == 0).all()
    xh = x.bfloat16(); xt.fill_(1)
    assert L.nr_transpose_bf16(xh.data_ptr(), 1, 1000, 300, 300, xt.data_ptr(), 1024, st()) == 0
    assert torch.equal(xt[:, :1000], xh.t()) and (xt[:, 1000:] == 0).all()
    print("transpose ok")
    run(200, 100, 64); run(333, 257, 192, act=2); run(128, 128, 1024, split=1)
    R = 14080
    for (M, N, K) in [(R, 2304, 768), (R, 768, 768), (R, 3072, 768), (R, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192)]:
        run(M, N, K, check=M * N < 5e7)
    for (M, N, K) in [(R, 2304, 768), (R, 768, 768), (R, 3072, 768), (R, 768, 3072)]:
        run(M, N, K, out="bf16"); run(M, N, K, out="f32")
    for (M, N, K) in [(768, 3072, R), (3072, 768, R), (768, 768, R), (2304, 768, R)]:
        run(M, N, K, split=1, check=True)
