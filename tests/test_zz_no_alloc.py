"""The entry point that allocates nothing (nr_add_layernorm_bwd_noalloc_f32) against the workspace one
(nr_add_layernorm_bwd_f32): bit-identical outputs, two calls in a row, the second on a smaller input."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(L, R, D, p, seed, noalloc):
    g = torch.Generator().manual_seed(R * 131 + D)
    dev = "cuda"
    dy = torch.randn(R, D, generator=g).to(dev)
    s = torch.randn(R, D, generator=g).to(dev)
    mean = s.mean(1).contiguous()
    rstd = (s.var(1, unbiased=False) + 1e-12).rsqrt().contiguous()
    gamma = torch.randn(D, generator=g).to(dev)
    ds = torch.full((R, D), float("nan"), device=dev)
    dh = torch.full((R, D), float("nan"), device=dev) if p > 0 else None
    dgamma = torch.full((D,), float("nan"), device=dev)
    dbeta = torch.full((D,), float("nan"), device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: None if t is None else t.data_ptr()
    if noalloc:
        rc = L.nr_add_layernorm_bwd_noalloc_f32(ptr(dy), ptr(s), ptr(mean), ptr(rstd), ptr(gamma), R, D, p, seed, 3, ptr(ds), ptr(dh),
                                                ptr(dgamma), ptr(dbeta), st)
    else:
        nbytes = L.nr_layernorm_bwd_ws_bytes(R, D)
        ws = torch.empty(nbytes // 4 + 64, device=dev)
        rc = L.nr_add_layernorm_bwd_f32(ptr(dy), ptr(s), ptr(mean), ptr(rstd), ptr(gamma), R, D, p, seed, 3, ptr(ds), ptr(dh),
                                        ptr(dgamma), ptr(dbeta), ptr(ws), ws.numel() * 4, st)
    assert rc == 0, L.nr_last_error()
    torch.cuda.synchronize()
    return ds, dh, dgamma, dbeta


def test_noalloc_layernorm_bwd_equals_workspace_entry_point():
    from newsrec_hip import _lib
    L = _lib.load()
    assert L.nr_noalloc_scratch_bytes() == (8 << 20) + (64 << 10)
    # first a large call (56 320 rows: 1024 blocks of partials), then a smaller one that reuses the dirty scratch
    for R, D, p in ((56320, 300, 0.1), (1600, 300, 0.0), (37, 768, 0.1), (5, 48, 0.0)):
        got = _run(L, R, D, p, 1234, True)
        want = _run(L, R, D, p, 1234, False)
        for a, b in zip(got, want):
            if b is None:
                assert a is None
                continue
            assert not torch.isnan(a).any()
            assert torch.equal(a, b), (R, D, p)


def test_noalloc_rejects_oversize():
    from newsrec_hip import _lib
    L = _lib.load()
    R, D = 56320, 2048
    assert L.nr_layernorm_bwd_ws_bytes(R, D) > L.nr_noalloc_scratch_bytes()
    t = torch.zeros(8, device="cuda")
    rc = L.nr_add_layernorm_bwd_noalloc_f32(t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), t.data_ptr(), R, D, 0.0, 0, 0,
                                            t.data_ptr(), None, t.data_ptr(), t.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 4 and b"scratch" in L.nr_last_error()      # NR_ERR_WORKSPACE, nothing launched
