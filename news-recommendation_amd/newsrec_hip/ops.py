"""torch.autograd glue over the C ABI of libnewsrec_hip.so.

torch is plumbing here: device memory (the caching allocator), the current HIP
stream and autograd's graph.  Every number is produced by the HIP kernels; a
tensor that is not a contiguous fp32/int64 tensor on a ROCm device is an
error, not a cue to fall back (there is nothing to fall back to).
"""
import ctypes as C
import math
import os
import time

import torch

from . import _lib
from ._lib import NewsrecError, TfmWeights, TfmGrads, TfmSaved, TfmSavedBf16, check

PAD_ID = 0          # BERT word_embeddings.padding_idx (manager.special_token_ids["[PAD]"])
LN_EPS = 1e-12      # models/modules/attention.py:106,151 of the reference


def lib():
    return _lib.load()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """handle of the current stream of the current device (the raw query: torch.cuda.current_stream() builds a Stream
    object per call, and a train step asks sixty times)"""
    if _raw_stream is not None and _cur_device is not None:
        return _raw_stream(_cur_device())
    return torch.cuda.current_stream().cuda_stream


def _f32(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NewsrecError(f"{name}: the MI355X HIP path needs a tensor on a ROCm device "
                           f"(got {type(t).__name__} on {getattr(t, 'device', None)}); there is no CPU path in this package")
    if t.dtype != torch.float32:
        raise NewsrecError(f"{name}: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _i64(t, name="tensor"):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NewsrecError(f"{name}: the MI355X HIP path needs a tensor on a ROCm device (got {getattr(t, 'device', None)})")
    if t.dtype == torch.bool:
        t = t.long()
    if t.dtype != torch.int64:
        raise NewsrecError(f"{name}: expected int64, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _p(t):
    return None if t is None else t.data_ptr()


def _ws(nbytes, device):
    return torch.empty(max(int(nbytes), 16) // 4 + 4, dtype=torch.float32, device=device)


# ----------------------------------------------------------------------------- dropout stream
class DropoutStream:
    """hands out (seed, site) pairs; site increments per dropout site per call so masks
    differ between layers and steps while staying reproducible from torch.initial_seed()."""

    def __init__(self):
        self._site = 0

    def next(self, n):
        s = self._site
        self._site = (self._site + n) & 0x7FFFFFFF
        return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, s

    def reset(self, site=0):
        self._site = site


dropout_stream = DropoutStream()


# ----------------------------------------------------------------------------- gather / embedding
def gather_rows(table, ids):
    """rows of `table` [V,E] by int64 `ids` [...] -> [..., E] (bit-exact, no autograd)."""
    table = _f32(table, "table")
    ids = _i64(ids, "ids")
    out = torch.empty(*ids.shape, table.shape[1], dtype=torch.float32, device=table.device)
    check(lib().nr_gather_rows_f32(_p(table), table.shape[0], table.shape[1], _p(ids), ids.numel(), _p(out), None,
                                   _stream()), "nr_gather_rows_f32")
    return out


def check_indices(where=""):
    """raise NewsrecError if any row gather since the last call met an index outside its table (a token id beyond the
    vocabulary, a news index beyond the cache: wrong vocab.txt, a stale cache of another split).  The kernels clamp such
    an index instead of faulting and raise a per-GPU flag; this reads and clears it (one 4-byte copy + a stream sync), so
    call it where the host syncs anyway.  The reference's nn.Embedding / tensor indexing raise IndexError there."""
    import ctypes
    flag = ctypes.c_int(0)
    check(lib().nr_index_status(ctypes.byref(flag), _stream()), "nr_index_status")
    if flag.value:
        raise NewsrecError(f"index out of range in a row gather{' (' + where + ')' if where else ''}: a token id or news index lies outside "
                           "its table (the kernels read a clamped row instead of faulting; results since the last check are invalid)")


def gather_rows_i64(table, idx):
    """rows of an int64 [Nn,T] table (news token ids / masks resident in HBM) by news index [...] -> [...,T].
    The copy kernel moves 32-bit words, so the int64 table is viewed as [Nn, 2T] words: bit-exact."""
    table, idx = _i64(table, "token table"), _i64(idx, "news index")
    words = gather_rows(table.view(torch.float32), idx)
    return words.view(torch.int64)


# callbacks run when a backward reaches the token table's scatter-add: every other parameter gradient of the step is
# final by then (utils/dp.py starts the dense parameters' all-reduce here so that it overlaps the scatter-add)
table_backward_hooks = []


def _table_backward():
    for fn in tuple(table_backward_hooks):
        fn()


# token table data_ptr -> callable giving the [V, E] tensor its dense gradient should be scatter-added into, or None
# (utils/dp.py hands out the table's slice of its flat gradient buffer, once per step: no 36.6 MB copy into it afterwards)
table_grad_dest = {}


class _Embedding(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, ids, padding_idx):
        out = gather_rows(table, ids)
        ctx.save_for_backward(ids)
        ctx.shape = table.shape
        ctx.table_ptr = table.data_ptr()
        ctx.padding_idx = -1 if padding_idx is None else int(padding_idx)
        return out

    @staticmethod
    def backward(ctx, g):
        (ids,) = ctx.saved_tensors
        g = _f32(g, "grad")
        V, E = ctx.shape
        if V * E >= 1 << 20:                   # a token table (not a small positional one)
            _table_backward()
        dest = table_grad_dest.get(ctx.table_ptr)
        gt = dest() if dest is not None else None
        if gt is not None and gt.shape == (V, E) and gt.dtype == torch.float32 and gt.device == g.device and gt.is_contiguous():
            gt.zero_()
        else:
            gt = torch.zeros(V, E, dtype=torch.float32, device=g.device)
        check(lib().nr_scatter_add_rows_f32(_p(g), _p(ids), ids.numel(), E, V, ctx.padding_idx, _p(gt), _stream()),
              "nr_scatter_add_rows_f32")
        return gt, None, None


def embedding(table, ids, padding_idx=PAD_ID):
    """nn.Embedding(padding_idx) forward/backward (encoder.py:52,135 of the reference)."""
    return _Embedding.apply(_f32(table, "embedding.weight"), _i64(ids, "token_id"), padding_idx)


# ----------------------------------------------------------------------------- title tensors of a train step
def assemble_titles(cdd_ids, his_ids, cdd_mask, his_attn_mask, his_mask, keep_mode):
    """block-major title tensors of one train step in one launch (nr_assemble_titles): ids / mask [B*C + B*L, T] — every
    candidate title, then every history title — and keep [B*C + B*L] (keep_mode 0: all ones, 1: his_mask != 0, 2: slot
    index < history length).  Bit-exact copies; no autograd."""
    cdd_ids, his_ids = _i64(cdd_ids, "cdd_token_id"), _i64(his_ids, "his_token_id")
    cdd_mask, his_attn_mask, his_mask = _i64(cdd_mask, "cdd_attn_mask"), _i64(his_attn_mask, "his_attn_mask"), _i64(his_mask, "his_mask")
    B, Cn, T = cdd_ids.shape
    L = his_ids.shape[1]
    if his_ids.shape != (B, L, T) or cdd_mask.shape != cdd_ids.shape or his_attn_mask.shape != his_ids.shape or his_mask.shape != (B, L):
        raise NewsrecError("assemble_titles: inconsistent batch shapes")
    n = B * (Cn + L)
    ids = torch.empty(n, T, dtype=torch.int64, device=cdd_ids.device)
    msk = torch.empty(n, T, dtype=torch.int64, device=cdd_ids.device)
    keep = torch.empty(n, dtype=torch.int64, device=cdd_ids.device)
    check(lib().nr_assemble_titles(_p(cdd_ids), _p(his_ids), _p(cdd_mask), _p(his_attn_mask), _p(his_mask), int(keep_mode), B, Cn, L, T,
                                   _p(ids), _p(msk), _p(keep), _stream()), "nr_assemble_titles")
    return ids, msk, keep


# ----------------------------------------------------------------------------- packed title tokens
class PackPlan:
    """row layout of N titles with the padded positions removed (include/newsrec_hip.h, "packed title tokens"):
    offs [N+1] int32, row_map [rows] int32 (packed row -> n*T + j), rows = offs[N]."""
    __slots__ = ("offs", "row_map", "rows", "N", "T")

    def __init__(self, offs, row_map, rows, N, T):
        self.offs, self.row_map, self.rows, self.N, self.T = offs, row_map, rows, N, T


class PendingPlan:
    """a pack plan whose row count is still on its way to the host (pack_titles_begin): the plan kernels and an
    8-byte copy into pinned memory are queued, `event` marks their end."""
    __slots__ = ("offs", "info", "host", "event", "N", "T", "mask_ptr")

    def __init__(self, offs, info, host, event, N, T, mask_ptr):
        self.offs, self.info, self.host, self.event, self.N, self.T, self.mask_ptr = offs, info, host, event, N, T, mask_ptr


def _wait_event(event, spin_s=3e-3):
    """wait for a short device->host hand-off by polling: in a GPU-bound loop the host arrives here a little early every
    step, and a blocking hipEventSynchronize that goes to sleep wakes up late (measured: a step of 2.3 ms became 3.1 ms
    whenever the host was 0.3 ms ahead of the GPU, tools/host_slack.py).  Long waits fall back to the blocking call."""
    if event.query():
        return
    t0 = time.perf_counter()
    while not event.query():
        if time.perf_counter() - t0 > spin_s:
            event.synchronize()
            return


def pack_titles_begin(mask, title_keep=None):
    """queue the plan of pack_titles for `mask` / `title_keep` without waiting for it.  A caller that knows its next
    batch (a training loop with one batch of look-ahead) starts the plan a step early, so that the row count is already
    on the host when that batch's forward needs it and the step has no device->host wait in it."""
    mask = _i64(mask, "attn_mask")
    N, T = mask.shape
    if title_keep is not None:
        title_keep = _i64(title_keep, "title_keep").reshape(-1)
        if title_keep.numel() != N:
            raise NewsrecError(f"pack_titles: title_keep has {title_keep.numel()} entries for {N} titles")
    offs = torch.empty(N + 1, dtype=torch.int32, device=mask.device)
    info = torch.empty(2, dtype=torch.int32, device=mask.device)
    check(lib().nr_pack_plan(_p(mask), _p(title_keep), N, T, _p(offs), _p(info), _stream()), "nr_pack_plan")
    host = torch.empty(2, dtype=torch.int32, pin_memory=True)
    host.copy_(info, non_blocking=True)
    event = torch.cuda.Event()
    event.record()
    return PendingPlan(offs, info, host, event, N, T, mask.data_ptr())


def pack_titles(ids, mask, title_keep=None, pending=None):
    """ids / mask [N,T] int64 on the device -> (ids_packed [rows] int64, PackPlan), or None when a mask row is not
    prefix-shaped (the packing is exact only for those; the caller then takes the padded path).  title_keep [N] 0/1:
    titles with 0 get no rows at all (the caller guarantees nothing downstream uses their vectors).  One 8-byte
    device->host read (the row count sizes every later allocation); `pending` = the PendingPlan of an earlier
    pack_titles_begin on the same mask / title_keep, whose read has usually finished long ago."""
    ids, mask = _i64(ids, "token_id"), _i64(mask, "attn_mask")
    N, T = mask.shape
    if pending is None:
        pending = pack_titles_begin(mask, title_keep)
    elif (pending.N, pending.T, pending.mask_ptr) != (N, T, mask.data_ptr()):
        raise NewsrecError("pack_titles: the pending plan was started for another mask")
    _wait_event(pending.event)
    rows, ok = pending.host.tolist()
    if not ok:
        return None
    offs = pending.offs
    ids_packed = torch.empty(rows, dtype=torch.int64, device=mask.device)
    row_map = torch.empty(rows, dtype=torch.int32, device=mask.device)
    check(lib().nr_pack_ids(_p(ids), _p(offs), N, T, _p(ids_packed), _p(row_map), _stream()), "nr_pack_ids")
    return ids_packed, PackPlan(offs, row_map, int(rows), N, T)


# ----------------------------------------------------------------------------- linear
ACT = {"none": 0, "relu": 1, "gelu": 2, "tanh": 3}


def linear_fwd(x2d, w, b, act="none", z=None):
    R, K = x2d.shape
    N = w.shape[0]
    y = torch.empty(R, N, dtype=torch.float32, device=x2d.device)
    check(lib().nr_linear_fwd_f32(_p(x2d), _p(w), _p(b), R, K, N, ACT[act], _p(y), _p(z), _stream()), "nr_linear_fwd_f32")
    return y


def linear_bwd_input(dy, w, mode=0, aux=None, aux2=None):
    R, N = dy.shape
    K = w.shape[1]
    dx = torch.empty(R, K, dtype=torch.float32, device=dy.device)
    check(lib().nr_linear_bwd_input_f32(_p(dy), _p(w), R, N, K, mode, _p(aux), _p(aux2), _p(dx), _stream()),
          "nr_linear_bwd_input_f32")
    return dx


def linear_bwd_weight(dy, x2d, want_bias=True, stream=None, keep=None):
    """stream: launch there instead of on the current stream.  keep: a list that receives the workspace tensor — a caller
    that launches on another stream holds it until that stream has been joined (the caching allocator would otherwise
    hand the block to the next allocation of the current stream at once)."""
    R, N = dy.shape
    K = x2d.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=dy.device)
    db = torch.empty(N, dtype=torch.float32, device=dy.device) if want_bias else None
    nbytes = lib().nr_linear_bwd_weight_ws_bytes(R, N, K)
    ws = _ws(nbytes, dy.device)
    check(lib().nr_linear_bwd_weight_f32(_p(dy), _p(x2d), R, N, K, _p(dw), _p(db), _p(ws), ws.numel() * 4,
                                         _stream() if stream is None else stream.cuda_stream), "nr_linear_bwd_weight_f32")
    if keep is not None:
        keep.append(ws)
    return dw, db


# A layer's weight gradient feeds nothing else in its backward: with enough rows it runs on a second stream beside the
# input-gradient product (the library does the same inside nr_tfm_layer_bwd_f32).  Every tensor is allocated on the
# caller's stream and the caller's stream waits for the side stream before the backward returns, so the caching
# allocator never sees the second stream.  NR_SIDE_STREAM=0 turns it off.
_SIDE = {}
_SIDE_MIN_ROWS = int(os.environ.get("NR_SIDE_MIN_ROWS", "1024"))


_SIDE_ON = os.environ.get("NR_SIDE_STREAM") != "0"


def set_side_stream(on):
    """weight-gradient products on a second stream (here and inside the library's layer backward): on by default; off
    runs every kernel on the current stream, one at a time (what a per-kernel timing wants).  Returns the old setting."""
    global _SIDE_ON
    old, _SIDE_ON = _SIDE_ON, bool(on)
    lib().nr_set_side_stream(1 if on else 0)
    return old


def _side_stream(device, rows):
    if rows < _SIDE_MIN_ROWS or not _SIDE_ON or torch.device(device).type != "cuda":
        return None
    key = torch.device(device).index
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


def linear_bwd_both(dy, w, x2d, want_input, want_bias=True, mode=0, aux=None, aux2=None):
    """(dx or None, dw, db): dW = dy^T x on the side stream while dX = dy W runs on the current one"""
    side = _side_stream(dy.device, dy.shape[0]) if want_input else None
    if side is None:
        dx = linear_bwd_input(dy, w, mode, aux, aux2) if want_input else None
        dw, db = linear_bwd_weight(dy, x2d, want_bias)
        return dx, dw, db
    main = torch.cuda.current_stream(dy.device)
    side.wait_stream(main)
    held = []
    dw, db = linear_bwd_weight(dy, x2d, want_bias, stream=side, keep=held)
    dx = linear_bwd_input(dy, w, mode, aux, aux2)
    main.wait_stream(side)
    del held                                    # the workspace may be reused only now
    return dx, dw, db


class _Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        y = linear_fwd(x2, w, b)
        ctx.save_for_backward(x2, w)
        ctx.has_bias = b is not None
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, w = ctx.saved_tensors
        g2 = _f32(g, "grad").reshape(-1, w.shape[0])
        dx, dw, db = linear_bwd_both(g2, w, x2, ctx.needs_input_grad[0], ctx.has_bias)
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db


def linear(x, w, b=None):
    return _Linear.apply(_f32(x, "x"), _f32(w, "weight"), None if b is None else _f32(b, "bias"))


class _LinearGelu(torch.autograd.Function):
    """gelu(x W^T + b) with the exact erf GELU (TFMIntermediate, attention.py:135-144)"""

    @staticmethod
    def forward(ctx, x, w, b):
        x2 = x.reshape(-1, x.shape[-1])
        z = torch.empty(x2.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
        y = linear_fwd(x2, w, b, "gelu", z)
        ctx.save_for_backward(x2, w, z)
        ctx.xshape = x.shape
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, g):
        x2, w, z = ctx.saved_tensors
        g2 = _f32(g, "grad").reshape(-1, w.shape[0])
        dz = torch.empty_like(g2)
        check(lib().nr_gelu_bwd_f32(_p(g2), _p(z), g2.numel(), _p(dz), _stream()), "nr_gelu_bwd_f32")
        dx, dw, db = linear_bwd_both(dz, w, x2, ctx.needs_input_grad[0], True)
        return (dx.view(ctx.xshape) if dx is not None else None), dw, db


def linear_gelu(x, w, b):
    return _LinearGelu.apply(_f32(x, "x"), _f32(w, "weight"), _f32(b, "bias"))


class _AddLayerNorm(torch.autograd.Function):
    """LayerNorm(dropout(h) + x) (TFMSelfOutput / TFMOutput, attention.py:109-113,154-158)"""

    @staticmethod
    def forward(ctx, h, x, gamma, beta, eps, p, seed, site):
        D = h.shape[-1]
        h2, x2 = h.reshape(-1, D), x.reshape(-1, D)
        R = h2.shape[0]
        y = torch.empty_like(h2)
        s = torch.empty_like(h2)
        stats = torch.empty(2, R, dtype=torch.float32, device=h.device)
        check(lib().nr_add_layernorm_fwd_f32(_p(h2), _p(x2), 0, _p(gamma), _p(beta), R, D, eps, p, seed, site, _p(y), _p(s),
                                             _p(stats[0]), _p(stats[1]), _stream()), "nr_add_layernorm_fwd_f32")
        ctx.save_for_backward(s, stats, gamma)
        ctx.cfg = (R, D, p, seed, site, h.shape)
        return y.view(h.shape)

    @staticmethod
    def backward(ctx, g):
        s, stats, gamma = ctx.saved_tensors
        R, D, p, seed, site, shape = ctx.cfg
        g = _f32(g, "grad").reshape(R, D)
        L = lib()
        ds = torch.empty(R, D, dtype=torch.float32, device=g.device)
        dh = torch.empty(R, D, dtype=torch.float32, device=g.device) if p > 0 else None
        dgamma = torch.empty(D, dtype=torch.float32, device=g.device)
        dbeta = torch.empty(D, dtype=torch.float32, device=g.device)
        ws = _ws(L.nr_layernorm_bwd_ws_bytes(R, D), g.device)
        check(L.nr_add_layernorm_bwd_f32(_p(g), _p(s), _p(stats[0]), _p(stats[1]), _p(gamma), R, D, p, seed, site, _p(ds), _p(dh),
                                         _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 4, _stream()), "nr_add_layernorm_bwd_f32")
        return (dh if dh is not None else ds).view(shape), ds.view(shape), dgamma, dbeta, None, None, None, None


def add_layernorm(h, x, gamma, beta, eps=LN_EPS, p_drop=0.0, seed=0, site=0):
    h, x = _f32(h, "hidden_states"), _f32(x, "input_tensor")
    if h.shape != x.shape:
        raise NewsrecError(f"add_layernorm: {tuple(h.shape)} vs {tuple(x.shape)}")
    return _AddLayerNorm.apply(h, x, _f32(gamma, "LayerNorm.weight"), _f32(beta, "LayerNorm.bias"), float(eps), float(p_drop),
                               int(seed), int(site))


class _MHSA(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh) + mask) v per head (TFMSelfAttention core, attention.py:87-97) over qkv [N*S, 3*H*dh]"""

    @staticmethod
    def forward(ctx, qkv, mask, N, S, H, dh, p, seed, site, mask_mode):
        A = H * dh
        out = torch.empty(N * S, A, dtype=torch.float32, device=qkv.device)
        rmax = torch.empty(N, H, S, dtype=torch.float32, device=qkv.device)
        rsum = torch.empty(N, H, S, dtype=torch.float32, device=qkv.device)
        check(lib().nr_mhsa_fwd_f32(_p(qkv), _p(mask), N, S, H, dh, p, seed, site, mask_mode, _p(out), _p(rmax), _p(rsum), _stream()),
              "nr_mhsa_fwd_f32")
        ctx.save_for_backward(qkv, mask, out, rmax, rsum)
        ctx.cfg = (N, S, H, dh, p, seed, site, mask_mode)
        return out

    @staticmethod
    def backward(ctx, g):
        qkv, mask, out, rmax, rsum = ctx.saved_tensors
        N, S, H, dh, p, seed, site, mask_mode = ctx.cfg
        g = _f32(g, "grad")
        dqkv = torch.empty_like(qkv)
        check(lib().nr_mhsa_bwd_f32(_p(qkv), _p(mask), _p(out), _p(g), _p(rmax), _p(rsum), N, S, H, dh, p, seed, site, mask_mode,
                                    _p(dqkv), _stream()), "nr_mhsa_bwd_f32")
        return (dqkv,) + (None,) * 9


def mhsa(qkv, mask, N, S, H, dh, p_drop=0.0, seed=0, site=0, hf_mask=False):
    """qkv [N*S, 3*H*dh] (q | k | v per row), mask [N,S] 0/1 -> context [N*S, H*dh]"""
    return _MHSA.apply(_f32(qkv, "qkv"), _i64(mask, "attention_mask"), int(N), int(S), int(H), int(dh), float(p_drop), int(seed),
                       int(site), 1 if hf_mask else 0)


# ----------------------------------------------------------------------------- conv1d(k=3, pad=1) + relu
class _ConvRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        # x [M, T, E]; w [D, E, 3]; b [D]
        M, T, E = x.shape
        D = w.shape[0]
        wpack = torch.empty(2 * D * 3 * E, dtype=torch.float32, device=x.device)
        L = lib()
        check(L.nr_conv1d_k3_pack(_p(w), D, E, _p(wpack), _stream()), "nr_conv1d_k3_pack")
        y = torch.empty(M, T, D, dtype=torch.float32, device=x.device)
        check(L.nr_conv1d_k3_relu_fwd_f32(_p(x), _p(wpack), _p(b), M, T, E, D, _p(y), _stream()),
              "nr_conv1d_k3_relu_fwd_f32")
        ctx.save_for_backward(x, wpack, y)
        ctx.dims = (M, T, E, D)
        return y

    @staticmethod
    def backward(ctx, g):
        x, wpack, y = ctx.saved_tensors
        M, T, E, D = ctx.dims
        g = _f32(g, "grad")
        L = lib()
        gm = torch.empty_like(g)
        check(L.nr_relu_bwd_f32(_p(g), _p(y), g.numel(), _p(gm), _stream()), "nr_relu_bwd_f32")
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty(D, E, 3, dtype=torch.float32, device=g.device)
        db = torch.empty(D, dtype=torch.float32, device=g.device)
        nbytes = L.nr_conv1d_k3_bwd_ws_bytes(M, T, E, D)
        ws = _ws(nbytes, g.device)
        check(L.nr_conv1d_k3_bwd_f32(_p(gm), _p(x), _p(wpack), M, T, E, D, _p(dx), _p(dw), _p(db), _p(ws), ws.numel() * 4,
                                     _stream()), "nr_conv1d_k3_bwd_f32")
        return dx, dw, db


def conv1d_k3_relu(x, w, b):
    """relu(Conv1d(E->D, 3, padding=1)(x^T))^T on token-major x [M,T,E] (encoder.py:55-56)."""
    return _ConvRelu.apply(_f32(x, "x"), _f32(w, "cnn.weight"), _f32(b, "cnn.bias"))


# ----------------------------------------------------------------------------- transformer layer
_TFM_ORDER = ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b", "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")


def _carve(device, sizes):
    """one allocation, 256-B aligned pieces (floats): (buffer, device address of every piece; 0-sized pieces -> None).
    Addresses, not views: a dozen tensor slices per layer call cost the host more than the layer's C call."""
    offs, tot = [], 0
    for s in sizes:
        offs.append(tot)
        tot += (int(s) + 63) // 64 * 64
    buf = torch.empty(max(tot, 64), dtype=torch.float32, device=device)
    base = buf.data_ptr()
    return buf, [base + 4 * o if s else None for o, s in zip(offs, sizes)]


class _TFMLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask, H, p, seed, site, mask_mode, plan, grad_mode, *params):
        N, S = mask.shape
        D = x.shape[-1]
        R = N * S if plan is None else plan.rows      # packed: x is [R, D]
        # grad_mode = torch.is_grad_enabled() at the call (inside a Function's forward it always reads False, and
        # needs_input_grad ignores no_grad: parameters "need" gradients in evaluation too): no backward buffers under no_grad
        train = grad_mode and any(ctx.needs_input_grad)
        if train and plan is not None and S > 32:
            raise NewsrecError(f"tfm_layer: packed sequences of {S} > 32 positions are forward-only (no packed backward)")
        sizes = [3 * D * D, 3 * D, R * 3 * D, R * D, R * D, R * D, R * 4 * D if train else 0, R * 4 * D, R * D,
                 4 * R if train else 0, N * H * S if train else 0, N * H * S if train else 0]
        buf, ptrs = _carve(x.device, sizes)           # wqkv bqkv qkv ctx s1 y1 z a s2 stats rmax rsum
        y = torch.empty_like(x)
        W = TfmWeights(*[_p(t) for t in params])
        SV = TfmSaved(*ptrs)
        if plan is None:
            check(lib().nr_tfm_layer_fwd_f32(_p(x), _p(mask), N, S, D, H, C.byref(W), p, seed, site, mask_mode, C.byref(SV),
                                            _p(y), _stream()), "nr_tfm_layer_fwd_f32")
        else:
            check(lib().nr_tfm_layer_packed_fwd_f32(_p(x), _p(mask), _p(plan.offs), _p(plan.row_map), N, S, R, D, H, C.byref(W), p, seed,
                                                   site, mask_mode, C.byref(SV), _p(y), _stream()), "nr_tfm_layer_packed_fwd_f32")
        if train:
            ctx.save_for_backward(x, mask, buf, *params)
            ctx.cfg = (N, S, D, H, p, seed, site, mask_mode, sizes, plan)
        return y

    @staticmethod
    def backward(ctx, g):
        x, mask, buf, *params = ctx.saved_tensors
        N, S, D, H, p, seed, site, mask_mode, sizes, plan = ctx.cfg
        R = N * S if plan is None else plan.rows
        g = _f32(g, "grad")
        offs, tot = [], 0
        for s in sizes:
            offs.append(tot)
            tot += (int(s) + 63) // 64 * 64
        base = buf.data_ptr()
        ptrs = [base + 4 * o if s else None for o, s in zip(offs, sizes)]
        W = TfmWeights(*[_p(t) for t in params])
        SV = TfmSaved(*ptrs)
        # packed q|k|v gradient: three contiguous views, handed out as the three parameters' grads
        gqkv_w = torch.empty(3, D, D, dtype=torch.float32, device=g.device)
        gqkv_b = torch.empty(3, D, dtype=torch.float32, device=g.device)
        grads = {n: torch.empty_like(t) for n, t in zip(_TFM_ORDER, params) if n not in ("wq", "wk", "wv", "bq", "bk", "bv")}
        grads.update(wq=gqkv_w[0], wk=gqkv_w[1], wv=gqkv_w[2], bq=gqkv_b[0], bk=gqkv_b[1], bv=gqkv_b[2])
        G = TfmGrads(*[_p(grads[n]) for n in _TFM_ORDER])
        dx = torch.empty_like(x)
        L = lib()
        nbytes = L.nr_tfm_layer_bwd_ws_bytes(R, 1, D)
        ws = _ws(nbytes, g.device)
        if plan is None:
            check(L.nr_tfm_layer_bwd_f32(_p(g), _p(x), _p(mask), N, S, D, H, C.byref(W), p, seed, site, mask_mode, C.byref(SV),
                                         C.byref(G), _p(dx), _p(ws), ws.numel() * 4, _stream()), "nr_tfm_layer_bwd_f32")
        else:
            check(L.nr_tfm_layer_packed_bwd_f32(_p(g), _p(x), _p(mask), _p(plan.offs), _p(plan.row_map), N, S, R, D, H, C.byref(W), p,
                                                seed, site, mask_mode, C.byref(SV), C.byref(G), _p(dx), _p(ws), ws.numel() * 4,
                                                _stream()), "nr_tfm_layer_packed_bwd_f32")
        return (dx, None, None, None, None, None, None, None, None) + tuple(grads[n] for n in _TFM_ORDER)


class _TFMLayerBf16(torch.autograd.Function):
    """the same layer in the bf16 build (nr_tfm_layer_{fwd,bwd}_bf16): GEMM operands and attention tensors are bf16 in HBM,
    residual stream / LayerNorm / softmax / parameter gradients f32"""

    @staticmethod
    def forward(ctx, x, mask, H, p, seed, site, mask_mode, plan, grad_mode, *params):
        N, S = mask.shape
        D = x.shape[-1]
        R = N * S if plan is None else plan.rows
        train = grad_mode and any(ctx.needs_input_grad)          # (see _TFMLayer.forward)
        dev = x.device
        h = lambda *shape: torch.empty(*shape, dtype=torch.bfloat16, device=dev)
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        bufs = dict(x_h=h(R, D), qkv_h=h(R, 3 * D), ctx_h=h(R, D), y1_h=h(R, D), z_h=h(R, 4 * D), a_h=h(R, 4 * D), s1=f(R, D), y1=f(R, D),
                    s2=f(R, D), stats=f(4, R), row_max=f(N, H, S), row_sum=f(N, H, S), w_h=h(12 * D * D))
        if train:
            bufs["wT_h"] = h(12 * D * D)                    # transposed bf16 weights: written with w_h, read by the backward's dX products
        y = torch.empty_like(x)
        W = TfmWeights(*[_p(t) for t in params])
        SV = TfmSavedBf16(*[_p(bufs.get(n)) for n, _ in TfmSavedBf16._fields_])
        offs, row_map = (None, None) if plan is None else (plan.offs, plan.row_map)
        check(lib().nr_tfm_layer_fwd_bf16(_p(x), _p(mask), _p(offs), _p(row_map), N, S, R, D, H, C.byref(W), p, seed, site, mask_mode,
                                          C.byref(SV), _p(y), _stream()), "nr_tfm_layer_fwd_bf16")
        if train:
            bufs.pop("y1"), bufs.pop("w_h")                 # forward-only scratch
            ctx.names = list(bufs)
            ctx.save_for_backward(mask, *bufs.values(), *params)
            ctx.cfg = (N, S, D, H, p, seed, site, mask_mode, plan, tuple(x.shape))
        return y

    @staticmethod
    def backward(ctx, g):
        N, S, D, H, p, seed, site, mask_mode, plan, xshape = ctx.cfg
        R = N * S if plan is None else plan.rows
        offs, row_map = (None, None) if plan is None else (plan.offs, plan.row_map)
        saved = ctx.saved_tensors
        mask, rest = saved[0], saved[1:]
        bufs = dict(zip(ctx.names, rest[:len(ctx.names)]))
        params = rest[len(ctx.names):]
        g = _f32(g, "grad")
        W = TfmWeights(*[_p(t) for t in params])
        SV = TfmSavedBf16(*[_p(bufs.get(n)) for n, _ in TfmSavedBf16._fields_])
        gqkv_w = torch.empty(3, D, D, dtype=torch.float32, device=g.device)
        gqkv_b = torch.empty(3, D, dtype=torch.float32, device=g.device)
        grads = {n: torch.empty_like(t) for n, t in zip(_TFM_ORDER, params) if n not in ("wq", "wk", "wv", "bq", "bk", "bv")}
        grads.update(wq=gqkv_w[0], wk=gqkv_w[1], wv=gqkv_w[2], bq=gqkv_b[0], bk=gqkv_b[1], bv=gqkv_b[2])
        G = TfmGrads(*[_p(grads[n]) for n in _TFM_ORDER])
        dx = torch.empty(xshape, dtype=torch.float32, device=g.device)
        L = lib()
        ws = _ws(L.nr_tfm_layer_bwd_bf16_ws_bytes(R, D), g.device)
        check(L.nr_tfm_layer_bwd_bf16(_p(g), _p(mask), _p(offs), _p(row_map), N, S, R, D, H, C.byref(W), p, seed, site, mask_mode, C.byref(SV),
                                      C.byref(G), _p(dx), _p(ws), ws.numel() * 4, _stream()), "nr_tfm_layer_bwd_bf16")
        return (dx, None, None, None, None, None, None, None, None) + tuple(grads[n] for n in _TFM_ORDER)


def _contiguous3(a, b, c):
    n = a.numel() * a.element_size()
    return b.data_ptr() == a.data_ptr() + n and c.data_ptr() == b.data_ptr() + n


def tfm_layer_bf16_supported(S, D, H):
    return D % 64 == 0 and D % H == 0 and S <= 32 and D // H <= 64


def tfm_layer(x, mask, head_num, params, p_drop=0.0, seed=0, site=0, hf_mask=False, plan=None):
    """TFMLayer.forward (attention.py:173-182).  x [N,S,D] fp32, mask [N,S] int64 (1 = real),
    params: dict with keys wq,bq,wk,bk,wv,bv,wo,bo,ln1_g,ln1_b,w1,b1,w2,b2,ln2_g,ln2_b.
    hf_mask: key-mask semantics of HF BertModel under SDPA (masked keys excluded; no valid key -> zero
    context) instead of the reference layer's additive -1e5.
    plan: a PackPlan from pack_titles -> x is the packed [rows, D] matrix and so is the result."""
    x = _f32(x, "hidden_states")
    mask = _i64(mask, "attention_mask")
    if plan is not None:
        if x.dim() != 2 or x.shape[0] != plan.rows or tuple(mask.shape) != (plan.N, plan.T):
            raise NewsrecError(f"tfm_layer: packed x {tuple(x.shape)} / mask {tuple(mask.shape)} do not match the plan "
                               f"({plan.rows} rows, {plan.N} x {plan.T})")
    elif x.dim() != 3 or mask.shape != x.shape[:2]:
        raise NewsrecError(f"tfm_layer: x {tuple(x.shape)} / mask {tuple(mask.shape)} shape mismatch")
    plist = [_f32(params[n], n) for n in _TFM_ORDER]
    D = x.shape[-1]
    for n, t in zip(_TFM_ORDER, plist):
        want = {"w1": (4 * D, D), "b1": (4 * D,), "w2": (D, 4 * D)}.get(n, (D, D) if n.startswith("w") else (D,))
        if tuple(t.shape) != want:
            raise NewsrecError(f"tfm_layer: parameter {n} has shape {tuple(t.shape)}, expected {want}")
    rows = x.shape[0] if plan is not None else x.shape[0] * x.shape[1]
    if (get_gemm_compute() == "bf16" and tfm_layer_bf16_supported(mask.shape[1], D, int(head_num)) and rows >= 128
            and _contiguous3(plist[0], plist[2], plist[4]) and _contiguous3(plist[1], plist[3], plist[5])):
        # the bf16 build proper: activations / attention tensors bf16 in HBM (BERT-sized layers, parity bar 1e-2)
        return _TFMLayerBf16.apply(x, mask, int(head_num), float(p_drop), int(seed), int(site), 1 if hf_mask else 0, plan,
                                   torch.is_grad_enabled(), *plist)
    return _TFMLayer.apply(x, mask, int(head_num), float(p_drop), int(seed), int(site), 1 if hf_mask else 0, plan,
                           torch.is_grad_enabled(), *plist)


# ----------------------------------------------------------------------------- GRU user encoder
_last_gru_ws = None      # workspace of the most recent GRU call (its head holds the barrier-timeout flag)


def gru_status():
    """raise NewsrecError if the grid barrier of the most recent persistent GRU launch timed out (host sync; the
    outputs of a timed-out launch are NaN-poisoned anyway, this names the cause)."""
    if _last_gru_ws is not None:
        check(lib().nr_gru_status(_p(_last_gru_ws), _stream()), "nr_gru_status")


class _GRUFinal(torch.autograd.Function):
    """final state of nn.GRU over a packed batch (RnnUserEncoder.forward, encoder.py:216-233)."""

    @staticmethod
    def forward(ctx, x, mask, w_ih, w_hh, b_ih, b_hh, grad_mode):
        B, L, Din = x.shape
        H = w_hh.shape[1]
        train = grad_mode and any(ctx.needs_input_grad)          # (see _TFMLayer.forward)
        dev = x.device
        gi = torch.empty(B * L, 3 * H, dtype=torch.float32, device=dev)
        gh = torch.empty(B, 3 * H, dtype=torch.float32, device=dev)
        hall = torch.empty(L + 1, B, H, dtype=torch.float32, device=dev)
        gates = torch.empty(L, B, 4 * H, dtype=torch.float32, device=dev) if train else None
        lens = torch.empty(B, dtype=torch.int32, device=dev)
        ws = _ws(lib().nr_gru_fwd_ws_bytes(B, H), dev)
        check(lib().nr_gru_fwd_f32(_p(x), _p(mask), _p(w_ih), _p(w_hh), _p(b_ih), _p(b_hh), B, L, Din, H, _p(gi), _p(gh), _p(hall),
                                   _p(gates), _p(lens), _p(ws), ws.numel() * 4, _stream()), "nr_gru_fwd_f32")
        global _last_gru_ws
        _last_gru_ws = ws
        if train:
            ctx.save_for_backward(x, w_ih, w_hh, hall, gates, lens)
        return hall[L].clone()

    @staticmethod
    def backward(ctx, g):
        x, w_ih, w_hh, hall, gates, lens = ctx.saved_tensors
        B, L, Din = x.shape
        H = w_hh.shape[1]
        g = _f32(g, "grad")
        dev = g.device
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw_ih, dw_hh = torch.empty_like(w_ih), torch.empty_like(w_hh)
        db_ih = torch.empty(3 * H, dtype=torch.float32, device=dev)
        db_hh = torch.empty(3 * H, dtype=torch.float32, device=dev)
        Lb = lib()
        ws = _ws(Lb.nr_gru_bwd_ws_bytes(B, L, Din, H), dev)
        check(Lb.nr_gru_bwd_f32(_p(g), _p(x), _p(w_ih), _p(w_hh), B, L, Din, H, _p(hall), _p(gates), _p(lens), _p(dx), _p(dw_ih),
                                _p(dw_hh), _p(db_ih), _p(db_hh), _p(ws), ws.numel() * 4, _stream()), "nr_gru_bwd_f32")
        global _last_gru_ws
        _last_gru_ws = ws
        return dx, None, dw_ih, dw_hh, db_ih, db_hh, None


def gru_final_state(x, mask, w_ih, w_hh, b_ih, b_hh):
    """x [B,L,Din], mask [B,L] (1 = clicked news present, at least one per row) -> [B,H]"""
    x, mask = _f32(x, "news_embedding"), _i64(mask, "his_mask")
    if x.dim() != 3 or mask.shape != x.shape[:2]:
        raise NewsrecError(f"gru_final_state: x {tuple(x.shape)} / mask {tuple(mask.shape)} shape mismatch")
    return _GRUFinal.apply(x, mask, _f32(w_ih, "rnn.weight_ih_l0"), _f32(w_hh, "rnn.weight_hh_l0"), _f32(b_ih, "rnn.bias_ih_l0"),
                           _f32(b_hh, "rnn.bias_hh_l0"), torch.is_grad_enabled())


# ----------------------------------------------------------------------------- BERT embedding stage
def colsum(x2d):
    """deterministic column sums of a [R,N] matrix -> [N]"""
    R, N = x2d.shape
    out = torch.empty(N, dtype=torch.float32, device=x2d.device)
    L = lib()
    ws = _ws(L.nr_colsum_ws_bytes(R, N), x2d.device)
    check(L.nr_colsum_f32(_p(x2d), R, N, _p(out), _p(ws), ws.numel() * 4, _stream()), "nr_colsum_f32")
    return out


class _BertEmbed(torch.autograd.Function):
    """HF BertEmbeddings: dropout(LayerNorm(word[ids] + position[0..T-1] + token_type[0])) over ids [M,T]."""

    @staticmethod
    def forward(ctx, word, pos, typ, gamma, beta, ids, eps, p, seed, site, plan):
        """ids [M,T], or with a PackPlan the packed ids [rows] (result then [rows, D])"""
        M, T = ids.shape if plan is None else (plan.N, plan.T)
        D = word.shape[1]
        R = M * T if plan is None else plan.rows
        L = lib()
        pt = torch.empty(T, D, dtype=torch.float32, device=word.device)
        check(L.nr_add_bias_rows_f32(_p(pos), _p(typ), T, D, _p(pt), _stream()), "nr_add_bias_rows_f32")   # pos[:T] + type[0]
        e = gather_rows(word, ids).view(R, D)
        y = torch.empty(R, D, dtype=torch.float32, device=word.device)
        stats = torch.empty(2, R, dtype=torch.float32, device=word.device)
        if plan is None:
            check(L.nr_add_layernorm_fwd_f32(_p(e), _p(pt), T, _p(gamma), _p(beta), R, D, eps, 0.0, 0, 0, _p(y), _p(e),
                                             _p(stats[0]), _p(stats[1]), _stream()), "nr_add_layernorm_fwd_f32")
        else:       # the position row of a packed row is that of its ORIGINAL position (row_map[r] % T)
            check(L.nr_add_layernorm_packed_fwd_f32(_p(e), _p(pt), T, _p(plan.row_map), _p(gamma), _p(beta), R, D, eps, 0.0, 0, 0, _p(y), _p(e),
                                                    _p(stats[0]), _p(stats[1]), _stream()), "nr_add_layernorm_packed_fwd_f32")
        if p > 0:
            yd = torch.empty_like(y)
            if plan is None:
                check(L.nr_dropout_f32(_p(y), y.numel(), p, seed, site, _p(yd), _stream()), "nr_dropout_f32")
            else:
                check(L.nr_dropout_packed_f32(_p(y), _p(plan.row_map), R, D, p, seed, site, _p(yd), _stream()), "nr_dropout_packed_f32")
            y = yd
        ctx.save_for_backward(e, stats, gamma, ids)
        ctx.cfg = (M, T, D, word.shape[0], pos.shape[0], typ.shape[0], p, seed, site, plan)
        return y.view(M, T, D) if plan is None else y

    @staticmethod
    def backward(ctx, g):
        s, stats, gamma, ids = ctx.saved_tensors
        M, T, D, V, NP, NT, p, seed, site, plan = ctx.cfg
        R = M * T if plan is None else plan.rows
        L = lib()
        g = _f32(g, "grad").reshape(R, D)
        if p > 0:
            gd = torch.empty_like(g)
            if plan is None:
                check(L.nr_dropout_f32(_p(g), g.numel(), p, seed, site, _p(gd), _stream()), "nr_dropout_f32")
            else:
                check(L.nr_dropout_packed_f32(_p(g), _p(plan.row_map), R, D, p, seed, site, _p(gd), _stream()), "nr_dropout_packed_f32")
            g = gd
        ds = torch.empty(R, D, dtype=torch.float32, device=g.device)
        dgamma = torch.empty(D, dtype=torch.float32, device=g.device)
        dbeta = torch.empty(D, dtype=torch.float32, device=g.device)
        ws = _ws(L.nr_layernorm_bwd_ws_bytes(R, D), g.device)
        check(L.nr_add_layernorm_bwd_f32(_p(g), _p(s), _p(stats[0]), _p(stats[1]), _p(gamma), R, D, 0.0, 0, 0, _p(ds), None,
                                         _p(dgamma), _p(dbeta), _p(ws), ws.numel() * 4, _stream()), "nr_add_layernorm_bwd_f32")   # (no dropout inside: row_map not needed)
        dword = torch.zeros(V, D, dtype=torch.float32, device=g.device)
        check(L.nr_scatter_add_rows_f32(_p(ds), _p(ids), R, D, V, PAD_ID, _p(dword), _stream()), "nr_scatter_add_rows_f32")
        dpos = torch.zeros(NP, D, dtype=torch.float32, device=g.device)
        if plan is None:
            ds_full = ds
        else:                                                        # packed rows back to [M*T, D] (zeros at dropped positions)
            ds_full = torch.empty(M * T, D, dtype=torch.float32, device=g.device)
            check(L.nr_unpack_rows_f32(_p(ds), _p(plan.row_map), R, D, M * T, _p(ds_full), _stream()), "nr_unpack_rows_f32")
        dpos[:T] = colsum(ds_full.view(M, T * D)).view(T, D)        # every title adds position rows 0..T-1
        dtyp = torch.zeros(NT, D, dtype=torch.float32, device=g.device)
        dtyp[0] = colsum(dpos[:T])                                   # ... and token-type row 0 at every token
        return dword, dpos, dtyp, dgamma, dbeta, None, None, None, None, None, None


def bert_embeddings(word, pos, typ, gamma, beta, ids, eps=LN_EPS, p_drop=0.0, seed=0, site=0, plan=None):
    ids = _i64(ids, "token_id")
    if plan is not None:
        if ids.dim() != 1 or ids.shape[0] != plan.rows or plan.T > pos.shape[0]:
            raise NewsrecError(f"bert_embeddings: packed ids {tuple(ids.shape)} do not match the plan ({plan.rows} rows, T = {plan.T})")
    elif ids.dim() != 2 or ids.shape[1] > pos.shape[0]:
        raise NewsrecError(f"bert_embeddings: ids {tuple(ids.shape)} must be [M,T] with T <= {pos.shape[0]} positions")
    return _BertEmbed.apply(_f32(word, "word_embeddings"), _f32(pos, "position_embeddings"), _f32(typ, "token_type_embeddings"),
                            _f32(gamma, "embeddings.LayerNorm.weight"), _f32(beta, "embeddings.LayerNorm.bias"), ids,
                            float(eps), float(p_drop), int(seed), int(site), plan)


class _FirstRows(torch.autograd.Function):
    """x [rows, D] packed -> the first row of every sequence [N, D] ([CLS] states).  Sequences without rows (titles the
    caller marked as never used) come back as zero vectors and send no gradient anywhere, as the pooling path does:
    a user encoder with a looser history_keep can then neither read another title's vector nor leak a gradient into it."""

    @staticmethod
    def forward(ctx, x, plan):
        idx = plan.offs[:-1].long().clamp_(max=max(plan.rows - 1, 0))
        empty = (plan.offs[1:] == plan.offs[:-1]).unsqueeze(1)
        ctx.save_for_backward(idx, empty)
        ctx.shape = x.shape
        return gather_rows(x, idx).masked_fill_(empty, 0.0)

    @staticmethod
    def backward(ctx, g):
        idx, empty = ctx.saved_tensors
        g = _f32(g, "grad").masked_fill(empty, 0.0)
        R, D = ctx.shape
        dx = torch.zeros(R, D, dtype=torch.float32, device=g.device)
        check(lib().nr_scatter_add_rows_f32(_p(g), _p(idx), idx.numel(), D, R, -1, _p(dx), _stream()), "nr_scatter_add_rows_f32")
        return dx, None


def first_rows(x, plan):
    return _FirstRows.apply(_f32(x, "hidden_states"), plan)


# ----------------------------------------------------------------------------- projection + query pooling
class _ProjPool(torch.autograd.Function):
    """out = softmax(q . tanh(x W^T + b)^T / sqrt(D) + mask) . x  — encoder.py:57,141,326.
    split (packed layout only): the pooled vectors are handed out as TWO tensors, sequences [0, split) and [split, nseq)
    (slices of one buffer: no copy), and the backward reads their two gradients where they are (no concatenation)."""

    @staticmethod
    def forward(ctx, x, w, b, query, mask, plan, split):
        nseq, S = mask.shape
        D = x.shape[-1]
        x2 = x.view(-1, D)
        key = linear_fwd(x2, w, b, "tanh")
        out = torch.empty(nseq, D, dtype=torch.float32, device=x.device)
        prob = torch.empty(nseq, S, dtype=torch.float32, device=x.device)
        if plan is None:
            check(lib().nr_query_pool_fwd_f32(_p(key), _p(x), None, None, 0, _p(query), _p(mask), nseq, S, D, _p(out), _p(prob),
                                              _stream()), "nr_query_pool_fwd_f32")
        else:
            check(lib().nr_query_pool_packed_fwd_f32(_p(key), _p(x), _p(plan.offs), plan.rows, _p(query), _p(mask), nseq, S, D,
                                                     _p(out), _p(prob), _stream()), "nr_query_pool_packed_fwd_f32")
        ctx.save_for_backward(x, w, query, key, prob)
        ctx.plan, ctx.dims, ctx.split = plan, (nseq, S, D), split
        ctx.mark_non_differentiable(prob)
        ctx.set_materialize_grads(False)        # no zero tensor for the output nobody differentiates
        if split is None:
            return out, prob
        return out[:split], out[split:], prob

    @staticmethod
    def backward(ctx, *grads):
        split = ctx.split
        if split is None:
            g, g1 = grads[0], None
            if g is None:
                return (None,) * 7
        else:
            g, g1 = grads[0], grads[1]
            if g is None and g1 is None:
                return (None,) * 7
        x, w, query, key, prob = ctx.saved_tensors
        nseq, S, D = ctx.dims
        plan = ctx.plan
        R = nseq * S if plan is None else plan.rows
        L = lib()
        dkeypre = torch.empty(R, D, dtype=torch.float32, device=x.device)
        dvalue = torch.empty(R, D, dtype=torch.float32, device=x.device)
        dq = torch.empty_like(query)
        ws = _ws(L.nr_query_pool_bwd_ws_bytes(nseq, D), x.device)
        if split is not None:
            g = _f32(g, "grad") if g is not None else torch.zeros(split, D, dtype=torch.float32, device=x.device)
            g1 = _f32(g1, "grad") if g1 is not None else torch.zeros(nseq - split, D, dtype=torch.float32, device=x.device)
            check(L.nr_query_pool_packed_bwd_split_f32(_p(key), _p(x), _p(plan.offs), R, _p(query), _p(prob), _p(g), _p(g1), split, nseq, S, D, 1,
                                                       _p(dkeypre), _p(dvalue), _p(dq), _p(ws), ws.numel() * 4, _stream()),
                  "nr_query_pool_packed_bwd_split_f32")
        elif plan is None:
            g = _f32(g, "grad")
            check(L.nr_query_pool_bwd_f32(_p(key), _p(x), _p(query), _p(prob), _p(g), nseq, S, D, 1, 0, _p(dkeypre), _p(dvalue),
                                          _p(dq), _p(ws), ws.numel() * 4, _stream()), "nr_query_pool_bwd_f32")
        else:
            g = _f32(g, "grad")
            check(L.nr_query_pool_packed_bwd_f32(_p(key), _p(x), _p(plan.offs), R, _p(query), _p(prob), _p(g), nseq, S, D, 1,
                                                 _p(dkeypre), _p(dvalue), _p(dq), _p(ws), ws.numel() * 4, _stream()),
                  "nr_query_pool_packed_bwd_f32")
        dx, dw, db = linear_bwd_both(dkeypre, w, x.view(-1, D), True, True, 1, dvalue)          # dx = dkeypre W + dvalue
        return dx.view(x.shape), dw, db, dq, None, None, None


def project_and_pool(x, w, b, query, mask, plan=None, split=None):
    """x [nseq,S,D] (or packed [rows,D] with a PackPlan) -> (pooled [nseq,D], prob [nseq,S]); with `split` (packed layout):
    (pooled[:split], pooled[split:], prob), two slices of one buffer whose gradients the backward reads in place."""
    if split is not None and (plan is None or not 0 < int(split) < mask.shape[0]):
        raise NewsrecError("project_and_pool: split needs the packed layout and 0 < split < number of sequences")
    return _ProjPool.apply(_f32(x, "x"), _f32(w, "project.weight"), _f32(b, "project.bias"), _f32(query, "query"),
                           _i64(mask, "mask"), plan, None if split is None else int(split))


class _Pool(torch.autograd.Function):
    """generic scaled_dp_attention with one query (attention.py:6-34): separate key / value tensors."""

    @staticmethod
    def forward(ctx, key, value, query, mask):
        nseq, S, D = key.shape
        out = torch.empty(nseq, D, dtype=torch.float32, device=key.device)
        prob = torch.empty(nseq, S, dtype=torch.float32, device=key.device)
        check(lib().nr_query_pool_fwd_f32(_p(key), _p(value), None, None, 0, _p(query), _p(mask), nseq, S, D, _p(out),
                                          _p(prob), _stream()), "nr_query_pool_fwd_f32")
        ctx.save_for_backward(key, value, query, prob)
        ctx.mark_non_differentiable(prob)
        ctx.set_materialize_grads(False)        # no zero tensor for the output nobody differentiates
        return out, prob

    @staticmethod
    def backward(ctx, g, _gprob):
        if g is None:
            return None, None, None, None
        key, value, query, prob = ctx.saved_tensors
        nseq, S, D = key.shape
        g = _f32(g, "grad")
        L = lib()
        dkey = torch.empty_like(key)
        dvalue = torch.empty_like(value)
        dq = torch.empty_like(query)
        ws = _ws(L.nr_query_pool_bwd_ws_bytes(nseq, D), g.device)
        check(L.nr_query_pool_bwd_f32(_p(key), _p(value), _p(query), _p(prob), _p(g), nseq, S, D, 0, 0, _p(dkey), _p(dvalue),
                                      _p(dq), _p(ws), ws.numel() * 4, _stream()), "nr_query_pool_bwd_f32")
        return dkey, dvalue, dq, None


def query_pool(key, value, query, mask=None):
    """-> (out [nseq,D], prob [nseq,S]); key/value [nseq,S,D], query [1,D] or [D], mask [nseq,S] or None."""
    key, value, query = _f32(key, "key"), _f32(value, "value"), _f32(query, "query")
    if key.shape != value.shape:
        raise NewsrecError("query_pool: this build pools values of the key's width only")
    return _Pool.apply(key, value, query, None if mask is None else _i64(mask, "attn_mask"))


class _SelfPool(torch.autograd.Function):
    """AttnUserEncoder: key = value = x, no projection, no mask (encoder.py:298)."""

    @staticmethod
    def forward(ctx, x, query):
        nseq, S, D = x.shape
        out = torch.empty(nseq, D, dtype=torch.float32, device=x.device)
        prob = torch.empty(nseq, S, dtype=torch.float32, device=x.device)
        check(lib().nr_query_pool_fwd_f32(_p(x), _p(x), None, None, 0, _p(query), None, nseq, S, D, _p(out), _p(prob),
                                          _stream()), "nr_query_pool_fwd_f32")
        ctx.save_for_backward(x, query, prob)
        return out

    @staticmethod
    def backward(ctx, g):
        x, query, prob = ctx.saved_tensors
        nseq, S, D = x.shape
        g = _f32(g, "grad")
        L = lib()
        dx = torch.empty_like(x)
        dq = torch.empty_like(query)
        ws = _ws(L.nr_query_pool_bwd_ws_bytes(nseq, D), g.device)
        check(L.nr_query_pool_bwd_f32(_p(x), _p(x), _p(query), _p(prob), _p(g), nseq, S, D, 0, 1, None, _p(dx), _p(dq),
                                      _p(ws), ws.numel() * 4, _stream()), "nr_query_pool_bwd_f32")
        return dx, dq


def self_pool(x, query):
    return _SelfPool.apply(_f32(x, "x"), _f32(query, "query"))


def pool_from_table(table, idx, query):
    """AttnUserEncoder over cached news vectors: rows table[idx[n,s]] (eval only, no autograd)."""
    table, idx, query = _f32(table, "news_embeddings"), _i64(idx, "his_idx"), _f32(query, "query")
    nseq, S = idx.shape
    D = table.shape[1]
    out = torch.empty(nseq, D, dtype=torch.float32, device=table.device)
    check(lib().nr_query_pool_fwd_f32(None, None, _p(table), _p(idx), table.shape[0], _p(query), None, nseq, S, D, _p(out),
                                      None, _stream()), "nr_query_pool_fwd_f32")
    return out


class _SeqReduce(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mean):
        nseq, S, D = x.shape
        out = torch.empty(nseq, D, dtype=torch.float32, device=x.device)
        check(lib().nr_seq_reduce_f32(_p(x), None, None, 0, nseq, S, D, int(mean), _p(out), _stream()), "nr_seq_reduce_f32")
        ctx.dims = (nseq, S, D, int(mean))
        return out

    @staticmethod
    def backward(ctx, g):
        nseq, S, D, mean = ctx.dims
        g = _f32(g, "grad")
        dx = torch.empty(nseq, S, D, dtype=torch.float32, device=g.device)
        check(lib().nr_seq_reduce_bwd_f32(_p(g), nseq, S, D, mean, _p(dx), _stream()), "nr_seq_reduce_bwd_f32")
        return dx, None


def seq_reduce(x, mean=False):
    return _SeqReduce.apply(_f32(x, "x"), bool(mean))


def seq_reduce_from_table(table, idx, mean=False):
    table, idx = _f32(table, "news_embeddings"), _i64(idx, "his_idx")
    nseq, S = idx.shape
    D = table.shape[1]
    out = torch.empty(nseq, D, dtype=torch.float32, device=table.device)
    check(lib().nr_seq_reduce_f32(None, _p(table), _p(idx), table.shape[0], nseq, S, D, int(mean), _p(out), _stream()),
          "nr_seq_reduce_f32")
    return out


# ----------------------------------------------------------------------------- candidate / history split
class _SplitTitles(torch.autograd.Function):
    """news [B, C+L, D] -> (candidates [B,C,D], history [B,L,D]), both contiguous.  Two slices would cost autograd two
    zero-filled [B,C+L,D] buffers, two copies into them and an add; the backward here is one concatenation."""

    @staticmethod
    def forward(ctx, news, C):
        ctx.C, ctx.shape = C, news.shape
        ctx.set_materialize_grads(False)
        return news[:, :C].contiguous(), news[:, C:].contiguous()

    @staticmethod
    def backward(ctx, g_cdd, g_his):
        if g_cdd is None and g_his is None:
            return None, None
        B, n, D = ctx.shape
        C = ctx.C
        ref = g_cdd if g_cdd is not None else g_his
        if g_cdd is None:
            g_cdd = ref.new_zeros(B, C, D)
        if g_his is None:
            g_his = ref.new_zeros(B, n - C, D)
        return torch.cat([g_cdd, g_his], dim=1), None


def split_titles(news, C):
    return _SplitTitles.apply(news, int(C))


class _SplitRows(torch.autograd.Function):
    """x [N, D] -> (x[:split], x[split:]) as two fresh tensors; the backward is one concatenation (two autograd slices
    would cost two zero-filled [N, D] buffers, two copies and an add)."""

    @staticmethod
    def forward(ctx, x, split):
        ctx.split, ctx.shape = split, x.shape
        ctx.set_materialize_grads(False)
        return x[:split].clone(), x[split:].clone()

    @staticmethod
    def backward(ctx, g0, g1):
        if g0 is None and g1 is None:
            return None, None
        N, D = ctx.shape
        ref = g0 if g0 is not None else g1
        if g0 is None:
            g0 = ref.new_zeros(ctx.split, D)
        if g1 is None:
            g1 = ref.new_zeros(N - ctx.split, D)
        return torch.cat([g0, g1], dim=0), None


def split_rows(x, split):
    if x.dim() != 2 or not 0 <= int(split) <= x.shape[0]:
        raise NewsrecError(f"split_rows: x {tuple(x.shape)} / split {split}")
    return _SplitRows.apply(x, int(split))


# ----------------------------------------------------------------------------- scoring (+ cross entropy)
class _Score(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cdd, user):
        B, Cn, D = cdd.shape
        logits = torch.empty(B, Cn, dtype=torch.float32, device=cdd.device)
        check(lib().nr_score_fwd_f32(_p(cdd), None, None, 0, _p(user), None, B, Cn, D, _p(logits), None, None, _stream()),
              "nr_score_fwd_f32")
        ctx.save_for_backward(cdd, user)
        return logits

    @staticmethod
    def backward(ctx, g):
        cdd, user = ctx.saved_tensors
        B, Cn, D = cdd.shape
        g = _f32(g, "grad")
        d_cdd = torch.empty_like(cdd)
        d_user = torch.empty_like(user)
        check(lib().nr_score_bwd_f32(_p(cdd), _p(user), _p(g), B, Cn, D, _p(d_cdd), _p(d_user), _stream()), "nr_score_bwd_f32")
        return d_cdd, d_user


def score(cdd, user):
    """_compute_logits (BaseModel.py:164-175): cdd [B,C,D], user [B,1,D] or [B,D] -> [B,C]."""
    cdd, user = _f32(cdd, "cdd_news_repr"), _f32(user, "user_repr")
    return _Score.apply(cdd, user.reshape(cdd.shape[0], cdd.shape[2]))


class _ScoreCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, cdd, user, label):
        B, Cn, D = cdd.shape
        logits = torch.empty(B, Cn, dtype=torch.float32, device=cdd.device)
        rows = torch.empty(B, dtype=torch.float32, device=cdd.device)
        loss = torch.empty((), dtype=torch.float32, device=cdd.device)
        check(lib().nr_score_fwd_f32(_p(cdd), None, None, 0, _p(user), _p(label), B, Cn, D, _p(logits), _p(rows), _p(loss),
                                     _stream()), "nr_score_fwd_f32")
        ctx.save_for_backward(cdd, user, label, logits)
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)        # no zero tensor for the output nobody differentiates
        return loss, logits

    @staticmethod
    def backward(ctx, g, _gl):
        if g is None:
            return None, None, None
        cdd, user, label, logits = ctx.saved_tensors
        B, Cn, D = cdd.shape
        g = _f32(g, "grad").reshape(1)
        d_cdd = torch.empty_like(cdd)
        d_user = torch.empty_like(user)
        check(lib().nr_score_ce_bwd_f32(_p(cdd), _p(user), _p(label), _p(logits), _p(g), B, Cn, D, _p(d_cdd), _p(d_user),
                                        _stream()), "nr_score_ce_bwd_f32")
        return d_cdd, d_user, None


def score_cross_entropy(cdd, user, label):
    """fused _compute_logits + nn.CrossEntropyLoss() (BaseModel.py:202-204) -> (loss, logits)."""
    cdd, user, label = _f32(cdd, "cdd_news_repr"), _f32(user, "user_repr"), _i64(label, "label")
    if label.numel() and (int(label.shape[0]) != cdd.shape[0]):
        raise NewsrecError("score_cross_entropy: label/batch mismatch")
    return _ScoreCE.apply(cdd, user.reshape(cdd.shape[0], cdd.shape[2]), label)


def score_from_table(table, cdd_idx, user):
    """evaluation gather-dot: logits[r,c] = table[cdd_idx[r,c]] . user[r] / sqrt(D) (BaseModel.py:214-217)."""
    table, cdd_idx, user = _f32(table, "news_embeddings"), _i64(cdd_idx, "cdd_idx"), _f32(user, "user_repr")
    B, Cn = cdd_idx.shape
    D = table.shape[1]
    user = user.reshape(B, D)
    logits = torch.empty(B, Cn, dtype=torch.float32, device=table.device)
    check(lib().nr_score_fwd_f32(None, _p(table), _p(cdd_idx), table.shape[0], _p(user), None, B, Cn, D, _p(logits), None,
                                 None, _stream()), "nr_score_fwd_f32")
    return logits


# ----------------------------------------------------------------------------- evaluation metrics
def impression_metrics(preds, labels, masks, row_start, row_count, ks):
    """per-impression (auc, mrr, ndcg@k...) [G, 2+len(ks)] float64 and flags [G] int32 (see newsrec_hip.h);
    preds [R,Ce] f32, labels [R,Ce] int64, masks [R,Ce] bool/uint8, row_start [G] int64, row_count [G] int32."""
    preds, labels = _f32(preds, "preds"), _i64(labels, "labels")
    masks = masks.to(torch.uint8).contiguous()
    row_start = _i64(row_start, "row_start")
    row_count = row_count.to(torch.int32).contiguous()
    G, Ce = row_start.numel(), preds.shape[1]
    out = torch.empty(G, 2 + len(ks), dtype=torch.float64, device=preds.device)
    flags = torch.empty(G, dtype=torch.int32, device=preds.device)
    karr = (C.c_int * max(len(ks), 1))(*[int(k) for k in ks])
    check(lib().nr_impression_metrics_f64(_p(preds), _p(labels), _p(masks), Ce, _p(row_start), _p(row_count), G, len(ks), karr,
                                          _p(out), _p(flags), _stream()), "nr_impression_metrics_f64")
    return out, flags


# ----------------------------------------------------------------------------- Adam
def adam_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    check(lib().nr_adam_step_f32(_p(param), _p(grad), _p(exp_avg), _p(exp_avg_sq), param.numel(), lr, beta1, beta2, eps, step,
                                 _stream()), "nr_adam_step_f32")


def adam_multi(params, grads, exp_avgs, exp_avg_sqs, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """one launch (per 48 tensors) for every parameter that shares `step`; tensors must be contiguous fp32 on one GPU"""
    n = len(params)
    if n == 0:
        return
    PT, NT = C.c_void_p * n, C.c_int64 * n
    tabs = [PT(*[_p(t) for t in ts]) for ts in (params, grads, exp_avgs, exp_avg_sqs)]
    numel = NT(*[t.numel() for t in params])
    check(lib().nr_adam_multi_f32(n, tabs[0], tabs[1], tabs[2], tabs[3], numel, lr, beta1, beta2, eps, step, _stream()),
          "nr_adam_multi_f32")


def adam_tables(params, exp_avgs, exp_avg_sqs):
    """the step-invariant part of adam_multi's arguments: pointer tables of the parameters and both moments, sizes.  The
    tensors are kept alive by the tuple."""
    n = len(params)
    PT, NT = C.c_void_p * n, C.c_int64 * n
    tabs = [PT(*[_p(t) for t in ts]) for ts in (params, exp_avgs, exp_avg_sqs)]
    return n, tabs[0], tabs[1], tabs[2], NT(*[t.numel() for t in params]), PT, (tuple(params), tuple(exp_avgs), tuple(exp_avg_sqs))


def adam_multi_cached(static, grads, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    n, tp, tm, tv, numel, PT, _keep = static
    if len(grads) != n:
        raise NewsrecError("adam_multi_cached: gradient list does not match the cached tables")
    tg = PT(*[g.data_ptr() for g in grads])
    check(lib().nr_adam_multi_f32(n, tp, tg, tm, tv, numel, lr, beta1, beta2, eps, step, _stream()), "nr_adam_multi_f32")


# ----------------------------------------------------------------------------- profiler
def prof_enable(on=True):
    check(lib().nr_prof_enable(1 if on else 0), "nr_prof_enable")


def prof_read(family):
    ms, n, work = C.c_double(), C.c_int64(), C.c_double()
    check(lib().nr_prof_read(family, C.byref(ms), C.byref(n), C.byref(work)), "nr_prof_read")
    return ms.value, n.value, work.value


# ----------------------------------------------------------------------------- GEMM precision
_GEMM_MODES = {"fp32": 2, "bf16": 1, "fp32-mfma": 0}


def set_gemm_compute(mode):
    """how the large dense products are computed (process-wide, sticky; nr_set_gemm_compute):
      "fp32"       the default.  f32 results from three bf16 pieces per operand on the bf16 matrix pipe (gemm_x3_kernel):
                   f32 accuracy, the 1e-5 parity bar, 1.3-1.6x the rate of the f32 matrix instruction; products too small
                   to fill the chip with its tiles stay on that instruction
      "fp32-mfma"  every product on the f32 matrix instruction (v_mfma_f32_32x32x2_f32), the round-1 path
      "bf16"       operands rounded to bf16, f32 accumulate: the 1e-2 parity bar of SURVEY §8 a7"""
    if mode not in _GEMM_MODES:
        raise NewsrecError(f"set_gemm_compute: unknown mode {mode!r} ({' | '.join(_GEMM_MODES)})")
    check(lib().nr_set_gemm_compute(_GEMM_MODES[mode]), "nr_set_gemm_compute")


def get_gemm_compute():
    v = lib().nr_get_gemm_compute()
    return next(k for k, m in _GEMM_MODES.items() if m == v)
