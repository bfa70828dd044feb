"""Attention primitives of the two-tower path, computed by the MI355X HIP kernels.

Mirrors the public names of the reference's models/modules/attention.py
(scaled_dp_attention, extend_attention_mask, TFM* modules) and keeps its
parameter names, so reference checkpoints load unchanged.  The torch modules
below only HOLD parameters; the arithmetic runs in libnewsrec_hip.so.
"""
import torch
import torch.nn as nn

from newsrec_hip import ops


def scaled_dp_attention(query, key, value, attn_mask=None, return_prob=False):
    """single-query scaled dot-product pooling (reference attention.py:6-34).

    query [1, D]; key/value [..., S, D]; attn_mask [..., 1, S] of 0/1 or None.
    Returns [..., 1, D] (and the probabilities [..., 1, S] if return_prob).
    Masked positions get an additive -1e5 (not -inf), exactly as the reference.
    """
    if query.dim() != 2 or query.shape[0] != 1:
        raise ops.NewsrecError("scaled_dp_attention: the HIP kernel pools with ONE query vector ([1, D])")
    lead = key.shape[:-2]
    S, D = key.shape[-2:]
    k3 = key.reshape(-1, S, D)
    v3 = value.reshape(-1, S, D)
    m2 = None if attn_mask is None else attn_mask.reshape(-1, S)
    out, prob = ops.query_pool(k3, v3, query, m2)
    out = out.view(*lead, 1, D)
    if return_prob:
        return out, prob.view(*lead, 1, S)
    return out


def extend_attention_mask(encoder_attention_mask):
    """0/1 mask -> additive fp32 mask ((1-m) * -1e5), broadcastable over heads
    (reference attention.py:37-51).  Kept for API parity; the fused layer takes
    the 0/1 mask directly and applies the same additive term in-kernel."""
    m = encoder_attention_mask
    if m.dim() == 3:
        m = m[:, None, :, :]
    elif m.dim() == 2:
        m = m[:, None, None, :]
    return (1 - m) * -1e5


def _key_mask01(attention_mask, N, S, device):
    """the sub-modules accept what the reference passes them — the EXTENDED additive mask [N,1,1,S] of
    extend_attention_mask (0 or -1e5) — or a plain 0/1 mask [N,S], or None"""
    if attention_mask is None:
        return torch.ones(N, S, dtype=torch.int64, device=device)
    m = attention_mask
    if m.is_floating_point():
        m = (m == 0)
    return m.reshape(N, S).long()


def _site_p(module):
    """(p, seed, site) of a module's dropout site: nn.Dropout's p in train mode, else 0"""
    p = float(module.dropout.p) if module.training else 0.0
    seed, site = ops.dropout_stream.next(1) if p > 0 else (0, 0)
    return p, seed, site


class TFMSelfAttention(nn.Module):
    def __init__(self, hidden_dim, head_num, dropout_p):
        super().__init__()
        self.num_attention_heads = head_num
        self.attention_head_size = int(hidden_dim / head_num)
        self.all_head_size = self.num_attention_heads * self.attention_head_size
        self.query = nn.Linear(hidden_dim, self.all_head_size)
        self.key = nn.Linear(hidden_dim, self.all_head_size)
        self.value = nn.Linear(hidden_dim, self.all_head_size)
        self.dropout = nn.Dropout(dropout_p)

    def forward(self, hidden_states, attention_mask=None):
        """[N,S,D] -> context [N,S,D] (reference attention.py:77-99): three dense layers + the attention core kernel.
        TFMLayer does not call this (it runs the whole layer as one fused C call); it exists so that the public
        sub-modules of the reference stay callable on their own."""
        N, S, _ = hidden_states.shape
        q = ops.linear(hidden_states, self.query.weight, self.query.bias)
        k = ops.linear(hidden_states, self.key.weight, self.key.bias)
        v = ops.linear(hidden_states, self.value.weight, self.value.bias)
        qkv = torch.cat([q, k, v], dim=-1).view(N * S, 3 * self.all_head_size)
        p, seed, site = _site_p(self)
        ctx = ops.mhsa(qkv, _key_mask01(attention_mask, N, S, hidden_states.device), N, S, self.num_attention_heads,
                       self.attention_head_size, p, seed, site)
        return ctx.view(N, S, self.all_head_size)


class TFMSelfOutput(nn.Module):
    def __init__(self, hidden_dim, dropout_p):
        super().__init__()
        self.dense = nn.Linear(hidden_dim, hidden_dim)
        self.LayerNorm = nn.LayerNorm(hidden_dim, eps=1e-12)
        self.dropout = nn.Dropout(dropout_p)

    def forward(self, hidden_states, input_tensor):
        """LayerNorm(dropout(dense(hidden_states)) + input_tensor) (reference attention.py:109-113)"""
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        p, seed, site = _site_p(self)
        return ops.add_layernorm(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, p, seed, site)


class TFMAttention(nn.Module):
    def __init__(self, hidden_dim, head_num, dropout_p):
        super().__init__()
        self.self = TFMSelfAttention(hidden_dim=hidden_dim, head_num=head_num, dropout_p=dropout_p)
        self.output = TFMSelfOutput(hidden_dim=hidden_dim, dropout_p=dropout_p)

    def forward(self, hidden_states, attention_mask=None):
        return self.output(self.self(hidden_states, attention_mask), hidden_states)


class TFMIntermediate(nn.Module):
    def __init__(self, hidden_dim):
        super().__init__()
        self.dense = nn.Linear(hidden_dim, 4 * hidden_dim)

    def forward(self, hidden_states):
        """erf-GELU(dense(hidden_states)) (reference attention.py:141-144)"""
        return ops.linear_gelu(hidden_states, self.dense.weight, self.dense.bias)


class TFMOutput(nn.Module):
    def __init__(self, hidden_dim, dropout_p):
        super().__init__()
        self.dense = nn.Linear(4 * hidden_dim, hidden_dim)
        self.LayerNorm = nn.LayerNorm(hidden_dim, eps=1e-12)
        self.dropout = nn.Dropout(dropout_p)

    def forward(self, hidden_states, input_tensor):
        """LayerNorm(dropout(dense(hidden_states)) + input_tensor) (reference attention.py:154-158)"""
        h = ops.linear(hidden_states, self.dense.weight, self.dense.bias)
        p, seed, site = _site_p(self)
        return ops.add_layernorm(h, input_tensor, self.LayerNorm.weight, self.LayerNorm.bias, self.LayerNorm.eps, p, seed, site)


class TFMLayer(nn.Module):
    """one post-LN transformer layer (reference attention.py:161-182), executed as ONE fused
    C call (QKV GEMM, per-head attention, out-proj, residual+LayerNorm, FFN with erf-GELU,
    residual+LayerNorm) through newsrec_hip.ops.tfm_layer."""

    def __init__(self, hidden_dim, head_num, dropout_p, hf_mask=False):
        super().__init__()
        # hf_mask: this layer sits inside an HF BertModel (AllBertNewsEncoder): same arithmetic, but masked keys
        # are excluded the SDPA way instead of getting the reference layer's additive -1e5
        self.hf_mask = bool(hf_mask)
        if hidden_dim % head_num != 0:
            raise ValueError(f"hidden_dim {hidden_dim} must be divisible by head_num {head_num} "
                             "(the reference's output dense layer requires it too)")
        self.hidden_dim = hidden_dim
        self.head_num = head_num
        self.dropout_p = float(dropout_p)
        # same construction order as the reference -> same default init for a given torch seed
        self.attention = TFMAttention(hidden_dim=hidden_dim, head_num=head_num, dropout_p=dropout_p)
        self.intermediate = TFMIntermediate(hidden_dim=hidden_dim)
        self.output = TFMOutput(hidden_dim=hidden_dim, dropout_p=dropout_p)

    def _coalesce_qkv(self):
        """keep query|key|value weights (and biases) back to back in memory: the layer multiplies by ONE [3D, D] operand,
        and contiguous parameters spare it six device copies per call.  The parameters stay three separate
        nn.Parameters with the reference's names; only their storage is shared.  Re-done whenever .to()/.cuda() or a
        load replaced the storage (the check is pointer arithmetic)."""
        a = self.attention.self
        ws, bs = (a.query.weight, a.key.weight, a.value.weight), (a.query.bias, a.key.bias, a.value.bias)
        def contiguous3(ts):
            n = ts[0].numel() * ts[0].element_size()
            return all(t.is_contiguous() for t in ts) and ts[1].data_ptr() == ts[0].data_ptr() + n and ts[2].data_ptr() == ts[1].data_ptr() + n
        if contiguous3(ws) and contiguous3(bs):
            return
        with torch.no_grad():
            for ts in (ws, bs):
                packed = torch.cat([t.data.reshape(-1) for t in ts])
                n = ts[0].numel()
                for k, t in enumerate(ts):
                    t.data = packed[k * n:(k + 1) * n].view(t.shape)

    def kernel_params(self):
        self._coalesce_qkv()
        a, i, o = self.attention, self.intermediate, self.output
        return dict(wq=a.self.query.weight, bq=a.self.query.bias, wk=a.self.key.weight, bk=a.self.key.bias,
                    wv=a.self.value.weight, bv=a.self.value.bias, wo=a.output.dense.weight, bo=a.output.dense.bias,
                    ln1_g=a.output.LayerNorm.weight, ln1_b=a.output.LayerNorm.bias, w1=i.dense.weight, b1=i.dense.bias,
                    w2=o.dense.weight, b2=o.dense.bias, ln2_g=o.LayerNorm.weight, ln2_b=o.LayerNorm.bias)

    def forward(self, hidden_states, attention_mask=None, plan=None):
        """plan (extension): a newsrec_hip.ops.PackPlan -> hidden_states is the packed [rows, D] matrix"""
        if attention_mask is None:
            attention_mask = torch.ones(hidden_states.shape[:2], dtype=torch.int64, device=hidden_states.device)
        p = self.dropout_p if self.training else 0.0
        seed, site = ops.dropout_stream.next(3) if p > 0 else (0, 0)
        return ops.tfm_layer(hidden_states, attention_mask, self.head_num, self.kernel_params(), p, seed, site, self.hf_mask,
                             plan=plan)
