"""News / user encoders selectable from main.twotower, on the MI355X HIP kernels.

Same class names, constructor argument (``manager``), forward signatures and
parameter names as the reference's models/modules/encoder.py, so its
checkpoints load and ``main.twotower`` picks them by the same flags.  torch
modules are parameter containers; all arithmetic goes through newsrec_hip.ops
(including the GRU of ``RnnUserEncoder``).
"""
import os

import torch
import torch.nn as nn

from newsrec_hip import ops
from .attention import TFMLayer


def load_word_embeddings(manager):
    """the token table of the reference = HF BERT ``embeddings.word_embeddings``
    (nn.Embedding(vocab, plm_dim, padding_idx=0)), read from the local ``manager.plm_dir``.

    Extension for synthetic runs (bench / tests, no PLM on disk): if the manager carries
    ``synthetic_plm=True`` a random table of the same shape is created instead.
    """
    if getattr(manager, "synthetic_plm", False):
        emb = nn.Embedding(manager.vocab_size, manager.plm_dim, padding_idx=0)
        nn.init.normal_(emb.weight, std=0.02)
        return emb
    if not os.path.isdir(manager.plm_dir):
        raise FileNotFoundError(
            f"PLM directory {manager.plm_dir} not found. This build never downloads models: place a local "
            "BERT directory there (config.json + weights), or set manager.synthetic_plm = True for synthetic runs.")
    from transformers import AutoModel
    return AutoModel.from_pretrained(manager.plm_dir).embeddings.word_embeddings


class BaseNewsEncoder(nn.Module):
    def __init__(self, manager):
        super().__init__()
        self.name = type(self).__name__[:-11]       # "CnnNewsEncoder" -> "Cnn"


class BaseUserEncoder(nn.Module):
    def __init__(self):
        super().__init__()
        self.name = type(self).__name__[:-11]


class _QueryPooled(object):
    """shared by the encoders that end in tanh-projected single-query pooling."""

    @staticmethod
    def make(dim):
        query = nn.Parameter(torch.randn((1, dim), requires_grad=True))
        nn.init.xavier_normal_(query)
        proj = nn.Linear(dim, dim)
        nn.init.xavier_normal_(proj.weight)
        return query, proj


class CnnNewsEncoder(BaseNewsEncoder):
    """token gather -> Conv1d(k=3,pad=1)+ReLU (implicit f32-MFMA GEMM) -> query pooling."""

    def __init__(self, manager):
        super().__init__(manager)
        self.embedding_dim = manager.plm_dim
        self.embedding = load_word_embeddings(manager)
        self.cnn = nn.Conv1d(in_channels=self.embedding_dim, out_channels=manager.hidden_dim, kernel_size=3, padding=1)
        nn.init.xavier_normal_(self.cnn.weight)
        self.news_query, self.newsProject = _QueryPooled.make(manager.hidden_dim)

    def forward(self, token_id, attn_mask, token_weight=None):
        shape = token_id.shape
        T = shape[-1]
        ids = token_id.reshape(-1, T)
        mask = attn_mask.reshape(-1, T)
        emb = ops.embedding(self.embedding.weight, ids, self.embedding.padding_idx)
        if token_weight is not None:
            emb = emb * token_weight.reshape(-1, T).unsqueeze(-1)
        tok = ops.conv1d_k3_relu(emb, self.cnn.weight, self.cnn.bias)
        news, _ = ops.project_and_pool(tok, self.newsProject.weight, self.newsProject.bias, self.news_query, mask)
        return tok.view(*shape, -1), news.view(*shape[:-1], -1)


class TfmNewsEncoder(BaseNewsEncoder):
    """token gather -> one fused transformer layer -> query pooling (NRMS-style)."""

    def __init__(self, manager):
        super().__init__(manager)
        self.embedding_dim = manager.plm_dim
        self.embedding = load_word_embeddings(manager)
        if self.embedding_dim != manager.hidden_dim:
            raise ValueError(f"TfmNewsEncoder feeds token embeddings ({self.embedding_dim}-d) straight into a "
                             f"{manager.hidden_dim}-d transformer layer: the two must be equal (as in the reference)")
        self.transformer = TFMLayer(manager.hidden_dim, manager.head_num, 0.1)
        self.news_query, self.newsProject = _QueryPooled.make(manager.hidden_dim)

    def forward(self, token_id, attn_mask, token_weight=None):
        shape = token_id.shape
        T = shape[-1]
        ids = token_id.reshape(-1, T)
        mask = attn_mask.reshape(-1, T)
        emb = ops.embedding(self.embedding.weight, ids, self.embedding.padding_idx)
        if token_weight is not None:
            emb = emb * token_weight.reshape(-1, T).unsqueeze(-1)
        tok = self.transformer(emb, attention_mask=mask)
        news, _ = ops.project_and_pool(tok, self.newsProject.weight, self.newsProject.bias, self.news_query, mask)
        return tok.view(*shape, -1), news.view(*shape[:-1], -1)

    def plan_begin(self, token_id, attn_mask, title_keep=None):
        """start the pack plan news_vectors() will need for these titles (ops.pack_titles_begin); None = no packing"""
        T = token_id.shape[-1]
        return ops.pack_titles_begin(attn_mask.reshape(-1, T), title_keep) if self.packable(T) else None

    def news_vectors(self, token_id, attn_mask, title_keep=None, pending=None, split=None):
        """the news vectors alone, for callers that drop token_embedding (TwoTowerBaseModel.forward / encode_news):
        same numbers as forward()[1], computed over the real token rows only (packed title tokens, include/newsrec_hip.h)
        whenever the masks are prefix-shaped and T <= 32; otherwise through forward().  title_keep [...] 0/1 marks
        titles whose vector the caller never uses (masked history slots): they are not encoded and come back as zeros.
        split (token_id must be [N, T]): return (vectors[:split], vectors[split:]) — two slices of one buffer, no copy,
        and the backward reads the two gradients where they are."""
        shape = token_id.shape
        T = shape[-1]
        mask = attn_mask.reshape(-1, T)
        packed = ops.pack_titles(token_id.reshape(-1, T), mask, title_keep, pending) if self.packable(T) else None
        if packed is None:
            news = self.forward(token_id, attn_mask)[1]
            return news if split is None else ops.split_rows(news.reshape(-1, news.shape[-1]), split)
        ids_p, plan = packed
        emb = ops.embedding(self.embedding.weight, ids_p, self.embedding.padding_idx)        # [rows, E]
        tok = self.transformer(emb, attention_mask=mask, plan=plan)
        if split is not None and 0 < split < mask.shape[0]:
            a, b, _ = ops.project_and_pool(tok, self.newsProject.weight, self.newsProject.bias, self.news_query, mask, plan=plan, split=split)
            return a, b
        news, _ = ops.project_and_pool(tok, self.newsProject.weight, self.newsProject.bias, self.news_query, mask, plan=plan)
        return news.view(*shape[:-1], -1) if split is None else ops.split_rows(news, split)

    def packable(self, T):
        return T <= 32 and self.transformer.hidden_dim // self.transformer.head_num <= 64 and not getattr(self, "no_packing", False)


class _BertEmbeddings(nn.Module):
    """parameter container with HF BertEmbeddings' names"""

    def __init__(self, cfg):
        super().__init__()
        D = cfg["hidden_size"]
        self.word_embeddings = nn.Embedding(cfg["vocab_size"], D, padding_idx=cfg.get("pad_token_id", 0))
        self.position_embeddings = nn.Embedding(cfg["max_position_embeddings"], D)
        self.token_type_embeddings = nn.Embedding(cfg.get("type_vocab_size", 2), D)
        self.LayerNorm = nn.LayerNorm(D, eps=cfg.get("layer_norm_eps", 1e-12))


class _BertEncoder(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layer = nn.ModuleList([TFMLayer(cfg["hidden_size"], cfg["num_attention_heads"], cfg.get("hidden_dropout_prob", 0.1),
                                             hf_mask=True) for _ in range(cfg["num_hidden_layers"])])


class HipBertModel(nn.Module):
    """HF ``BertModel`` without its pooler (the reference drops it, encoder.py:66), as a parameter container
    whose state-dict keys equal HF's (embeddings.*, encoder.layer.N.*) and whose arithmetic runs in the HIP
    kernels: the BERT layer IS the reference's TFMLayer (attention.py copies BERT's structure and names)."""

    def __init__(self, cfg):
        super().__init__()
        if cfg.get("hidden_act", "gelu") != "gelu":
            raise ValueError("HipBertModel: only the exact erf GELU of bert-base is built")
        if cfg.get("intermediate_size", 4 * cfg["hidden_size"]) != 4 * cfg["hidden_size"]:
            raise ValueError("HipBertModel: intermediate_size must be 4 x hidden_size (bert-base: 3072 = 4 x 768)")
        if cfg.get("attention_probs_dropout_prob", 0.1) != cfg.get("hidden_dropout_prob", 0.1):
            raise ValueError("HipBertModel: the fused layer uses one dropout probability for its three sites")
        if cfg.get("position_embedding_type", "absolute") != "absolute":
            raise ValueError("HipBertModel: absolute position embeddings only")
        self.cfg = dict(cfg)
        self.embeddings = _BertEmbeddings(cfg)
        self.encoder = _BertEncoder(cfg)
        self.dropout_p = float(cfg.get("hidden_dropout_prob", 0.1))
        std = cfg.get("initializer_range", 0.02)                    # HF _init_weights
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.normal_(m.weight, std=std)
                nn.init.zeros_(m.bias)
            elif isinstance(m, nn.Embedding):
                nn.init.normal_(m.weight, std=std)
                if m.padding_idx is not None:
                    with torch.no_grad():
                        m.weight[m.padding_idx].zero_()

    @classmethod
    def from_pretrained(cls, plm_dir):
        """local directory only (config.json + model.safetensors | pytorch_model.bin); never downloads."""
        import json
        with open(os.path.join(plm_dir, "config.json")) as f:
            cfg = json.load(f)
        model = cls(cfg)
        st_path = os.path.join(plm_dir, "model.safetensors")
        if os.path.exists(st_path):
            from safetensors.torch import load_file
            state = load_file(st_path)
        else:
            state = torch.load(os.path.join(plm_dir, "pytorch_model.bin"), map_location="cpu", weights_only=True)
        clean = {}
        for k, v in state.items():
            k = k[5:] if k.startswith("bert.") else k
            if k.startswith("pooler.") or k.endswith("position_ids") or k.startswith("cls."):
                continue
            clean[k.replace("LayerNorm.gamma", "LayerNorm.weight").replace("LayerNorm.beta", "LayerNorm.bias")] = v
        model.load_state_dict(clean, strict=True)
        return model

    def forward(self, token_id, attn_mask, plan=None):
        """ids/mask [M,T] -> last_hidden_state [M,T,D]; with a PackPlan, token_id is the packed id vector [rows] and the
        result the packed [rows, D] states"""
        e = self.embeddings
        p = self.dropout_p if self.training else 0.0
        seed, site = ops.dropout_stream.next(1) if p > 0 else (0, 0)
        x = ops.bert_embeddings(e.word_embeddings.weight, e.position_embeddings.weight, e.token_type_embeddings.weight,
                                e.LayerNorm.weight, e.LayerNorm.bias, token_id, e.LayerNorm.eps, p, seed, site, plan=plan)
        for layer in self.encoder.layer:
            x = layer(x, attention_mask=attn_mask, plan=plan)
        return x


class AllBertNewsEncoder(BaseNewsEncoder):
    """BERT title encoder, news vector = [CLS] state (reference encoder.py:62-77).  ``self.plm`` keeps HF
    BertModel's parameter names (minus the pooler the reference removes), so reference checkpoints load."""

    def __init__(self, manager):
        super().__init__(manager)
        if getattr(manager, "synthetic_plm", False):
            self.plm = HipBertModel(dict(
                vocab_size=manager.vocab_size, hidden_size=manager.plm_dim, num_attention_heads=getattr(manager, "plm_heads", 12),
                num_hidden_layers=getattr(manager, "plm_layers", 12), intermediate_size=4 * manager.plm_dim,
                max_position_embeddings=getattr(manager, "plm_max_positions", 512), type_vocab_size=2, layer_norm_eps=1e-12, hidden_dropout_prob=0.1,
                attention_probs_dropout_prob=0.1))
        else:
            if not os.path.isdir(manager.plm_dir):
                raise FileNotFoundError(f"PLM directory {manager.plm_dir} not found (this build never downloads models)")
            self.plm = HipBertModel.from_pretrained(manager.plm_dir)

    def forward(self, token_id, attn_mask, token_weight=None):
        shape = token_id.shape
        T = shape[-1]
        tok = self.plm(token_id.reshape(-1, T), attn_mask.reshape(-1, T))
        news = tok[:, 0].reshape(*shape[:-1], -1)
        return tok.view(*shape, -1), news

    def packable(self, T):
        cfg = self.plm.cfg
        return T <= 32 and cfg["hidden_size"] // cfg["num_attention_heads"] <= 64 and not getattr(self, "no_packing", False)

    def plan_begin(self, token_id, attn_mask, title_keep=None):
        T = token_id.shape[-1]
        return ops.pack_titles_begin(attn_mask.reshape(-1, T), title_keep) if self.packable(T) else None

    def news_vectors(self, token_id, attn_mask, title_keep=None, pending=None, split=None):
        """the [CLS] vectors alone, over the real token rows only (packed title tokens, include/newsrec_hip.h): HF's masked
        keys are excluded outright, so a padded position reaches neither another token nor the [CLS] row.  Same numbers as
        forward()[1]; titles `title_keep` marks 0 are not encoded.  split (token_id [N, T]): (vectors[:split], vectors[split:])."""
        shape = token_id.shape
        T = shape[-1]
        mask = attn_mask.reshape(-1, T)
        packed = ops.pack_titles(token_id.reshape(-1, T), mask, title_keep, pending) if self.packable(T) else None
        if packed is None:
            news = self.forward(token_id, attn_mask)[1]
        else:
            ids_p, plan = packed
            tok = self.plm(ids_p, mask, plan=plan)
            news = ops.first_rows(tok, plan).view(*shape[:-1], -1)
        return news if split is None else ops.split_rows(news.reshape(-1, news.shape[-1]), split)


class RnnUserEncoder(BaseUserEncoder):
    """final GRU state over the clicked-news vectors (the default -ue of main.twotower).  ``self.rnn`` is an
    nn.GRU used as a parameter container only (reference key names rnn.weight_ih_l0 …); the recurrence runs
    in the HIP library (input half as one MFMA GEMM, 2 small launches per step, no lens.cpu() sync)."""
    history_keep_mode = 2      # nr_assemble_titles: slot index < history length (the same rule as history_keep below)

    @staticmethod
    def history_keep(his_mask):
        """[B,L] 0/1: the history slots that can reach the result.  The packed GRU reads the first len = sum(his_mask)
        slots and nothing else (reference encoder.py:226-232)."""
        L = his_mask.shape[1]
        return (torch.arange(L, device=his_mask.device)[None, :] < his_mask.sum(1, keepdim=True)).long()

    def __init__(self, manager):
        super().__init__()
        self.rnn = nn.GRU(manager.hidden_dim, manager.hidden_dim, batch_first=True)
        for name, param in self.rnn.named_parameters():
            if "weight" in name:
                nn.init.orthogonal_(param)

    def forward(self, news_embedding, his_mask):
        r = self.rnn
        h = ops.gru_final_state(news_embedding, his_mask, r.weight_ih_l0, r.weight_hh_l0, r.bias_ih_l0, r.bias_hh_l0)
        return h.unsqueeze(1)


class SumUserEncoder(BaseUserEncoder):
    def __init__(self, manager):
        super().__init__()

    def forward(self, news_embedding, **kargs):
        return ops.seq_reduce(news_embedding, mean=False).unsqueeze(-2)


class AvgUserEncoder(BaseUserEncoder):
    def __init__(self, manager):
        super().__init__()

    def forward(self, news_embedding, **kargs):
        return ops.seq_reduce(news_embedding, mean=True).unsqueeze(-2)


class AttnUserEncoder(BaseUserEncoder):
    def __init__(self, manager):
        super().__init__()
        self.user_query = nn.Parameter(torch.randn((1, manager.hidden_dim), requires_grad=True))
        nn.init.xavier_normal_(self.user_query)

    def forward(self, news_embedding, **kargs):
        # key = value = history, no projection and no mask (reference encoder.py:298)
        return ops.self_pool(news_embedding, self.user_query).unsqueeze(-2)


class TfmUserEncoder(BaseUserEncoder):
    history_keep_mode = 1      # nr_assemble_titles: his_mask != 0 (the same rule as history_keep below)
    pack_min_slots = 65536     # from_cache: history slots (B * L) from which the clicked-slots-only path is taken

    @staticmethod
    def history_keep(his_mask):
        """[B,L] 0/1: his_mask == 0 slots are masked keys of the layer and of the pooling (weight exactly 0, reference
        encoder.py:325-326 / attention.py:24,90) and their own rows are dropped by the pooling the same way."""
        return (his_mask != 0).long()

    def __init__(self, manager):
        super().__init__()
        self.transformer = TFMLayer(manager.hidden_dim, manager.head_num, 0.1)
        self.user_query, self.userProject = _QueryPooled.make(manager.hidden_dim)

    def forward(self, news_embedding, his_mask):
        h = self.transformer(news_embedding, attention_mask=his_mask)
        user, _ = ops.project_and_pool(h, self.userProject.weight, self.userProject.bias, self.user_query, his_mask)
        return user.unsqueeze(-2)

    def from_cache(self, cache, his_idx, his_mask):
        """inference from the news-vector cache (BaseModel.infer) over the CLICKED slots only: his_mask == 0 slots are
        masked keys of the layer and of the pooling (weight exactly 0) and their own rows are dropped by the pooling, so
        the user vector is bit-equal to forward(cache[his_idx], his_mask) while the layer runs on sum(his_mask) rows
        instead of B * L.  None when the mask is not prefix-shaped, gradients are wanted, or the batch is small (the
        plan's row count is one 8-byte device->host read: worth it for the 8 192-row evaluation batches, not for the
        reference's 200-row default, measured 511 k -> 445 k rows/s there) — the caller then takes forward()."""
        if torch.is_grad_enabled() or not cache.is_cuda or his_mask.shape[1] > 64 or his_mask.numel() < self.pack_min_slots:
            return None
        packed = ops.pack_titles(his_idx, his_mask, title_keep=None)
        if packed is None:
            return None
        idx_rows, plan = packed
        rows = ops.gather_rows(cache, idx_rows)                       # [rows, D]
        h = self.transformer(rows, attention_mask=his_mask, plan=plan)
        user, _ = ops.project_and_pool(h, self.userProject.weight, self.userProject.bias, self.user_query, his_mask, plan=plan)
        return user.unsqueeze(-2)
