"""ctypes binding of libnewsrec_hip.so (C ABI: include/newsrec_hip.h).

There is no fallback: if the library is missing or cannot be loaded, importing
the product's ops raises.  The library is built in-tree by
``__graft_entry__.build()`` (``make -C news-recommendation_amd/csrc``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnewsrec_hip.so")

c_fp = C.c_void_p      # device pointers travel as integers (tensor.data_ptr())
c_i64 = C.c_int64
c_u64 = C.c_uint64
c_u32 = C.c_uint32
c_f = C.c_float
c_int = C.c_int
c_sz = C.c_size_t


class NewsrecError(RuntimeError):
    pass


class TfmWeights(C.Structure):
    _fields_ = [(n, c_fp) for n in ("wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo", "ln1_g", "ln1_b",
                                    "w1", "b1", "w2", "b2", "ln2_g", "ln2_b")]


class TfmGrads(C.Structure):
    _fields_ = [(n, c_fp) for n in ("dwq", "dbq", "dwk", "dbk", "dwv", "dbv", "dwo", "dbo", "dln1_g", "dln1_b",
                                    "dw1", "db1", "dw2", "db2", "dln2_g", "dln2_b")]


class TfmSaved(C.Structure):
    _fields_ = [(n, c_fp) for n in ("wqkv", "bqkv", "qkv", "ctx", "s1", "y1", "z", "a", "s2", "stats",
                                    "row_max", "row_sum")]


class TfmSavedBf16(C.Structure):
    _fields_ = [(n, c_fp) for n in ("x_h", "qkv_h", "ctx_h", "y1_h", "z_h", "a_h", "s1", "y1", "s2", "stats", "row_max", "row_sum", "w_h", "wT_h")]


_SIGS = {
    "nr_version": (c_int, []),
    "nr_last_error": (C.c_char_p, []),
    "nr_prof_enable": (c_int, [c_int]),
    "nr_index_status": (c_int, [C.POINTER(c_int), c_fp]),
    "nr_set_gemm_compute": (c_int, [c_int]),
    "nr_set_side_stream": (c_int, [c_int]),
    "nr_get_gemm_compute": (c_int, []),
    "nr_assemble_titles": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_int, c_i64, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp]),
    "nr_query_pool_packed_bwd_split_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_int,
                                                   c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_prof_read": (c_int, [c_int, C.POINTER(C.c_double), C.POINTER(c_i64), C.POINTER(C.c_double)]),
    "nr_gather_rows_f32": (c_int, [c_fp, c_i64, c_i64, c_fp, c_i64, c_fp, c_fp, c_fp]),
    "nr_scatter_add_rows_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_fp, c_fp]),
    "nr_linear_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_int, c_fp, c_fp, c_fp]),
    "nr_linear_bwd_input_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_int, c_fp, c_fp, c_fp, c_fp]),
    "nr_linear_bwd_weight_ws_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "nr_linear_bwd_weight_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_conv1d_k3_pack": (c_int, [c_fp, c_i64, c_i64, c_fp, c_fp]),
    "nr_conv1d_k3_relu_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_fp, c_fp]),
    "nr_conv1d_k3_bwd_ws_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "nr_conv1d_k3_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_add_layernorm_fwd_f32": (c_int, [c_fp, c_fp, c_i64, c_fp, c_fp, c_i64, c_i64, c_f, c_f, c_u64, c_u32,
                                         c_fp, c_fp, c_fp, c_fp, c_fp]),
    "nr_layernorm_bwd_ws_bytes": (c_sz, [c_i64, c_i64]),
    "nr_add_layernorm_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_f, c_u64, c_u32,
                                         c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_noalloc_scratch_bytes": (c_sz, []),
    "nr_add_layernorm_bwd_noalloc_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_f, c_u64, c_u32,
                                                 c_fp, c_fp, c_fp, c_fp, c_fp]),
    "nr_mhsa_fwd_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_f, c_u64, c_u32, c_int, c_fp, c_fp, c_fp, c_fp]),
    "nr_mhsa_fwd_bf16": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_f, c_u64, c_u32, c_int, c_fp, c_fp, c_fp, c_fp]),
    "nr_mhsa_bwd_bf16": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_f, c_u64, c_u32, c_int, c_fp, c_fp]),
    "nr_mhsa_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_f, c_u64, c_u32, c_int,
                                c_fp, c_fp]),
    "nr_query_pool_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp]),
    "nr_query_pool_bwd_ws_bytes": (c_sz, [c_i64, c_i64]),
    "nr_query_pool_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_int, c_int,
                                      c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_seq_reduce_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_int, c_fp, c_fp]),
    "nr_seq_reduce_bwd_f32": (c_int, [c_fp, c_i64, c_i64, c_i64, c_int, c_fp, c_fp]),
    "nr_score_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp]),
    "nr_score_ce_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp]),
    "nr_score_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp]),
    "nr_gelu_bwd_f32": (c_int, [c_fp, c_fp, c_i64, c_fp, c_fp]),
    "nr_relu_bwd_f32": (c_int, [c_fp, c_fp, c_i64, c_fp, c_fp]),
    "nr_tfm_layer_fwd_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, C.POINTER(TfmWeights), c_f, c_u64, c_u32, c_int,
                                     C.POINTER(TfmSaved), c_fp, c_fp]),
    "nr_tfm_layer_bwd_ws_bytes": (c_sz, [c_i64, c_i64, c_i64]),
    "nr_tfm_layer_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, C.POINTER(TfmWeights), c_f, c_u64,
                                     c_u32, c_int, C.POINTER(TfmSaved), C.POINTER(TfmGrads), c_fp, c_fp, c_sz, c_fp]),
    "nr_pack_plan": (c_int, [c_fp, c_fp, c_i64, c_i64, c_fp, c_fp, c_fp]),
    "nr_pack_ids": (c_int, [c_fp, c_fp, c_i64, c_i64, c_fp, c_fp, c_fp]),
    "nr_tfm_layer_packed_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_i64, C.POINTER(TfmWeights), c_f, c_u64,
                                            c_u32, c_int, C.POINTER(TfmSaved), c_fp, c_fp]),
    "nr_tfm_layer_packed_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_i64, C.POINTER(TfmWeights), c_f,
                                            c_u64, c_u32, c_int, C.POINTER(TfmSaved), C.POINTER(TfmGrads), c_fp, c_fp, c_sz, c_fp]),
    "nr_query_pool_packed_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp]),
    "nr_query_pool_packed_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_int,
                                             c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_cast_bf16": (c_int, [c_fp, c_i64, c_fp, c_fp]),
    "nr_transpose_bf16": (c_int, [c_fp, c_int, c_i64, c_i64, c_i64, c_fp, c_i64, c_fp]),
    "nr_gemm_nt_bf16_ws_bytes": (c_sz, [c_i64, c_i64, c_i64, c_int]),
    "nr_gemm_nt_bf16": (c_int, [c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_int, c_fp, c_fp, c_fp, c_int, c_fp, c_sz, c_fp]),
    "nr_gemm_bf16": (c_int, [c_fp, c_int, c_i64, c_fp, c_int, c_i64, c_fp, c_i64, c_i64, c_i64, c_int, c_fp, c_fp, c_fp, c_int, c_fp, c_sz, c_fp]),
    "nr_tfm_layer_fwd_bf16": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_i64, C.POINTER(TfmWeights), c_f, c_u64, c_u32, c_int,
                                      C.POINTER(TfmSavedBf16), c_fp, c_fp]),
    "nr_tfm_layer_bwd_bf16_ws_bytes": (c_sz, [c_i64, c_i64]),
    "nr_tfm_layer_bwd_bf16": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_i64, C.POINTER(TfmWeights), c_f, c_u64, c_u32, c_int,
                                      C.POINTER(TfmSavedBf16), C.POINTER(TfmGrads), c_fp, c_fp, c_sz, c_fp]),
    "nr_add_layernorm_packed_fwd_f32": (c_int, [c_fp, c_fp, c_i64, c_fp, c_fp, c_fp, c_i64, c_i64, c_f, c_f, c_u64, c_u32,
                                                c_fp, c_fp, c_fp, c_fp, c_fp]),
    "nr_add_layernorm_packed_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_f, c_u64, c_u32,
                                                c_fp, c_fp, c_fp, c_fp, c_fp, c_sz, c_fp]),
    "nr_dropout_packed_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_f, c_u64, c_u32, c_fp, c_fp]),
    "nr_unpack_rows_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_i64, c_fp, c_fp]),
    "nr_gru_fwd_ws_bytes": (c_sz, [c_i64, c_i64]),
    "nr_gru_fwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                               c_sz, c_fp]),
    "nr_gru_bwd_ws_bytes": (c_sz, [c_i64, c_i64, c_i64, c_i64]),
    "nr_gru_bwd_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, c_i64, c_i64, c_i64, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp, c_fp,
                               c_fp, c_sz, c_fp]),
    "nr_gru_status": (c_int, [c_fp, c_fp]),
    "nr_gru_debug_set_spin_limit": (c_int, [c_u32]),
    "nr_impression_metrics_f64": (c_int, [c_fp, c_fp, c_fp, c_i64, c_fp, c_fp, c_i64, c_int, C.POINTER(c_int), c_fp, c_fp, c_fp]),
    "nr_dropout_f32": (c_int, [c_fp, c_i64, c_f, c_u64, c_u32, c_fp, c_fp]),
    "nr_add_bias_rows_f32": (c_int, [c_fp, c_fp, c_i64, c_i64, c_fp, c_fp]),
    "nr_colsum_ws_bytes": (c_sz, [c_i64, c_i64]),
    "nr_colsum_f32": (c_int, [c_fp, c_i64, c_i64, c_fp, c_fp, c_sz, c_fp]),
    "nr_adam_step_f32": (c_int, [c_fp, c_fp, c_fp, c_fp, c_i64, C.c_double, C.c_double, C.c_double, C.c_double, c_i64, c_fp]),
    "nr_adam_multi_f32": (c_int, [c_int, C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_fp), C.POINTER(c_i64),
                                  C.c_double, C.c_double, C.c_double, C.c_double, c_i64, c_fp]),
}

EXPORTS = tuple(_SIGS)
_lib = None


def load():
    """dlopen the HIP library (once) and type its entry points; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NewsrecError(
            f"{LIB_PATH} not found: the MI355X HIP library is the only compute path of this package "
            "(no CPU/torch fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C news-recommendation_amd/csrc`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError here == ABI drift; let it surface
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().nr_last_error()
        raise NewsrecError(f"{what} failed (status {rc}): {msg.decode() if msg else '?'}")
