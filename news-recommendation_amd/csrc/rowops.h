// internal interface of the HBM-bound row kernels (rowops.hip)
#pragma once
#include "common.h"
#include <algorithm>
#include <math.h>

namespace nr {

struct DropoutCfg {
    float p = 0.f;        // drop probability; 0 => identity (eval mode)
    uint64_t seed = 0;    // philox key
    uint32_t site = 0;    // stream id of this dropout site
    // attention kernels only: additive term of a masked key.  -1e5 = the reference's TFMLayer (attention.py:44-49);
    // -INFINITY = HF BertModel under torch SDPA (masked keys excluded, a sequence with no valid key gets zero
    // probabilities = zero context: torch's safe softmax)
    float mask_fill = -1e5f;
    // packed title tokens (rows of padded positions removed, see nr_pack_plan): dropout keep-bits stay indexed by the
    // ORIGINAL position, so results equal the padded layout's.
    const int* row_map = nullptr;   // LayerNorm kernels: packed row -> original row (n*S + i); NULL = identity
    const int* offs = nullptr;      // attention kernels: sequence n owns packed rows offs[n] .. offs[n+1]; NULL = dense [N,S]
};

// the library's index flag (one int per GPU, in a file-scope __device__ variable): raised by every kernel that clamps a
// row index into its table instead of faulting (gather with no caller flag, the *_from_table forms of pooling / scoring /
// sequence sums); read and cleared by nr_index_status.  nullptr if its address cannot be resolved (kernels then only clamp).
int* index_flag_ptr();
int index_status(int* host_out, hipStream_t s);
int gather_rows(const float* table, int64_t V, int E, const int64_t* ids, int64_t n, float* out, int* oob_flag, hipStream_t s);
int scatter_add_rows(const float* grad_rows, const int64_t* ids, int64_t n, int E, int64_t V, int64_t padding_idx,
                     float* grad_table, hipStream_t s);
int ln_fwd(const float* h, const float* x, const float* gamma, const float* beta, float* y, float* s_out, float* mean,
           float* rstd, int64_t R, int D, float eps, DropoutCfg dc, hipStream_t s, int64_t x_period = 0,
           unsigned short* y_h = nullptr);      // y_h: a bf16 copy of y written alongside (the bf16 build's GEMM operand)
int ln_bwd_blocks(int64_t R);
int ln_bwd(const float* dy, const float* sin, const float* mean, const float* rstd, const float* gamma, float* ds,
           float* dh, float* dgamma_part, float* dbeta_part, int64_t R, int D, DropoutCfg dc, hipStream_t s,
           unsigned short* dh_h = nullptr);     // dh_h: a bf16 copy of the dropped-branch gradient (dh, or ds when p == 0)
int colsum_chunks(int64_t R);
int assemble_titles(const int64_t* cdd_ids, const int64_t* his_ids, const int64_t* cdd_msk, const int64_t* his_msk, const int64_t* his_mask,
                    int keep_mode, int B, int C, int L, int T, int64_t* ids, int64_t* msk, int64_t* keep, hipStream_t s);
int pack_plan(const int64_t* mask, const int64_t* title_keep, int64_t N, int T, int* offs, int* info, hipStream_t s);
int pack_ids(const int64_t* ids, const int* offs, int64_t N, int T, int64_t* ids_packed, int* row_map, hipStream_t s);
int colsum(const float* in, int64_t ld, int64_t R, int N, float* out, float* part, hipStream_t s);
int colsum2(const float* a, const float* b, int64_t R, int N, float* out_a, float* out_b, float* part, hipStream_t s);
int colsum_h(const unsigned short* in, int64_t ld, int64_t R, int N, float* out, float* part, hipStream_t s);
int cast_bf16(const float* x, unsigned short* y, int64_t n, hipStream_t s);
int transpose_bf16(const void* src, int src_is_bf16, int64_t ld, int64_t R, int64_t Cn, unsigned short* dst, int64_t Rp, hipStream_t s,
                   unsigned short* straight = nullptr);      // straight: also the plain bf16 copy [R, Cn] (f32 sources)
int conv_pack(const float* w, float* wf, float* wb, int D, int E, hipStream_t s);
int conv_unpack_grad(const float* gwf, float* gw, int D, int E, hipStream_t s);
int add_f32(const float* a, const float* b, float* o, int64_t n, hipStream_t s);
int relu_bwd_f32(const float* dy, const float* y, float* o, int64_t n, hipStream_t s);
int gelu_bwd_f32(const float* dy, const float* z, float* o, int64_t n, hipStream_t s);
int mean_f32(const float* v, int n, float* out, hipStream_t s);
int dropout_f32(const float* x, float* o, int64_t n, DropoutCfg dc, hipStream_t s, int D = 0);
int unpack_rows(const float* src, const int* row_map, int64_t R, int D, int64_t Rfull, float* dst, hipStream_t s);
int add_bias_rows_f32(const float* x, const float* b, float* o, int64_t R, int D, hipStream_t s);
int adam_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps, int64_t step,
             hipStream_t s);
constexpr int ADAM_MAX_TENSORS = 48;   // tensors per launch (the table travels in the kernel arguments)
int adam_multi_f32(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                   double lr, double b1, double b2, double eps, int64_t step, hipStream_t s);

}  // namespace nr
