// internal interface of the per-sequence kernels (attention.hip, seqops.hip)
#pragma once
#include "common.h"
#include <algorithm>
#include "rowops.h"

namespace nr {

int mhsa_fwd(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
             int dh, DropoutCfg dc, hipStream_t s);
int mhsa_bwd(const float* qkv, const int64_t* mask, const float* ctx, const float* dctx, const float* row_max,
             const float* row_sum, float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s);

// title-length fast path on the f32 matrix cores (attention_mfma.hip): S <= 32, dh <= 64
bool mhsa_mfma_supported(int S, int dh);
int mhsa_fwd_mfma(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
                  int dh, DropoutCfg dc, hipStream_t s);
int mhsa_bwd_mfma(const float* qkv, const int64_t* mask, const float* dctx, const float* row_max, const float* row_sum,
                  float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s);
// 32 < S <= 64 (two 32-row tiles per head, two waves per head); f32 tensors, sequences not packed
bool mhsa_mfma2_supported(int S, int dh);
int mhsa_fwd_mfma2(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
                   int dh, DropoutCfg dc, hipStream_t s);
int mhsa_bwd_mfma2(const float* qkv, const int64_t* mask, const float* dctx, const float* row_max, const float* row_sum,
                   float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s);

// the same kernels over bf16 tensors in HBM (the bf16 build); tiles and arithmetic stay f32
int mhsa_fwd_mfma_h(const unsigned short* qkv, const int64_t* mask, unsigned short* ctx, float* row_max, float* row_sum, int64_t N, int S,
                    int H, int dh, DropoutCfg dc, hipStream_t s);
int mhsa_bwd_mfma_h(const unsigned short* qkv, const int64_t* mask, const unsigned short* dctx, const float* row_max, const float* row_sum,
                    unsigned short* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s);

int pool_fwd(const float* key, const float* value, const float* table, const int64_t* idx, int64_t nrows, const float* query,
             const int64_t* mask, float* out, float* prob, int64_t nseq, int S, int D, hipStream_t s,
             const int* offs = nullptr, int64_t rows = 0);       // offs: packed sequences (rows = their total)
int pool_bwd(const float* key, const float* value, const float* query, const float* prob, const float* dout, float* dkeypre,
             float* dvalue, float* dq_part, int64_t nseq, int S, int D, int tanh_key, int key_is_value, hipStream_t s,
             const int* offs = nullptr, int64_t rows = 0, const float* dout1 = nullptr, int64_t split = 0);   // dout1: gradient of sequences split ..
int score_fwd(const float* cdd, const float* table, const int64_t* idx, int64_t nrows, const float* user, const int64_t* label,
              float* logits, float* loss_b, int64_t B, int C, int D, hipStream_t s);
int score_bwd(const float* cdd, const float* user, const int64_t* label, const float* logits, const float* grad_out,
              const float* dlogits, float* d_cdd, float* d_user, int64_t B, int C, int D, hipStream_t s);
int seq_reduce(const float* x, const float* table, const int64_t* idx, int64_t nrows, float* out, int64_t nseq, int S, int D,
               int mean, hipStream_t s);
int seq_broadcast(const float* dout, float* dx, int64_t nseq, int S, int D, float scale, hipStream_t s);


// GRU recurrence (gru.hip): one persistent launch for the whole forward, or the per-step halves
bool gru_seq_fwd_supported(int B, int H);
bool gru_seq_bwd_supported(int B, int H);
void gru_set_spin_limit(unsigned polls);
int gru_seq_bwd(const float* dh_out, const float* gates, const float* hall, const int* lens, const float* w_hh, float* dgi, float* dgh,
                int B, int L, int H, unsigned* counter, hipStream_t s);
int gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, const int* lens, float* hall, float* gates, int B, int L, int H,
                unsigned* counter, hipStream_t s);
int gru_lens(const int64_t* mask, int* lens, int B, int L, hipStream_t s);
int gru_gate_fwd(const float* gi, const float* gh, const float* hprev, const int* lens, float* hnext, float* saved, int B, int L,
                 int H, int t, hipStream_t s);
int gru_gate_bwd(const float* dh, const float* saved, const float* hprev, const int* lens, float* dgi, float* dgh, float* dh_direct,
                 int B, int L, int H, int t, hipStream_t s);


// per-impression AUC / MRR / nDCG@k on the device (metrics.hip)
int impression_metrics(const float* preds, const int64_t* labels, const uint8_t* masks, int Ce, const int64_t* row_start,
                       const int32_t* row_count, int64_t G, int nk, const int* ks, double* out, int32_t* flags, hipStream_t s);

}  // namespace nr
