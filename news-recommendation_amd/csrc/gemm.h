// internal GEMM interface (see gemm.hip)
#pragma once
#include "common.h"
#include <algorithm>

namespace nr {

enum Epilogue {
    EPI_NONE = 0,              // y = acc + bias
    EPI_RELU = 1,              // y = relu(acc + bias)
    EPI_GELU = 2,              // out2 = z = acc + bias ; y = gelu(z)
    EPI_TANH = 3,              // y = tanh(acc + bias)
    EPI_ADD_AUX = 4,           // y = acc + bias + aux
    EPI_MUL_GELU_GRAD = 5,     // y = (acc + bias) * gelu'(aux)
    EPI_ADD_AUX_RELUMASK = 6,  // y = (acc + bias + aux) * (aux2 > 0)
    EPI_RELUMASK = 7,          // y = (acc + bias) * (aux2 > 0)
};

struct GemmParams {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0;
    int64_t lda = 0, ldb = 0, ldc = 0;
    const float* bias = nullptr;
    const float* aux = nullptr;
    const float* aux2 = nullptr;
    float* out2 = nullptr;
    int64_t ldaux = 0, ldaux2 = 0, ldout2 = 0;
    int epi = EPI_NONE;
    int conv_on = 0;           // 0 none, 1: A is an im2col view (k-contig), 2: B is one (reduction-major)
    int conv_T = 0, conv_E = 0;
    int ksplit = 1;
    int64_t split_stride = 0;
    // reduction-major A only (dW = dY^T X): also emit csum[m] = sum_k A(m,k) — the bias gradient — from the
    // tiles already in LDS; csum_part [ksplit, M] is scratch for the split-K case
    float* csum = nullptr;
    float* csum_part = nullptr;
};

// bf16-in-HBM "NT" GEMM (gemm_h.hip): C[m,n] = epi(sum_k A[m,k] B[n,k] + bias[n]); A [M,K], B [N,K] bf16 k-contiguous
// (raw 16-bit storage), f32 accumulate.  Outputs: C (f32, may be NULL) and / or Ch (bf16, may be NULL).
struct GemmHParams {
    const unsigned short* A = nullptr;
    const unsigned short* B = nullptr;
    int M = 0, N = 0, K = 0;                     // K a multiple of 64 unless both operands are reduction-major
    int64_t lda = 0, ldb = 0;
    int a_rm = 0, b_rm = 0;                      // reduction-major operand: A(m,k) = A[k*lda + m] / B(n,k) = B[k*ldb + n] (M / N % 8 == 0)
    const unsigned short* zero = nullptr;        // 256 bytes of zeros (set by gemm_h): what reduction rows past K read
    float* C = nullptr;
    int64_t ldc = 0;
    unsigned short* Ch = nullptr;
    int64_t ldch = 0;
    unsigned short* ChT = nullptr;               // the same result transposed, bf16 [N, ldcht] with ldcht % 64 == 0, >= M; columns M .. are zero
    int64_t ldcht = 0;
    const float* bias = nullptr;
    int epi = EPI_NONE;                          // NONE | RELU | GELU (+ out2h = pre-activation) | TANH | ADD_AUX (aux f32) | MUL_GELU_GRAD (auxh bf16)
    const float* aux = nullptr;
    int64_t ldaux = 0;
    const unsigned short* auxh = nullptr;
    int64_t ldauxh = 0;
    unsigned short* out2h = nullptr;
    int64_t ldout2h = 0;
    int ksplit = 1;                              // > 1: plain f32 C only, summed from slabs in the workspace
    int64_t split_stride = 0;
    int vec = 0;                                 // set by gemm_h(): every epilogue pointer / pitch allows 16-byte accesses
};
int gemm_h(GemmHParams p, float* workspace, size_t ws_bytes, hipStream_t s);
int choose_ksplit_h(int M, int N, int K);
size_t gemm_h_workspace_bytes(int M, int N, int ksplit);
bool gemm_h_supported(int M, int N, int K);

int gemm_f32(GemmParams p, bool a_kc, bool b_kc, float* workspace, size_t ws_bytes, hipStream_t s);
size_t gemm_workspace_bytes(int M, int N, int ksplit);
int choose_ksplit(int M, int N, int K);
bool gemm_fast_path(const GemmParams& p, bool a_kc, bool b_kc);
// 0: exact fp32 matrix cores (default); 1: operands rounded to bf16 in LDS, f32 accumulate (large non-conv products only)
void set_gemm_compute(int mode);
int get_gemm_compute();

}  // namespace nr
