// Elementwise halves of the GRU recurrence (RnnUserEncoder, reference encoder.py:204-233: one-layer
// nn.GRU(D, D, batch_first) over the clicked-news vectors, packed by len_b = sum(his_mask[b]); the user
// vector is the state after len_b steps).  The matrix halves — GI = X W_ih^T + b_ih for all steps at once,
// GH_t = h_{t-1} W_hh^T + b_hh per step, and the weight / input gradients — run on the f32-MFMA GEMM
// (gemm.hip); the driver loop is in capi.hip (nr_gru_fwd_f32 / nr_gru_bwd_f32).
//
//   r = sigmoid(gi_r + gh_r)   z = sigmoid(gi_z + gh_z)   n = tanh(gi_n + r * gh_n)   h' = (1-z) n + z h
// torch gate order in weight_ih / weight_hh rows: r | z | n.   Sequences shorter than L freeze their state.
#include "common.h"
#include "seqops.h"

namespace nr {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// lens[b] = sum_l mask[b, l]
__global__ void gru_lens_kernel(const int64_t* __restrict__ mask, int* __restrict__ lens, int B, int L) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int l = 0; l < L; ++l) n += mask[(int64_t)b * L + l] != 0;
    lens[b] = n;
}

// one time step, all sequences: gi rows are (b*L + t), gh [B,3H]; saves (r, z, n, gh_n) for the backward
__global__ __launch_bounds__(256) void gru_gate_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ gh,
                                                           const float* __restrict__ hprev, const int* __restrict__ lens,
                                                           float* __restrict__ hnext, float* __restrict__ saved, int B, int L,
                                                           int H, int t) {
    const int64_t total = (int64_t)B * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), j = (int)(i - (int64_t)b * H);
        const float hp = hprev[i];
        if (t >= lens[b]) { hnext[i] = hp; continue; }             // packed sequence ended: state frozen
        const float* gib = gi + ((int64_t)b * L + t) * 3 * H;
        const float* ghb = gh + (int64_t)b * 3 * H;
        const float r = sigmoid_f(gib[j] + ghb[j]);
        const float z = sigmoid_f(gib[H + j] + ghb[H + j]);
        const float ghn = ghb[2 * H + j];
        const float n = tanhf(gib[2 * H + j] + r * ghn);
        hnext[i] = (1.0f - z) * n + z * hp;
        if (saved) {
            float* sv = saved + (int64_t)b * 4 * H;
            sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = ghn;
        }
    }
}

// backward of one step: dh = d(loss)/d(h_t) -> dgi rows (b*L + t), dgh rows (t*B + b), dh_direct = dh * z
// (the W_hh path is added by the following GEMM: dh_{t-1} = dh_direct + dgh W_hh)
__global__ __launch_bounds__(256) void gru_gate_bwd_kernel(const float* __restrict__ dh, const float* __restrict__ saved,
                                                           const float* __restrict__ hprev, const int* __restrict__ lens,
                                                           float* __restrict__ dgi, float* __restrict__ dgh,
                                                           float* __restrict__ dh_direct, int B, int L, int H, int t) {
    const int64_t total = (int64_t)B * H;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / H), j = (int)(i - (int64_t)b * H);
        float* gib = dgi + ((int64_t)b * L + t) * 3 * H;
        float* ghb = dgh + (int64_t)b * 3 * H;
        const float g = dh[i];
        if (t >= lens[b]) {
            gib[j] = 0.f; gib[H + j] = 0.f; gib[2 * H + j] = 0.f;
            ghb[j] = 0.f; ghb[H + j] = 0.f; ghb[2 * H + j] = 0.f;
            dh_direct[i] = g;
            continue;
        }
        const float* sv = saved + (int64_t)b * 4 * H;
        const float r = sv[j], z = sv[H + j], n = sv[2 * H + j], ghn = sv[3 * H + j];
        const float hp = hprev[i];
        const float dn_pre = g * (1.0f - z) * (1.0f - n * n);
        const float dz_pre = g * (hp - n) * z * (1.0f - z);
        const float dr_pre = dn_pre * ghn * r * (1.0f - r);
        gib[j] = dr_pre; gib[H + j] = dz_pre; gib[2 * H + j] = dn_pre;
        ghb[j] = dr_pre; ghb[H + j] = dz_pre; ghb[2 * H + j] = dn_pre * r;
        dh_direct[i] = g * z;
    }
}

int gru_lens(const int64_t* mask, int* lens, int B, int L, hipStream_t s) {
    hipLaunchKernelGGL(gru_lens_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, mask, lens, B, L);
    return check_launch("gru_lens");
}
int gru_gate_fwd(const float* gi, const float* gh, const float* hprev, const int* lens, float* hnext, float* saved, int B, int L,
                 int H, int t, hipStream_t s) {
    const int grid = (int)std::min<int64_t>(cdiv((int64_t)B * H, 256), 4096);
    hipLaunchKernelGGL(gru_gate_fwd_kernel, dim3(grid), dim3(256), 0, s, gi, gh, hprev, lens, hnext, saved, B, L, H, t);
    return check_launch("gru_gate_fwd");
}
int gru_gate_bwd(const float* dh, const float* saved, const float* hprev, const int* lens, float* dgi, float* dgh, float* dh_direct,
                 int B, int L, int H, int t, hipStream_t s) {
    const int grid = (int)std::min<int64_t>(cdiv((int64_t)B * H, 256), 4096);
    hipLaunchKernelGGL(gru_gate_bwd_kernel, dim3(grid), dim3(256), 0, s, dh, saved, hprev, lens, dgi, dgh, dh_direct, B, L, H, t);
    return check_launch("gru_gate_bwd");
}

}  // namespace nr
