// Per-impression ranking metrics on the device: AUC, MRR and nDCG@k of every impression of a
// cached-evaluation run, from the flat result tensors the scoring pass leaves in HBM — replaces the
// reference's .tolist() / all_gather_object / pack_results / python metric loops
// (models/BaseModel.py:254-287, utils/util.py:37-57,126-158,197-250) for the common case.
//
// An impression = `row_count[g]` consecutive result rows (<= Ce-candidate chunks) starting at row_start[g]; its
// candidates are the mask-selected entries in row order (pack_results' order).  One wave per impression:
// compact (score, label) into LDS, then rank by counting — rank_i = #{s_j > s_i} + #{s_j == s_i, j > i}, the
// position of i in np.argsort(score)[::-1] when the sort is stable.  numpy's default sort is NOT stable beyond
// 16 elements, so an impression with an exact score tie is only FLAGGED here (bit 0) and the caller recomputes
// its MRR / nDCG with the reference's own numpy expression; AUC (ties count 1/2) does not depend on tie order.
//   mrr   = sum_i y_i / (rank_i + 1) / sum_i y_i                                   (util.py:137-143)
//   dcg@k = sum_{rank_i < k} (2^y_i - 1) / log2(rank_i + 2),  ndcg = dcg / ideal    (util.py:146-194)
//   auc   = (#{pos > neg} + 0.5 #{pos == neg}) / (npos * nneg)                      (sklearn roc_auc_score, util.py:205)
// Sums are formed in double like numpy's.  Flags: 1 tie, 2 more than CAP candidates, 4 a label outside {0,1},
// 8 only one class present (the reference raises there) — any of 2/4/8 means "not computed".
#include "common.h"
#include "seqops.h"

namespace nr {

constexpr int MET_CAP = 1024;          // candidates per impression held in LDS
constexpr int MET_WPB = 4;

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void impression_metrics_kernel(const float* __restrict__ preds, const int64_t* __restrict__ labels,
                                                                 const uint8_t* __restrict__ masks, int Ce,
                                                                 const int64_t* __restrict__ row_start,
                                                                 const int32_t* __restrict__ row_count, int64_t G, int nk, int k0,
                                                                 int k1, int k2, int k3, double* __restrict__ out,
                                                                 int32_t* __restrict__ flags) {
    __shared__ float sc[MET_WPB][MET_CAP];
    __shared__ signed char lb[MET_WPB][MET_CAP];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t g = (int64_t)blockIdx.x * MET_WPB + w;
    if (g >= G) return;
    const int64_t r0 = row_start[g];
    const int slots = row_count[g] * Ce;
    // ---- ordered compaction of the valid candidates
    int n = 0, bad = 0;
    for (int base = 0; base < slots; base += 64) {
        const int s = base + lane;
        const bool ok = s < slots && masks[r0 * Ce + s] != 0;
        const unsigned long long bal = __ballot(ok);
        const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
        if (ok) {
            if (pos < MET_CAP) {
                const int64_t y = labels[r0 * Ce + s];
                sc[w][pos] = preds[r0 * Ce + s];
                lb[w][pos] = (signed char)(y == 1 ? 1 : 0);
                if (y != 0 && y != 1) bad |= 4;
            } else bad |= 2;
        }
        n += __popcll(bal);
    }
    bad = __any(bad & 4) ? 4 : 0;
    if (n > MET_CAP) bad |= 2;
    const int nko = 2 + nk;
    if (bad) {
        if (lane == 0) { flags[g] = bad; for (int c = 0; c < nko; ++c) out[g * nko + c] = 0.0; }
        return;
    }
    __builtin_amdgcn_wave_barrier();
    const int ks[4] = {k0, k1, k2, k3};
    double mrr = 0.0, aucn = 0.0, dcg[4] = {0.0, 0.0, 0.0, 0.0};
    int npos = 0, tie = 0;
    for (int i = lane; i < n; i += 64) {
        const float si = sc[w][i];
        const int yi = lb[w][i];
        int gt = 0, eq_after = 0, eq_any = 0, gt_neg = 0, eq_neg = 0;
        for (int j = 0; j < n; ++j) {
            const float sj = sc[w][j];
            const int yj = lb[w][j];
            gt += sj > si;
            const int e = (sj == si) && (j != i);
            eq_any += e;
            eq_after += e && (j > i);
            gt_neg += (yj == 0) && (si > sj);
            eq_neg += (yj == 0) && (si == sj);
        }
        tie |= eq_any > 0;
        if (yi) {
            const int rank = gt + eq_after;
            npos += 1;
            mrr += 1.0 / (double)(rank + 1);
            const double gain = 1.0 / log2((double)(rank + 2));          // (2^1 - 1) / log2(rank + 2)
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < nk && rank < ks[c]) dcg[c] += gain;
            aucn += (double)gt_neg + 0.5 * (double)eq_neg;
        }
    }
    mrr = wave_sum_d(mrr);
    aucn = wave_sum_d(aucn);
#pragma unroll
    for (int c = 0; c < 4; ++c) dcg[c] = wave_sum_d(dcg[c]);
    int np_all = npos;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) np_all += __shfl_xor(np_all, o, 64);
    tie = __any(tie) ? 1 : 0;
    if (lane == 0) {
        const int nneg = n - np_all;
        int fl = tie;
        if (np_all == 0 || nneg == 0) fl |= 8;
        flags[g] = fl;
        double* o = out + g * nko;
        o[0] = (fl & 8) ? 0.0 : aucn / ((double)np_all * (double)nneg);
        o[1] = np_all ? mrr / (double)np_all : 0.0;
        for (int c = 0; c < nk; ++c) {
            double ideal = 0.0;
            const int m = min(min(ks[c], n), np_all);
            for (int r = 0; r < m; ++r) ideal += 1.0 / log2((double)(r + 2));
            o[2 + c] = ideal > 0.0 ? dcg[c] / ideal : 0.0;
        }
    }
}

int impression_metrics(const float* preds, const int64_t* labels, const uint8_t* masks, int Ce, const int64_t* row_start,
                       const int32_t* row_count, int64_t G, int nk, const int* ks, double* out, int32_t* flags, hipStream_t s) {
    if (G == 0) return NR_OK;
    int k[4] = {0, 0, 0, 0};
    for (int c = 0; c < nk; ++c) k[c] = ks[c];
    hipLaunchKernelGGL(impression_metrics_kernel, dim3((unsigned)cdiv(G, MET_WPB)), dim3(256), 0, s, preds, labels, masks, Ce,
                       row_start, row_count, G, nk, k[0], k[1], k[2], k[3], out, flags);
    return check_launch("impression_metrics");
}

}  // namespace nr
