// Multi-head self-attention over short sequences (titles S=32, histories S=50, title+abstract S=96; S <= 128),
// one wave (two for S > 64) per (sequence, head).  The S x S score tile is tiny (<= 16 KiB), so the whole
// head lives in LDS/registers: lane i owns query row i (fwd, bwd phase A) or key row j
// (bwd phase B).  Score arithmetic follows the reference literally:
//   x_ij = fl32( fl32(q_i . k_j) / sqrt(dh) ) + (1 - mask_j) * (-1e5)      (attention.py:87-90)
//   p_ij = exp(x_ij - max_j x_ij) / sum_j exp(...)
// so fully masked rows reproduce the 2^-7 quantisation of `score - 1e5`.
// ~1-2 % of the layer's flops: plain VALU fma (the f32 MFMA rate equals the f32 VALU rate).
#include "common.h"
#include "seqops.h"

namespace nr {


// cooperative [S, dh] tile copy global(row stride ld) -> LDS rows of DHP+1 floats (zero padded)
// LDS row stride LDS_LD: DHP+1 (odd: conflict-free when lane i reads row i) or DHP (16-B aligned
// rows for broadcast float4 reads of K/V)
template <int DHP, int LDS_LD, int NT = 64>
__device__ __forceinline__ void tile_to_lds(float* lds, const float* __restrict__ g, int64_t ld, int S, int dh, int lane) {
    // eight loads in flight per lane before the first LDS store: one load per round trip made the copy the
    // longest part of a (sequence, head) at S = 50 (a single wave has nothing else to hide the latency behind)
    constexpr int UN = 8;
    const int total = S * DHP;
    for (int base = lane; base < total; base += NT * UN) {
        float v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int idx = base + u * NT;
            const int r = idx / DHP, d = idx - r * DHP;
            v[u] = (idx < total && d < dh) ? g[(int64_t)r * ld + d] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int idx = base + u * NT;
            const int r = idx / DHP, d = idx - r * DHP;
            if (idx < total) lds[r * LDS_LD + d] = v[u];
        }
    }
}
template <int DHP, int LDS_LD, int NT = 64>
__device__ __forceinline__ void lds_to_tile(float* __restrict__ g, int64_t ld, const float* lds, int S, int dh, int lane) {
    for (int idx = lane; idx < S * dh; idx += NT) {
        const int r = idx / dh, d = idx - r * dh;
        g[(int64_t)r * ld + d] = lds[r * LDS_LD + d];
    }
}
// dot of a register row with a broadcast LDS row (16-B aligned)
template <int DHP>
__device__ __forceinline__ float dot_bcast(const float (&q)[DHP], const float* row, float acc) {
#pragma unroll
    for (int d = 0; d < DHP; d += 4) {
        const float4 k = *reinterpret_cast<const float4*>(row + d);
        acc = fmaf(q[d], k.x, acc); acc = fmaf(q[d + 1], k.y, acc);
        acc = fmaf(q[d + 2], k.z, acc); acc = fmaf(q[d + 3], k.w, acc);
    }
    return acc;
}
template <int DHP>
__device__ __forceinline__ void axpy_bcast(float (&o)[DHP], float a, const float* row) {
#pragma unroll
    for (int d = 0; d < DHP; d += 4) {
        const float4 v = *reinterpret_cast<const float4*>(row + d);
        o[d] = fmaf(a, v.x, o[d]); o[d + 1] = fmaf(a, v.y, o[d + 1]);
        o[d + 2] = fmaf(a, v.z, o[d + 2]); o[d + 3] = fmaf(a, v.w, o[d + 3]);
    }
}

__device__ __forceinline__ float keep_scale(const DropoutCfg& dc, uint32_t thr, float scale, int64_t row_linear, int S, int j) {
    // dropout on attention probabilities: element index = row_linear * S + j
    const uint64_t idx = (uint64_t)(row_linear * S + j);
    const Philox4 rn = philox4x32_10(dc.seed, idx >> 2, dc.site);
    return philox_pick(rn, (int)(idx & 3)) >= thr ? scale : 0.f;
}

template <int DHP, int NT>
__global__ __launch_bounds__(NT) void mhsa_fwd_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                      float* __restrict__ ctx, float* __restrict__ row_max,
                                                      float* __restrict__ row_sum, int S, int H, int dh, DropoutCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = DHP + 1;
    float* Ks = smem;                 // [S][DHP]  broadcast reads
    float* Vs = Ks + S * DHP;
    float* Qs = Vs + S * DHP;         // [S][LD]   (reused for the output tile)
    float* Ss = Qs + S * LD;          // [S][S+1]
    float* Ms = Ss + S * (S + 1);     // [S] additive key mask (0 or mask_fill): one global read per key, not one per (query, key)
    const int lane = threadIdx.x;
    const int n = blockIdx.x / H, h = blockIdx.x - n * H;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const float* base = qkv + (int64_t)n * S * ld + h * dh;
    if (lane < S) Ms[lane] = mask[(int64_t)n * S + lane] ? 0.f : dc.mask_fill;
    tile_to_lds<DHP, LD, NT>(Qs, base, ld, S, dh, lane);
    tile_to_lds<DHP, DHP, NT>(Ks, base + A, ld, S, dh, lane);
    tile_to_lds<DHP, DHP, NT>(Vs, base + 2 * A, ld, S, dh, lane);
    __syncthreads();

    const int i = lane;
    const bool live = i < S;
    float q[DHP];
#pragma unroll
    for (int d = 0; d < DHP; ++d) q[d] = live ? Qs[i * LD + d] : 0.f;
    const float sq = sqrtf((float)dh);
    float* srow = Ss + i * (S + 1);
    float m = -INFINITY;
    if (live) {
        for (int j = 0; j < S; ++j) {
            const float acc = dot_bcast<DHP>(q, Ks + j * DHP, 0.f);
            float x = acc / sq;
            x += Ms[j];
            srow[j] = x;
            m = fmaxf(m, x);
        }
        if (m == -INFINITY) m = 0.f;             // no valid key (mask_fill = -inf): every exp() below is 0
        float sum = 0.f;
        for (int j = 0; j < S; ++j) {
            const float e = expf(srow[j] - m);
            srow[j] = e;
            sum += e;
        }
        if (sum == 0.f) sum = 1.f;               // ... and the probabilities are 0, not 0/0
        if (row_max) {
            row_max[((int64_t)n * H + h) * S + i] = m;
            row_sum[((int64_t)n * H + h) * S + i] = sum;
        }
        const uint32_t thr = drop_threshold(dc.p);
        const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
        const int64_t row_linear = ((int64_t)n * H + h) * S + i;
        float o[DHP];
#pragma unroll
        for (int d = 0; d < DHP; ++d) o[d] = 0.f;
        for (int j = 0; j < S; ++j) {
            float p = srow[j] / sum;
            if (dc.p > 0.f) p *= keep_scale(dc, thr, scale, row_linear, S, j);
            axpy_bcast<DHP>(o, p, Vs + j * DHP);
        }
#pragma unroll
        for (int d = 0; d < DHP; ++d) Qs[i * LD + d] = o[d];   // row i of Qs is only read by lane i
    }
    __syncthreads();
    lds_to_tile<DHP, LD, NT>(ctx + (int64_t)n * S * A + h * dh, A, Qs, S, dh, lane);
}

// backward: dqkv[:, q|k|v block of head h] from dctx; recomputes p from (qkv, row_max, row_sum)
template <int DHP, int NT>
__global__ __launch_bounds__(NT) void mhsa_bwd_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                      const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                      const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                      float* __restrict__ dqkv, int S, int H, int dh, DropoutCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = DHP + 1;
    float* Ks = smem;                 // [S][DHP] broadcast reads
    float* Vs = Ks + S * DHP;
    float* Qs = Vs + S * DHP;         // [S][LD]
    float* Gs = Qs + S * LD;          // dO tile
    float* Ts = Gs + S * LD;          // staging tile for dq
    float* Ps = Ts + S * LD;          // [S][S+1] dropped probabilities
    float* Ds = Ps + S * (S + 1);     // [S][S+1] dS
    float* Ms = Ds + S * (S + 1);     // [S] additive key mask
    const int lane = threadIdx.x;
    const int n = blockIdx.x / H, h = blockIdx.x - n * H;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const float* base = qkv + (int64_t)n * S * ld + h * dh;
    tile_to_lds<DHP, LD, NT>(Qs, base, ld, S, dh, lane);
    tile_to_lds<DHP, DHP, NT>(Ks, base + A, ld, S, dh, lane);
    tile_to_lds<DHP, DHP, NT>(Vs, base + 2 * A, ld, S, dh, lane);
    tile_to_lds<DHP, LD, NT>(Gs, dctx + (int64_t)n * S * A + h * dh, A, S, dh, lane);
    if (lane < S) Ms[lane] = mask[(int64_t)n * S + lane] ? 0.f : dc.mask_fill;
    __syncthreads();

    const bool live = lane < S;
    const float sq = sqrtf((float)dh);
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    // ---- phase A: lane = query row i
    {
        const int i = lane;
        float q[DHP], go[DHP], dq[DHP];
        float Di = 0.f;
#pragma unroll
        for (int d = 0; d < DHP; ++d) {
            q[d] = live ? Qs[i * LD + d] : 0.f;
            go[d] = live ? Gs[i * LD + d] : 0.f;
            dq[d] = 0.f;
        }
        if (live) {
            const float* orow = ctx + ((int64_t)n * S + i) * A + h * dh;
            for (int d = 0; d < dh; ++d) Di = fmaf(Gs[i * LD + d], orow[d], Di);   // D_i = dO_i . O_i
            const float m = row_max[((int64_t)n * H + h) * S + i];
            const float sum = row_sum[((int64_t)n * H + h) * S + i];
            const int64_t row_linear = ((int64_t)n * H + h) * S + i;
            for (int j = 0; j < S; ++j) {
                const float acc = dot_bcast<DHP>(q, Ks + j * DHP, 0.f);
                const float dpd = dot_bcast<DHP>(go, Vs + j * DHP, 0.f);
                float x = acc / sq;
                x += Ms[j];
                const float p = expf(x - m) / sum;
                float pd = p;
                if (dc.p > 0.f) pd *= keep_scale(dc, thr, scale, row_linear, S, j);
                const float ds = (pd * dpd - p * Di) / sq;
                Ps[i * (S + 1) + j] = pd;
                Ds[i * (S + 1) + j] = ds;
                axpy_bcast<DHP>(dq, ds, Ks + j * DHP);
            }
#pragma unroll
            for (int d = 0; d < DHP; ++d) Ts[i * LD + d] = dq[d];
        }
    }
    __syncthreads();
    float* dbase = dqkv + (int64_t)n * S * ld + h * dh;
    lds_to_tile<DHP, LD, NT>(dbase, ld, Ts, S, dh, lane);
    // ---- phase B: lane = key row j
    {
        const int j = lane;
        float dk[DHP], dv[DHP];
#pragma unroll
        for (int d = 0; d < DHP; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        if (live) {
            for (int i = 0; i < S; ++i) {
                const float ds = Ds[i * (S + 1) + j];
                const float pd = Ps[i * (S + 1) + j];
#pragma unroll
                for (int d = 0; d < DHP; ++d) {
                    dk[d] = fmaf(ds, Qs[i * LD + d], dk[d]);
                    dv[d] = fmaf(pd, Gs[i * LD + d], dv[d]);
                }
            }
        }
        __syncthreads();     // all reads of Qs / Gs done
        if (live) {
#pragma unroll
            for (int d = 0; d < DHP; ++d) { Qs[j * LD + d] = dk[d]; Gs[j * LD + d] = dv[d]; }
        }
    }
    __syncthreads();
    lds_to_tile<DHP, LD, NT>(dbase + A, ld, Qs, S, dh, lane);
    lds_to_tile<DHP, LD, NT>(dbase + 2 * A, ld, Gs, S, dh, lane);
}

// The same backward for shapes whose P / dS staging does not fit the LDS (S = 96 or 128 at head dim 64): phase B
// RECOMPUTES p_ij and dS_ij from Q, K, V, dO and the per-query (max, sum, D) instead of reading two [S][S+1] tiles, so
// the LDS need is the four operand tiles, 4*S*(DHP+1) + 4*S floats (135 KB at S = 128, dh = 64).  All four tiles use the
// odd pitch (a lane reads its OWN row in one phase and broadcast rows in the other).  About twice the arithmetic of
// the staged kernel and scalar LDS reads: the fallback that lets every shape the forward accepts train, not a fast path.
template <int DHP, int NT>
__global__ __launch_bounds__(NT) void mhsa_bwd_recompute_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                                const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                                const float* __restrict__ row_max, const float* __restrict__ row_sum,
                                                                float* __restrict__ dqkv, int S, int H, int dh, DropoutCfg dc) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = DHP + 1;
    float* Ks = smem;                 // [S][LD]
    float* Vs = Ks + S * LD;
    float* Qs = Vs + S * LD;
    float* Gs = Qs + S * LD;          // dO tile
    float* Ms = Gs + S * LD;          // [S] additive key mask
    float* Mx = Ms + S;               // [S] softmax max of query i
    float* Sm = Mx + S;               // [S] softmax sum
    float* Dd = Sm + S;               // [S] D_i = dO_i . O_i
    const int lane = threadIdx.x;
    const int n = blockIdx.x / H, h = blockIdx.x - n * H;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const float* base = qkv + (int64_t)n * S * ld + h * dh;
    tile_to_lds<DHP, LD, NT>(Qs, base, ld, S, dh, lane);
    tile_to_lds<DHP, LD, NT>(Ks, base + A, ld, S, dh, lane);
    tile_to_lds<DHP, LD, NT>(Vs, base + 2 * A, ld, S, dh, lane);
    tile_to_lds<DHP, LD, NT>(Gs, dctx + (int64_t)n * S * A + h * dh, A, S, dh, lane);
    const bool live = lane < S;
    if (live) {
        Ms[lane] = mask[(int64_t)n * S + lane] ? 0.f : dc.mask_fill;
        Mx[lane] = row_max[((int64_t)n * H + h) * S + lane];
        Sm[lane] = row_sum[((int64_t)n * H + h) * S + lane];
    }
    __syncthreads();
    const float sq = sqrtf((float)dh);
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    float* dbase = dqkv + (int64_t)n * S * ld + h * dh;
    // ---- phase A: lane = query row i -> dq_i (straight to HBM: no staging tile), D_i
    if (live) {
        const int i = lane;
        float q[DHP], go[DHP], dq[DHP];
        float Di = 0.f;
#pragma unroll
        for (int d = 0; d < DHP; ++d) { q[d] = Qs[i * LD + d]; go[d] = Gs[i * LD + d]; dq[d] = 0.f; }
        const float* orow = ctx + ((int64_t)n * S + i) * A + h * dh;
        for (int d = 0; d < dh; ++d) Di = fmaf(go[d], orow[d], Di);
        Dd[i] = Di;
        const float m = Mx[i], sum = Sm[i];
        const int64_t row_linear = ((int64_t)n * H + h) * S + i;
        for (int j = 0; j < S; ++j) {
            float acc = 0.f, dpd = 0.f;
#pragma unroll
            for (int d = 0; d < DHP; ++d) { acc = fmaf(q[d], Ks[j * LD + d], acc); dpd = fmaf(go[d], Vs[j * LD + d], dpd); }
            float x = acc / sq;
            x += Ms[j];
            const float p = expf(x - m) / sum;
            float pd = p;
            if (dc.p > 0.f) pd *= keep_scale(dc, thr, scale, row_linear, S, j);
            const float ds = (pd * dpd - p * Di) / sq;
#pragma unroll
            for (int d = 0; d < DHP; ++d) dq[d] = fmaf(ds, Ks[j * LD + d], dq[d]);
        }
        for (int d = 0; d < dh; ++d) dbase[(int64_t)i * ld + d] = dq[d];
    }
    __syncthreads();                  // D_i visible
    // ---- phase B: lane = key row j -> dk_j, dv_j with p_ij / dS_ij recomputed (same expressions, same operand order)
    if (live) {
        const int j = lane;
        float dk[DHP], dv[DHP];
#pragma unroll
        for (int d = 0; d < DHP; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        const float mj = Ms[j];
        for (int i = 0; i < S; ++i) {
            float acc = 0.f, dpd = 0.f;
#pragma unroll
            for (int d = 0; d < DHP; ++d) { acc = fmaf(Qs[i * LD + d], Ks[j * LD + d], acc); dpd = fmaf(Gs[i * LD + d], Vs[j * LD + d], dpd); }
            float x = acc / sq;
            x += mj;
            const float p = expf(x - Mx[i]) / Sm[i];
            float pd = p;
            if (dc.p > 0.f) pd *= keep_scale(dc, thr, scale, ((int64_t)n * H + h) * S + i, S, j);
            const float ds = (pd * dpd - p * Dd[i]) / sq;
#pragma unroll
            for (int d = 0; d < DHP; ++d) {
                dk[d] = fmaf(ds, Qs[i * LD + d], dk[d]);
                dv[d] = fmaf(pd, Gs[i * LD + d], dv[d]);
            }
        }
        for (int d = 0; d < dh; ++d) {
            dbase[A + (int64_t)j * ld + d] = dk[d];
            dbase[2 * A + (int64_t)j * ld + d] = dv[d];
        }
    }
}

constexpr size_t MHSA_LDS_MAX = 160 * 1024;    // one workgroup may take the CU's whole LDS
static int pick_dhp(int dh) { return dh <= 32 ? 32 : (dh <= 64 ? 64 : (dh <= 128 ? 128 : 0)); }

int mhsa_fwd(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
             int dh, DropoutCfg dc, hipStream_t s) {
    if (N == 0) return NR_OK;
    if (mhsa_mfma_supported(S, dh)) {
        ProfScope prof(FAM_ATTN, s, 4.0 * N * H * S * S * dh);
        return mhsa_fwd_mfma(qkv, mask, ctx, row_max, row_sum, N, S, H, dh, dc, s);
    }
    if (mhsa_mfma2_supported(S, dh)) {                        // (packed rows too: forward only, see mhsa_bwd)
        ProfScope prof(FAM_ATTN, s, 4.0 * N * H * S * S * dh);
        return mhsa_fwd_mfma2(qkv, mask, ctx, row_max, row_sum, N, S, H, dh, dc, s);
    }
    NR_UNSUPPORTED(dc.offs != nullptr, "mhsa: packed sequences need S <= 64 and head dim <= 64 (got S=%d, dh=%d)", S, dh);
    NR_UNSUPPORTED(S > 128 || S < 1, "mhsa: sequence length %d not in [1,128]", S);
    const int dhp = pick_dhp(dh);
    NR_UNSUPPORTED(dhp == 0, "mhsa: head dim %d > 128", dh);
    const size_t lds = (size_t)(2 * S * dhp + S * (dhp + 1) + S * (S + 1) + S) * sizeof(float);
    NR_UNSUPPORTED(lds > MHSA_LDS_MAX, "mhsa: S=%d with head dim %d needs %zu B of LDS (> %zu)", S, dh, lds, MHSA_LDS_MAX);
    ProfScope prof(FAM_ATTN, s, 4.0 * N * H * S * S * dh);
    const dim3 grid((unsigned)(N * H));
#define NR_FWD(P, T)                                                                                                \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_fwd_kernel<P, T>),                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        hipLaunchKernelGGL((mhsa_fwd_kernel<P, T>), grid, dim3(T), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc); \
    } while (0)
    if (S <= 64) { if (dhp == 32) NR_FWD(32, 64); else if (dhp == 64) NR_FWD(64, 64); else NR_FWD(128, 64); }
    else         { if (dhp == 32) NR_FWD(32, 128); else if (dhp == 64) NR_FWD(64, 128); else NR_FWD(128, 128); }
#undef NR_FWD
    return check_launch("mhsa_fwd");
}

int mhsa_bwd(const float* qkv, const int64_t* mask, const float* ctx, const float* dctx, const float* row_max,
             const float* row_sum, float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s) {
    if (N == 0) return NR_OK;
    if (mhsa_mfma_supported(S, dh)) {
        ProfScope prof(FAM_ATTN, s, 10.0 * N * H * S * S * dh);
        return mhsa_bwd_mfma(qkv, mask, dctx, row_max, row_sum, dqkv, N, S, H, dh, dc, s);
    }
    if (mhsa_mfma2_supported(S, dh) && !dc.offs) {
        ProfScope prof(FAM_ATTN, s, 10.0 * N * H * S * S * dh);
        return mhsa_bwd_mfma2(qkv, mask, dctx, row_max, row_sum, dqkv, N, S, H, dh, dc, s);
    }
    NR_UNSUPPORTED(dc.offs != nullptr, "mhsa bwd: packed sequences need S <= 32 and head dim <= 64 (got S=%d, dh=%d; 32 < S <= 64 is packed in the forward only)", S, dh);
    NR_UNSUPPORTED(S > 128 || S < 1, "mhsa bwd: sequence length %d not in [1,128]", S);
    const int dhp = pick_dhp(dh);
    NR_UNSUPPORTED(dhp == 0 || dhp > 64, "mhsa bwd: head dim %d > 64", dh);
    size_t lds = (size_t)(2 * S * dhp + 3 * S * (dhp + 1) + 2 * S * (S + 1) + S) * sizeof(float);
    ProfScope prof(FAM_ATTN, s, 10.0 * N * H * S * S * dh);
    const dim3 grid((unsigned)(N * H));
    if (lds > MHSA_LDS_MAX) {
        // the P / dS staging tiles do not fit (S = 96 / 128 at head dim 64): the recompute kernel needs the operand tiles only
        lds = (size_t)(4 * S * (dhp + 1) + 4 * S) * sizeof(float);
        NR_UNSUPPORTED(lds > MHSA_LDS_MAX, "mhsa bwd: S=%d with head dim %d needs %zu B of LDS (> %zu)", S, dh, lds, MHSA_LDS_MAX);
#define NR_BWD_R(P, T)                                                                                              \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_recompute_kernel<P, T>),                  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        hipLaunchKernelGGL((mhsa_bwd_recompute_kernel<P, T>), grid, dim3(T), lds, s, qkv, mask, ctx, dctx, row_max, row_sum, dqkv, S, H, dh, dc); \
    } while (0)
        if (S <= 64) { if (dhp == 32) NR_BWD_R(32, 64); else NR_BWD_R(64, 64); }
        else         { if (dhp == 32) NR_BWD_R(32, 128); else NR_BWD_R(64, 128); }
#undef NR_BWD_R
        return check_launch("mhsa_bwd_recompute");
    }
#define NR_BWD(P, T)                                                                                                \
    do {                                                                                                            \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_kernel<P, T>),                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                            \
        hipLaunchKernelGGL((mhsa_bwd_kernel<P, T>), grid, dim3(T), lds, s, qkv, mask, ctx, dctx, row_max, row_sum, dqkv, S, H, dh, dc); \
    } while (0)
    if (S <= 64) { if (dhp == 32) NR_BWD(32, 64); else NR_BWD(64, 64); }
    else         { if (dhp == 32) NR_BWD(32, 128); else NR_BWD(64, 128); }
#undef NR_BWD
    return check_launch("mhsa_bwd");
}

}  // namespace nr
