// Multi-head self-attention for title-length sequences (S <= 32, dh <= 64) on the f32 matrix cores.
// One wave per (sequence, head); the whole head is one 32x32 score tile.
//
// Orientation trick (v_mfma_f32_32x32x2_f32: lane l feeds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// the result puts column j on the lane and 16 rows in registers, row(r,h) = (r&3)+8(r>>2)+4h):
//   * "T" tile  St = K.Q^T : row = key, col = query  -> a lane owns ONE query and 16 of its 32 keys,
//     so the softmax over keys is in-register + one exchange with the other half-wave;
//   * an accumulator tile is directly the B operand of a product that sums over its ROW index
//     (register r of half h is k-slot (r,h); the A operand is read from LDS in the same slot order):
//       O^T  = V^T . Pd^T     dQ^T = K^T . dS^T          (from the T tiles)
//       dK^T = Q^T . dS       dV^T = dO^T . Pd           (from the "N" tiles S = Q.K^T, row = query)
//   so no P / dS tile ever moves between lanes or through LDS.  The backward recomputes both
//   orientations (13 MFMAs each at dh = 25) from the saved per-query softmax max/sum.
// Score arithmetic is the reference's: x = fl32(q.k / sqrt(dh)) + (1-mask_key)*(-1e5); p = exp(x-max)/sum.
#include "common.h"
#include "seqops.h"

namespace nr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr float MASK_FILL_M = -1e5f;

__device__ __forceinline__ int krow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// [S, dh] global tile (row stride ld) -> LDS [32][LD] zero padded (rows >= S, cols >= dh)
template <int DHP>
__device__ __forceinline__ void tile_in(float* lds, const float* __restrict__ g, int64_t ld, int S, int dh, int lane) {
    constexpr int LD = DHP + 1;
    for (int idx = lane; idx < 32 * DHP; idx += 64) {
        const int r = idx / DHP, d = idx - r * DHP;
        lds[r * LD + d] = (r < S && d < dh) ? g[(int64_t)r * ld + d] : 0.f;
    }
}
template <int DHP>
__device__ __forceinline__ void tile_out(float* __restrict__ g, int64_t ld, const float* lds, int S, int dh, int lane) {
    constexpr int LD = DHP + 1;
    for (int idx = lane; idx < S * dh; idx += 64) {
        const int r = idx / dh, d = idx - r * dh;
        g[(int64_t)r * ld + d] = lds[r * LD + d];
    }
}

// acc[r] (+)= sum_d X[row(r,h)][d] * Y[l31][d]  i.e. tile = X . Y^T with row = X's row, col = Y's row
template <int DHP>
__device__ __forceinline__ f32x16 tile_xyT(const float* Xs, const float* Ys, int dsteps, int l31, int h) {
    constexpr int LD = DHP + 1;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int s = 0; s < dsteps; ++s) {
        const float a = Xs[l31 * LD + 2 * s + h];
        const float b = Ys[l31 * LD + 2 * s + h];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    }
    return acc;
}

// out[r] = sum over the tile's ROW index k of Z[k][dt*32 + row(r,h)] * tile[k][l31]   (Z^T . tile)
template <int DHP>
__device__ __forceinline__ f32x16 zT_tile(const float* Zs, const f32x16& tile, int dt, int l31, int h) {
    constexpr int LD = DHP + 1;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float a = Zs[krow(s, h) * LD + dt * 32 + l31];
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, tile[s], o, 0, 0, 0);
    }
    return o;
}

// result tile rows = feature d (dt*32 + row(r,h)), col = sequence position l31 -> LDS [pos][d]
template <int DHP>
__device__ __forceinline__ void put_T(float* Ts, const f32x16& o, int dt, int l31, int h) {
    constexpr int LD = DHP + 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[l31 * LD + dt * 32 + krow(r, h)] = o[r];
}

__device__ __forceinline__ float keep_of(const DropoutCfg& dc, uint32_t thr, float scale, uint64_t idx) {
    const Philox4 rn = philox4x32_10(dc.seed, idx >> 2, dc.site);
    return philox_pick(rn, (int)(idx & 3)) >= thr ? scale : 0.f;
}

template <int DHP>
__global__ __launch_bounds__(64) void mhsa_fwd_mfma_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                           float* __restrict__ ctx, float* __restrict__ row_max,
                                                           float* __restrict__ row_sum, int S, int H, int dh, DropoutCfg dc) {
    constexpr int LD = DHP + 1;
    constexpr int DT = DHP / 32;
    __shared__ float Qs[32 * LD], Ks[32 * LD], Vs[32 * LD];
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / H, hd = blockIdx.x - n * H;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const float* base = qkv + (int64_t)n * S * ld + hd * dh;
    tile_in<DHP>(Qs, base, ld, S, dh, lane);
    tile_in<DHP>(Ks, base + A, ld, S, dh, lane);
    tile_in<DHP>(Vs, base + 2 * A, ld, S, dh, lane);
    __syncthreads();
    const int dsteps = (dh + 1) >> 1;
    const float sq = sqrtf((float)dh);

    f32x16 st = tile_xyT<DHP>(Ks, Qs, dsteps, l31, h);       // row = key, col = query (this lane's query = l31)
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int j = krow(r, h);
        float x = st[r] / sq;
        x += (j < S) ? (mask[(int64_t)n * S + j] ? 0.f : MASK_FILL_M) : -INFINITY;
        st[r] = x;
        m = fmaxf(m, x);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float e = expf(st[r] - m);
        st[r] = e;
        sum += e;
    }
    sum += __shfl_xor(sum, 32, 64);
    const int64_t qrow = ((int64_t)n * H + hd) * S + l31;
    if (row_max && h == 0 && l31 < S) { row_max[qrow] = m; row_sum[qrow] = sum; }
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float p = st[r] / sum;
        if (dc.p > 0.f) p *= keep_of(dc, thr, scale, (uint64_t)(qrow * S + krow(r, h)));
        st[r] = p;
    }
    __syncthreads();                                           // every lane is done reading Qs
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const f32x16 o = zT_tile<DHP>(Vs, st, dt, l31, h);     // O^T[d][query]
        put_T<DHP>(Qs, o, dt, l31, h);
    }
    __syncthreads();
    tile_out<DHP>(ctx + (int64_t)n * S * A + hd * dh, A, Qs, S, dh, lane);
}

template <int DHP>
__global__ __launch_bounds__(64) void mhsa_bwd_mfma_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                           const float* __restrict__ dctx, const float* __restrict__ row_max,
                                                           const float* __restrict__ row_sum, float* __restrict__ dqkv,
                                                           int S, int H, int dh, DropoutCfg dc) {
    constexpr int LD = DHP + 1;
    constexpr int DT = DHP / 32;
    __shared__ float Qs[32 * LD], Ks[32 * LD], Vs[32 * LD], Gs[32 * LD], Ts[32 * LD];
    __shared__ float Ms[32], Ls[32], Dsh[32], Mk[32];
    const int lane = threadIdx.x, l31 = lane & 31, h = lane >> 5;
    const int n = blockIdx.x / H, hd = blockIdx.x - n * H;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const float* base = qkv + (int64_t)n * S * ld + hd * dh;
    tile_in<DHP>(Qs, base, ld, S, dh, lane);
    tile_in<DHP>(Ks, base + A, ld, S, dh, lane);
    tile_in<DHP>(Vs, base + 2 * A, ld, S, dh, lane);
    tile_in<DHP>(Gs, dctx + (int64_t)n * S * A + hd * dh, A, S, dh, lane);
    const int64_t qrow0 = ((int64_t)n * H + hd) * S;
    if (lane < 32) {
        const bool ok = lane < S;
        Ms[lane] = ok ? row_max[qrow0 + lane] : 0.f;
        Ls[lane] = ok ? row_sum[qrow0 + lane] : 1.f;
        Mk[lane] = ok ? (mask[(int64_t)n * S + lane] ? 0.f : MASK_FILL_M) : -INFINITY;
    }
    __syncthreads();
    const int dsteps = (dh + 1) >> 1;
    const float sq = sqrtf((float)dh);
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    float* dbase = dqkv + (int64_t)n * S * ld + hd * dh;

    // ---------------- T orientation: row = key, col = query (lane's query = l31)
    {
        f32x16 st = tile_xyT<DHP>(Ks, Qs, dsteps, l31, h);
        f32x16 dpt = tile_xyT<DHP>(Vs, Gs, dsteps, l31, h);   // dPd^T[key][query] = V[key].dO[query]
        const float m = Ms[l31], sum = Ls[l31];
        float pr[16];
        float dpart = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int j = krow(r, h);
            const float x = st[r] / sq + Mk[j];
            const float p = expf(x - m) / sum;
            float pd = p;
            if (dc.p > 0.f) pd *= keep_of(dc, thr, scale, (uint64_t)((qrow0 + l31) * S + j));
            pr[r] = p;
            st[r] = pd * dpt[r];                               // Pd * dPd
            dpart += st[r];
        }
        const float D = dpart + __shfl_xor(dpart, 32, 64);
        if (h == 0) Dsh[l31] = D;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = (st[r] - pr[r] * D) / sq;      // dS^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x16 o = zT_tile<DHP>(Ks, st, dt, l31, h);              // dQ^T[d][query]
            put_T<DHP>(Ts, o, dt, l31, h);
        }
    }
    __syncthreads();
    tile_out<DHP>(dbase, ld, Ts, S, dh, lane);
    // ---------------- N orientation: row = query, col = key (lane's key = l31)
    f32x16 ds, pdn;
    {
        ds = tile_xyT<DHP>(Qs, Ks, dsteps, l31, h);
        const f32x16 dpn = tile_xyT<DHP>(Gs, Vs, dsteps, l31, h);
        const float mk = Mk[l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = krow(r, h);
            const float x = ds[r] / sq + mk;
            const float p = expf(x - Ms[i]) / Ls[i];
            float pd = p;
            if (dc.p > 0.f) pd *= keep_of(dc, thr, scale, (uint64_t)((qrow0 + i) * S + l31));
            pdn[r] = pd;
            ds[r] = (pd * dpn[r] - p * Dsh[i]) / sq;           // dS[query][key]
        }
    }
    __syncthreads();                                           // Ts free again (dq copied out)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const f32x16 o = zT_tile<DHP>(Qs, ds, dt, l31, h);     // dK^T[d][key]
        put_T<DHP>(Ts, o, dt, l31, h);
    }
    __syncthreads();
    tile_out<DHP>(dbase + A, ld, Ts, S, dh, lane);
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
        const f32x16 o = zT_tile<DHP>(Gs, pdn, dt, l31, h);    // dV^T[d][key]
        put_T<DHP>(Ts, o, dt, l31, h);
    }
    __syncthreads();
    tile_out<DHP>(dbase + 2 * A, ld, Ts, S, dh, lane);
}

bool mhsa_mfma_supported(int S, int dh) { return S >= 1 && S <= 32 && dh >= 1 && dh <= 64; }

int mhsa_fwd_mfma(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
                  int dh, DropoutCfg dc, hipStream_t s) {
    const dim3 grid((unsigned)(N * H));
    if (dh <= 32) hipLaunchKernelGGL(mhsa_fwd_mfma_kernel<32>, grid, dim3(64), 0, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc);
    else          hipLaunchKernelGGL(mhsa_fwd_mfma_kernel<64>, grid, dim3(64), 0, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc);
    return check_launch("mhsa_fwd_mfma");
}

int mhsa_bwd_mfma(const float* qkv, const int64_t* mask, const float* dctx, const float* row_max, const float* row_sum,
                  float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s) {
    const dim3 grid((unsigned)(N * H));
    if (dh <= 32) hipLaunchKernelGGL(mhsa_bwd_mfma_kernel<32>, grid, dim3(64), 0, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc);
    else          hipLaunchKernelGGL(mhsa_bwd_mfma_kernel<64>, grid, dim3(64), 0, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc);
    return check_launch("mhsa_bwd_mfma");
}

}  // namespace nr
