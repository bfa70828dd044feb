// Multi-head self-attention for title-length sequences (S <= 32, dh <= 64) on the f32 matrix cores.
// One workgroup per (sequence, group of NHG heads: 4 at head dim <= 32, 2 above — half the LDS, twice the workgroups per CU); one wave per head; a head is one 32x32
// score tile.
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
// HBM access: the 4 heads of a group are 4*dh contiguous floats of every qkv row, so the workgroup
// moves [S rows] x [4*dh] segments as 16-byte vectors (400-B runs at dh = 25) and scatters them into
// per-head LDS tiles; results leave the same way.
// Score arithmetic is the reference's: x = fl32(q.k / sqrt(dh)) + (1-mask_key)*(-1e5); p = exp(x-max)/sum.
// Dropout keep bits are drawn once, in the T orientation (4 consecutive keys share one Philox call
// when S % 4 == 0), and handed to the N orientation as a 32-bit word per query through LDS.
#include "common.h"
#include <cstdlib>
#include "seqops.h"

namespace nr {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifdef NR_ATTN_STAMP          // diagnostic build: cycle stamps of wave 0 of every workgroup (tools/attn_stamps.py)
__device__ unsigned long long attn_stamps[4096 * 16];
#define AT_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) attn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define AT_STAMP_RT(i) do { if (threadIdx.x == 0 && blockIdx.x < 4096) attn_stamps[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define AT_STAMP(i) do {} while (0)
#define AT_STAMP_RT(i) do {} while (0)
#endif
constexpr int HPG = 4;                 // heads per workgroup

__device__ __forceinline__ int krow(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ int fastdiv(int x, int inv20) { return (int)(((unsigned)x * (unsigned)inv20) >> 20); }   // x < 16384/d

// LDS pitch of a head tile: dh rounded up to odd.  No pad columns: the d loops mask what lies past dh, so a 25-wide head
// costs 25 floats per row, not 33 (3 backward / 4 forward workgroups per CU instead of 2 / 3 at dh = 25).
__device__ __forceinline__ int tile_ld(int dh) { return dh | 1; }

// zero what the MFMAs read but the copy does not write: rows >= S
template <int ROWS = 32>
__device__ __forceinline__ void tile_zero_pad(float* tile, int S, int dh, int lane) {
    const int LD = tile_ld(dh);
    for (int idx = lane; idx < (ROWS - S) * dh; idx += 64) {
        const int r = S + idx / dh, c = idx - (idx / dh) * dh;
        tile[r * LD + c] = 0.f;
    }
}

// [S rows] x [nh*dh] global segment (row stride ld) -> nh per-head tiles [32][LD], bf16 tensor
// the same for a bf16 tensor (the bf16 build keeps qkv / ctx and their gradients as bf16 in HBM; the tiles and all the
// arithmetic stay f32): 16-byte pieces of 8 elements when the segment allows it
template <int DHP>
__device__ __forceinline__ void heads_in(float* tiles, const unsigned short* __restrict__ g, int64_t ld, int S, int dh, int nh, int tid, bool vec, int nthr = 256) {
    const int LD = tile_ld(dh), TS = 32 * LD;
    const int W = nh * dh;
    const int inv_dh = (1048576 + dh - 1) / dh;
    if (vec && (W & 7) == 0) {
        const int nq = W >> 3;
        for (int idx = tid; idx < S * nq; idx += nthr) {
            const int r = idx / nq;
            const int c = (idx - r * nq) << 3;
            const uint4 u = *reinterpret_cast<const uint4*>(g + (int64_t)r * ld + c);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            int hh = fastdiv(c, inv_dh), d = c - hh * dh;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t bits = (k & 1) ? (w[k >> 1] & 0xffff0000u) : (w[k >> 1] << 16);
                tiles[hh * TS + r * LD + d] = __builtin_bit_cast(float, bits);
                if (++d == dh) { d = 0; ++hh; }
            }
        }
    } else {
        for (int idx = tid; idx < S * W; idx += nthr) {
            const int r = idx / W;
            const int c = idx - r * W;
            const int hh = fastdiv(c, inv_dh);
            tiles[hh * TS + r * LD + (c - hh * dh)] = __builtin_bit_cast(float, (uint32_t)g[(int64_t)r * ld + c] << 16);
        }
    }
}
// NSRC segments at once (q, k, v [, dO] of the same rows): the loads of one iteration are all in flight before the
// first scatter, so the workgroup pays one HBM round trip for its operands, not NSRC.  vec = 4 / 2: 16- / 8-byte pieces.
template <int DHP, int ROWS, int NSRC, int VW>
__device__ __forceinline__ void heads_in_vec(float* const (&tiles)[NSRC], const float* const (&g)[NSRC], const int64_t (&ld)[NSRC],
                                             int S, int dh, int nh, int tid, int nthr) {
    typedef float vt __attribute__((ext_vector_type(VW)));
    const int LD = tile_ld(dh), TS = ROWS * LD;
    const int inv_dh = (1048576 + dh - 1) / dh;
    const int nq = (nh * dh) / VW;
    const int inv_nq = (1048576 + nq - 1) / nq;
    for (int idx = tid; idx < S * nq; idx += nthr) {
        const int r = fastdiv(idx, inv_nq);
        const int c = (idx - r * nq) * VW;
        vt v[NSRC];
#pragma unroll
        for (int k = 0; k < NSRC; ++k) v[k] = *reinterpret_cast<const vt*>(g[k] + (int64_t)r * ld[k] + c);
        int hh = fastdiv(c, inv_dh), d = c - hh * dh;
        int off[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
            off[e] = hh * TS + r * LD + d;
            if (++d == dh) { d = 0; ++hh; }
        }
#pragma unroll
        for (int k = 0; k < NSRC; ++k)
#pragma unroll
            for (int e = 0; e < VW; ++e) tiles[k][off[e]] = v[k][e];
    }
}
template <int DHP, int ROWS, int NSRC>
__device__ __forceinline__ void heads_in_n(float* const (&tiles)[NSRC], const float* const (&g)[NSRC], const int64_t (&ld)[NSRC],
                                           int S, int dh, int nh, int tid, int vec, int nthr) {
    if (vec == 4) heads_in_vec<DHP, ROWS, NSRC, 4>(tiles, g, ld, S, dh, nh, tid, nthr);
    else if (vec == 2) heads_in_vec<DHP, ROWS, NSRC, 2>(tiles, g, ld, S, dh, nh, tid, nthr);
    else {
        const int LD = tile_ld(dh), TS = ROWS * LD;
        const int W = nh * dh;
        const int inv_dh = (1048576 + dh - 1) / dh;
        for (int idx = tid; idx < S * W; idx += nthr) {
            const int r = idx / W;
            const int c = idx - r * W;
            const int hh = fastdiv(c, inv_dh);
            float v[NSRC];
#pragma unroll
            for (int k = 0; k < NSRC; ++k) v[k] = g[k][(int64_t)r * ld[k] + c];
#pragma unroll
            for (int k = 0; k < NSRC; ++k) tiles[k][hh * TS + r * LD + (c - hh * dh)] = v[k];
        }
    }
}
// bf16 tensors: one segment after the other
template <int DHP, int ROWS, int NSRC>
__device__ __forceinline__ void heads_in_n(float* const (&tiles)[NSRC], const unsigned short* const (&g)[NSRC], const int64_t (&ld)[NSRC],
                                           int S, int dh, int nh, int tid, int vec, int nthr) {
#pragma unroll
    for (int k = 0; k < NSRC; ++k) heads_in<DHP>(tiles[k], g[k], ld[k], S, dh, nh, tid, vec != 0, nthr);
}

typedef __bf16 abf16x2 __attribute__((ext_vector_type(2)));
typedef float af32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t apack2(float lo, float hi) {
    const af32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, abf16x2));
}
template <int DHP>
__device__ __forceinline__ void heads_out(unsigned short* __restrict__ g, int64_t ld, const float* tiles, int S, int dh, int nh, int tid, int vec, int nthr = 256) {
    const int LD = tile_ld(dh), TS = 32 * LD;
    const int W = nh * dh;
    const int inv_dh = (1048576 + dh - 1) / dh;
    if (vec && (W & 7) == 0) {
        const int nq = W >> 3;
        for (int idx = tid; idx < S * nq; idx += nthr) {
            const int r = idx / nq;
            const int c = (idx - r * nq) << 3;
            int hh = fastdiv(c, inv_dh), d = c - hh * dh;
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                e[k] = tiles[hh * TS + r * LD + d];
                if (++d == dh) { d = 0; ++hh; }
            }
            *reinterpret_cast<uint4*>(g + (int64_t)r * ld + c) = make_uint4(apack2(e[0], e[1]), apack2(e[2], e[3]), apack2(e[4], e[5]), apack2(e[6], e[7]));
        }
    } else {
        for (int idx = tid; idx < S * W; idx += nthr) {
            const int r = idx / W;
            const int c = idx - r * W;
            const int hh = fastdiv(c, inv_dh);
            g[(int64_t)r * ld + c] = (unsigned short)(apack2(tiles[hh * TS + r * LD + (c - hh * dh)], 0.f) & 0xffffu);
        }
    }
}

template <int DHP, int ROWS, int VW>
__device__ __forceinline__ void heads_out_vec(float* __restrict__ g, int64_t ld, const float* tiles, int S, int dh, int nh, int tid, int nthr) {
    typedef float vt __attribute__((ext_vector_type(VW)));
    const int LD = tile_ld(dh), TS = ROWS * LD;
    const int inv_dh = (1048576 + dh - 1) / dh;
    const int nq = (nh * dh) / VW;
    const int inv_nq = (1048576 + nq - 1) / nq;
    for (int idx = tid; idx < S * nq; idx += nthr) {
        const int r = fastdiv(idx, inv_nq);
        const int c = (idx - r * nq) * VW;
        int hh = fastdiv(c, inv_dh), d = c - hh * dh;
        vt e;
#pragma unroll
        for (int k = 0; k < VW; ++k) {
            e[k] = tiles[hh * TS + r * LD + d];
            if (++d == dh) { d = 0; ++hh; }
        }
        *reinterpret_cast<vt*>(g + (int64_t)r * ld + c) = e;
    }
}
template <int DHP, int ROWS = 32>
__device__ __forceinline__ void heads_out(float* __restrict__ g, int64_t ld, const float* tiles, int S, int dh, int nh, int tid, int vec, int nthr = 256) {
    if (vec == 4) heads_out_vec<DHP, ROWS, 4>(g, ld, tiles, S, dh, nh, tid, nthr);
    else if (vec == 2) heads_out_vec<DHP, ROWS, 2>(g, ld, tiles, S, dh, nh, tid, nthr);
    else {
        const int LD = tile_ld(dh), TS = ROWS * LD;
        const int W = nh * dh;
        const int inv_dh = (1048576 + dh - 1) / dh;
        for (int idx = tid; idx < S * W; idx += nthr) {
            const int r = idx / W;
            const int c = idx - r * W;
            const int hh = fastdiv(c, inv_dh);
            g[(int64_t)r * ld + c] = tiles[hh * TS + r * LD + (c - hh * dh)];
        }
    }
}

// acc[r] = sum_d X[row(r,h)][d] * Y[l31][d]  i.e. tile = X . Y^T with row = X's row, col = Y's row.  The d loop runs in
// groups of 4 k-steps (8 columns): the 8 LDS reads of a group are issued together, so the MFMA chain does not wait one LDS
// round trip per step.  The last group masks the columns past dh and skips the k-steps that lie wholly past it.
__device__ __forceinline__ f32x16 tile_xyT(const float* Xs, const float* Ys, int dh, int l31, int h) {
    const int LD = tile_ld(dh);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* xp = Xs + l31 * LD + h;
    const float* yp = Ys + l31 * LD + h;
    const int last = (dh - 1) >> 3;                        // the group that holds column dh-1
    for (int g = 0; g < last; ++g) {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = xp[8 * g + 2 * u]; b[u] = yp[8 * g + 2 * u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
    }
    {
        float a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = 8 * last + 2 * u + h < dh;
            a[u] = ok ? xp[8 * last + 2 * u] : 0.f;
            b[u] = ok ? yp[8 * last + 2 * u] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (8 * last + 2 * u < dh) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);     // uniform
    }
    return acc;
}

// out[r] = sum over the tile's ROW index k of Z[k][dt*32 + row(r,h)] * tile[k][l31]   (Z^T . tile).  Feature rows past dh
// read whatever follows in LDS: they only reach result rows past dh, which put_T does not store.
__device__ __forceinline__ f32x16 zT_tile(const float* Zs, const f32x16& tile, int dt, int l31, int h, int LD) {
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

// the same, added to o (a second 32-row tile of the summed index)
__device__ __forceinline__ void zT_tile_acc(f32x16& o, const float* Zs, const f32x16& tile, int dt, int l31, int h, int LD) {
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const float a = Zs[krow(s, h) * LD + dt * 32 + l31];
        o = __builtin_amdgcn_mfma_f32_32x32x2f32(a, tile[s], o, 0, 0, 0);
    }
}

// result tile rows = feature d (dt*32 + row(r,h)), col = sequence position l31 -> LDS [pos][d], features < dh only
__device__ __forceinline__ void put_T(float* Ts, const f32x16& o, int dt, int l31, int h, int LD, int dh) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int d = dt * 32 + krow(r, h);
        if (d < dh) Ts[l31 * LD + d] = o[r];
    }
}

// keep flags (bit r = keep element krow(r,h)) of query row `qrow` for this lane's 16 keys
__device__ __forceinline__ unsigned keep_bits16(const DropoutCfg& dc, uint32_t thr, int64_t qrow, int S, int h) {
    unsigned bits = 0;
    if ((S & 3) == 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {                       // keys 8q+4h .. +3 share one Philox counter
            const uint64_t idx = (uint64_t)(qrow * S + 8 * q + 4 * h);
            const Philox4 rn = philox4x32_10(dc.seed, idx >> 2, dc.site);
            bits |= (rn.x >= thr ? 1u : 0u) << (4 * q);
            bits |= (rn.y >= thr ? 1u : 0u) << (4 * q + 1);
            bits |= (rn.z >= thr ? 1u : 0u) << (4 * q + 2);
            bits |= (rn.w >= thr ? 1u : 0u) << (4 * q + 3);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const uint64_t idx = (uint64_t)(qrow * S + krow(r, h));
            const Philox4 rn = philox4x32_10(dc.seed, idx >> 2, dc.site);
            bits |= (philox_pick(rn, (int)(idx & 3)) >= thr ? 1u : 0u) << r;
        }
    }
    return bits;
}

// 64-bit keep word of query row `qrow` (bit j = keep key j, j < S <= 64): the two lanes of a query split the row's
// Philox counters between them (any S: the counters need not start on a row boundary) and exchange halves
__device__ __forceinline__ uint64_t keep_word64(const DropoutCfg& dc, uint32_t thr, int64_t qrow, int S, int h) {
    const uint64_t i0 = (uint64_t)(qrow * S);
    const uint64_t c_last = (i0 + S - 1) >> 2;
    uint64_t word = 0;
    for (uint64_t c = (i0 >> 2) + h; c <= c_last; c += 2) {
        const Philox4 rn = philox4x32_10(dc.seed, c, dc.site);
        const uint32_t v[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t j = (int64_t)(4 * c + e) - (int64_t)i0;
            if (j >= 0 && j < S && v[e] >= thr) word |= 1ull << j;
        }
    }
    uint32_t lo = (uint32_t)word, hi = (uint32_t)(word >> 32);
    lo |= __shfl_xor(lo, 32, 64);
    hi |= __shfl_xor(hi, 32, 64);
    return ((uint64_t)hi << 32) | lo;
}

template <int DHP, typename T, int NHG>
__global__ __launch_bounds__(64 * NHG) void mhsa_fwd_mfma_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                            T* __restrict__ ctx, float* __restrict__ row_max,
                                                            float* __restrict__ row_sum, int S, int H, int dh, DropoutCfg dc,
                                                            int vec) {
    const int LD = tile_ld(dh), TS = 32 * LD;
    constexpr int DT = DHP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qt = smem;                  // [NHG][TS]  (reused for the output tiles)
    float* Kt = Qt + NHG * TS;
    float* Vt = Kt + NHG * TS;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int groups = (H + NHG - 1) / NHG;
    const int n = blockIdx.x / groups, gidx = blockIdx.x - n * groups;
    const int hd0 = gidx * NHG, nh = min(NHG, H - hd0), hd = hd0 + w;
    const bool active = w < nh;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    // packed layout: this sequence owns rows offs[n] .. offs[n+1] (Sn <= S of them); S stays the ORIGINAL length for
    // every index that names a position (mask, softmax statistics, dropout counters)
    const int64_t row0 = dc.offs ? (int64_t)dc.offs[n] : (int64_t)n * S;
    const int Sn = dc.offs ? dc.offs[n + 1] - dc.offs[n] : S;
    if (Sn == 0) return;                                        // a title with no rows (block-uniform)
    const T* base = qkv + row0 * ld + hd0 * dh;
    float* Mk = Vt + NHG * TS;         // [32] additive key-mask term of this sequence (one global read per key)
    if (tid < 32) Mk[tid] = tid < Sn ? (mask[(int64_t)n * S + tid] ? 0.f : dc.mask_fill) : -INFINITY;
    {
        float* const tl[3] = {Qt, Kt, Vt};
        const T* const gp[3] = {base, base + A, base + 2 * A};
        const int64_t lds3[3] = {ld, ld, ld};
        heads_in_n<DHP, 32, 3>(tl, gp, lds3, Sn, dh, nh, tid, vec, 64 * NHG);
    }
    float* Qs = Qt + w * TS; float* Ks = Kt + w * TS; float* Vs = Vt + w * TS;
    if (active) { tile_zero_pad<32>(Qs, Sn, dh, lane); tile_zero_pad<32>(Ks, Sn, dh, lane); tile_zero_pad<32>(Vs, Sn, dh, lane); }
    __syncthreads();
    f32x16 st;
    if (active) {
                const float sq = sqrtf((float)dh), rsq = 1.0f / sq;
        st = tile_xyT(Ks, Qs, dh, l31, h);           // row = key, col = query (this lane's query = l31)
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = div_by(st[r], sq, rsq) + Mk[krow(r, h)];
            st[r] = x;
            m = fmaxf(m, x);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (m == -INFINITY) m = 0.f;                           // no valid key (mask_fill = -inf): all exp() are 0
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(st[r] - m);
            st[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32, 64);
        if (sum == 0.f) sum = 1.f;                             // ... and the probabilities are 0, not 0/0
        const int64_t qrow = ((int64_t)n * H + hd) * S + l31;
        if (row_max && h == 0 && l31 < Sn) { row_max[qrow] = m; row_sum[qrow] = sum; }
        unsigned bits = 0xFFFFu;
        float scale = 1.f;
        if (dc.p > 0.f) { bits = keep_bits16(dc, drop_threshold(dc.p), qrow, S, h); scale = 1.0f / (1.0f - dc.p); }
        const float rsum = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = div_by(st[r], sum, rsum);
            if (dc.p > 0.f) p = ((bits >> r) & 1u) ? p * scale : 0.f;
            st[r] = p;
        }
        // the output goes into this wave's own Q tile, which only this wave reads: no barrier (a wave's LDS operations
        // complete in order)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x16 o = zT_tile(Vs, st, dt, l31, h, LD);     // O^T[d][query]
            put_T(Qs, o, dt, l31, h, LD, dh);
        }
    }
    __syncthreads();
    heads_out<DHP>(ctx + row0 * A + hd0 * dh, A, Qt, Sn, dh, nh, tid, vec, 64 * NHG);
}

template <int DHP, typename T, int NHG>
__global__ __launch_bounds__(64 * NHG) void mhsa_bwd_mfma_kernel(const T* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                            const T* __restrict__ dctx, const float* __restrict__ row_max,
                                                            const float* __restrict__ row_sum, T* __restrict__ dqkv,
                                                            int S, int H, int dh, DropoutCfg dc, int vec) {
    const int LD = tile_ld(dh), TS = 32 * LD;
    constexpr int DT = DHP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qt = smem;
    float* Kt = Qt + NHG * TS;
    float* Vt = Kt + NHG * TS;
    float* Gt = Vt + NHG * TS;
    float* St = Gt + NHG * TS;          // [NHG][4][32]: max, 1/sum, D, key-mask term ; then [NHG][32] keep words
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int groups = (H + NHG - 1) / NHG;
    const int n = blockIdx.x / groups, gidx = blockIdx.x - n * groups;
    const int hd0 = gidx * NHG, nh = min(NHG, H - hd0), hd = hd0 + w;
    const bool active = w < nh;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const int64_t row0 = dc.offs ? (int64_t)dc.offs[n] : (int64_t)n * S;       // packed layout: see the forward kernel
    const int Sn = dc.offs ? dc.offs[n + 1] - dc.offs[n] : S;
    if (Sn == 0) return;
    const T* base = qkv + row0 * ld + hd0 * dh;
    {
        float* const tl[4] = {Qt, Kt, Vt, Gt};
        const T* const gp[4] = {base, base + A, base + 2 * A, dctx + row0 * A + hd0 * dh};
        const int64_t lds4[4] = {ld, ld, ld, (int64_t)A};
        heads_in_n<DHP, 32, 4>(tl, gp, lds4, Sn, dh, nh, tid, vec, 64 * NHG);
    }
    float* Qs = Qt + w * TS; float* Ks = Kt + w * TS; float* Vs = Vt + w * TS; float* Gs = Gt + w * TS;
    // Rl = 1 / row_sum: the backward forms p = exp(x - max) * Rl (within an ulp of the forward's quotient; one array and
    // two instructions per element less — with the dh-wide tiles this is what lets three workgroups share a CU at dh = 25)
    float* Ms = St + w * 128; float* Rl = Ms + 32; float* Dsh = Ms + 64; float* Mk = Ms + 96;
    unsigned* Kb = reinterpret_cast<unsigned*>(St + NHG * 128) + w * 32;
    const int64_t qrow0 = ((int64_t)n * H + hd) * S;
    if (active) {
        tile_zero_pad<32>(Qs, Sn, dh, lane); tile_zero_pad<32>(Ks, Sn, dh, lane);
        tile_zero_pad<32>(Vs, Sn, dh, lane); tile_zero_pad<32>(Gs, Sn, dh, lane);
        if (lane < 32) {
            const bool ok = lane < Sn;
            Ms[lane] = ok ? row_max[qrow0 + lane] : 0.f;
            Rl[lane] = ok ? 1.0f / row_sum[qrow0 + lane] : 1.f;
            Mk[lane] = ok ? (mask[(int64_t)n * S + lane] ? 0.f : dc.mask_fill) : -INFINITY;
        }
    }
    __syncthreads();
        const float sq = sqrtf((float)dh), rsq = 1.0f / sq;
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    f32x16 dq[DT], ds, pdn;
    if (active) {
        // ---------------- T orientation: row = key, col = query (lane's query = l31)
        f32x16 st = tile_xyT(Ks, Qs, dh, l31, h);
        const f32x16 dpt = tile_xyT(Vs, Gs, dh, l31, h);   // dPd^T[key][query] = V[key].dO[query]
        const float m = Ms[l31], rsum = Rl[l31];
        unsigned bits = 0xFFFFu;
        if (dc.p > 0.f) {
            bits = keep_bits16(dc, drop_threshold(dc.p), qrow0 + l31, S, h);
            // 32-bit keep word of query l31: bit j = keep key j  (register r of half h is key krow(r,h))
            unsigned word = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) word |= ((bits >> r) & 1u) << krow(r, h);
            word |= __shfl_xor(word, 32, 64);
            if (h == 0) Kb[l31] = word;
        }
        float pr[16];
        float dpart = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = div_by(st[r], sq, rsq) + Mk[krow(r, h)];
            const float p = expf(x - m) * rsum;
            const float pd = (dc.p > 0.f) ? (((bits >> r) & 1u) ? p * scale : 0.f) : p;
            pr[r] = p;
            st[r] = pd * dpt[r];                               // Pd * dPd
            dpart += st[r];
        }
        const float D = dpart + __shfl_xor(dpart, 32, 64);
        if (h == 0) Dsh[l31] = D;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = div_by(st[r] - pr[r] * D, sq, rsq);      // dS^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) dq[dt] = zT_tile(Ks, st, dt, l31, h, LD);   // dQ^T[d][query], kept in registers
        // Dsh / Kb were written by this wave's own lanes and are read by it alone: no barrier
        // ---------------- N orientation: row = query, col = key (lane's key = l31)
        ds = tile_xyT(Qs, Ks, dh, l31, h);
        const f32x16 dpn = tile_xyT(Gs, Vs, dh, l31, h);
        const float mk = Mk[l31];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = krow(r, h);
            const float x = div_by(ds[r], sq, rsq) + mk;
            const float p = expf(x - Ms[i]) * Rl[i];
            float pd = p;
            if (dc.p > 0.f) pd = ((Kb[i] >> l31) & 1u) ? p * scale : 0.f;
            pdn[r] = pd;
            ds[r] = div_by(pd * dpn[r] - p * Dsh[i], sq, rsq);           // dS[query][key]
        }
        // this head's K and V tiles are dead and were this wave's alone: they become the staging area
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            put_T(Ks, dq[dt], dt, l31, h, LD, dh);
            const f32x16 o = zT_tile(Qs, ds, dt, l31, h, LD);     // dK^T[d][key]
            put_T(Vs, o, dt, l31, h, LD, dh);
        }
    }
    __syncthreads();
    T* dbase = dqkv + row0 * ld + hd0 * dh;
    heads_out<DHP>(dbase, ld, Kt, Sn, dh, nh, tid, vec, 64 * NHG);
    heads_out<DHP>(dbase + A, ld, Vt, Sn, dh, nh, tid, vec, 64 * NHG);
    __syncthreads();
    if (active) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x16 o = zT_tile(Gs, pdn, dt, l31, h, LD);    // dV^T[d][key]
            put_T(Ks, o, dt, l31, h, LD, dh);
        }
    }
    __syncthreads();
    heads_out<DHP>(dbase + 2 * A, ld, Kt, Sn, dh, nh, tid, vec, 64 * NHG);
}

// ---------------------------------------------------------------------------------------------------------------------
// 32 < S <= 64 (the 50-slot user tower): a head is a 2x2 grid of 32x32 score tiles and takes TWO waves.  In the T
// orientation wave t owns query tile t and both key tiles (the softmax row stays inside one lane pair); in the backward's
// N orientation wave t owns key tile t and both query tiles.  Products that sum over 64 positions chain the second
// tile's 16 MFMAs onto the first's accumulator.  Head tiles are [64][LD]; everything else is the S <= 32 kernel.
template <int DHP, int NHG>
__global__ __launch_bounds__(128 * NHG) void mhsa_fwd_mfma2_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                              float* __restrict__ ctx, float* __restrict__ row_max,
                                                              float* __restrict__ row_sum, int S, int H, int dh, DropoutCfg dc,
                                                              int vec) {
    const int LD = tile_ld(dh), TS = 64 * LD, HT = 32 * LD;
    constexpr int NTHR = 128 * NHG;
    constexpr int DT = DHP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qt = smem;                  // [NHG][TS]  (reused for the output tiles)
    float* Kt = Qt + NHG * TS;
    float* Vt = Kt + NHG * TS;
    AT_STAMP(0); AT_STAMP_RT(1);
    const int tid = threadIdx.x, w = tid >> 6, hw = w >> 1, t = w & 1, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int groups = (H + NHG - 1) / NHG;
    const int n = blockIdx.x / groups, gidx = blockIdx.x - n * groups;
    const int hd0 = gidx * NHG, nh = min(NHG, H - hd0), hd = hd0 + hw;
    const bool active = hw < nh;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    // packed layout (forward only: the cached-evaluation user tower): this sequence owns rows offs[n] .. offs[n+1]
    // (Sn <= S of them); S stays the ORIGINAL length for every index that names a position
    const int64_t row0 = dc.offs ? (int64_t)dc.offs[n] : (int64_t)n * S;
    const int Sn = dc.offs ? dc.offs[n + 1] - dc.offs[n] : S;
    if (Sn == 0) return;                                        // block-uniform
    const float* base = qkv + row0 * ld + hd0 * dh;
    float* Mk = Vt + NHG * TS;         // [64] additive key-mask term of this sequence
    if (tid < 64) Mk[tid] = tid < Sn ? (mask[(int64_t)n * S + tid] ? 0.f : dc.mask_fill) : -INFINITY;
    {
        float* const tl[3] = {Qt, Kt, Vt};
        const float* const gp[3] = {base, base + A, base + 2 * A};
        const int64_t lds3[3] = {ld, ld, ld};
        heads_in_n<DHP, 64, 3>(tl, gp, lds3, Sn, dh, nh, tid, vec, NTHR);
    }
    float* Qs = Qt + hw * TS; float* Ks = Kt + hw * TS; float* Vs = Vt + hw * TS;
    if (active) {
        if (t == 0) { tile_zero_pad<64>(Qs, Sn, dh, lane); tile_zero_pad<64>(Ks, Sn, dh, lane); }
        else tile_zero_pad<64>(Vs, Sn, dh, lane);
    }
    AT_STAMP(2);
    __syncthreads();
    AT_STAMP(3);
    f32x16 st[2];
    if (active) {
                const float sq = sqrtf((float)dh), rsq = 1.0f / sq;
        const int q = t * 32 + l31;                               // this lane's query
        float m = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = tile_xyT(Ks + kt * HT, Qs + t * HT, dh, l31, h);    // row = key, col = query
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = div_by(st[kt][r], sq, rsq) + Mk[kt * 32 + krow(r, h)];
                st[kt][r] = x;
                m = fmaxf(m, x);
            }
        }
        AT_STAMP(4);
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (m == -INFINITY) m = 0.f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(st[kt][r] - m);
                st[kt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        if (sum == 0.f) sum = 1.f;
        const int64_t qrow = ((int64_t)n * H + hd) * S + q;
        if (row_max && h == 0 && q < Sn) { row_max[qrow] = m; row_sum[qrow] = sum; }
        const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.f;
        uint64_t keep = ~0ull;
        if (dc.p > 0.f) keep = keep_word64(dc, drop_threshold(dc.p), qrow, S, h);
        const float rsum = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            const uint32_t kw = (uint32_t)(keep >> (32 * kt));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float p = div_by(st[kt][r], sum, rsum);
                if (dc.p > 0.f) p = ((kw >> krow(r, h)) & 1u) ? p * scale : 0.f;
                st[kt][r] = p;
            }
        }
    }
    AT_STAMP(5);
    __syncthreads();                                           // all waves are done reading the Q tiles
    AT_STAMP(6);
    if (active) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x16 o = zT_tile(Vs, st[0], dt, l31, h, LD);        // O^T[d][query], keys 0..31
            zT_tile_acc(o, Vs + HT, st[1], dt, l31, h, LD);       // ... + keys 32..63
            put_T(Qs + t * HT, o, dt, l31, h, LD, dh);
        }
    }
    AT_STAMP(7);
    __syncthreads();
    AT_STAMP(8);
    heads_out<DHP, 64>(ctx + row0 * A + hd0 * dh, A, Qt, Sn, dh, nh, tid, vec, NTHR);
    AT_STAMP(9); AT_STAMP_RT(10);
}

template <int DHP, int NHG>
__global__ __launch_bounds__(128 * NHG) void mhsa_bwd_mfma2_kernel(const float* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                              const float* __restrict__ dctx, const float* __restrict__ row_max,
                                                              const float* __restrict__ row_sum, float* __restrict__ dqkv,
                                                              int S, int H, int dh, DropoutCfg dc, int vec) {
    const int LD = tile_ld(dh), TS = 64 * LD, HT = 32 * LD;
    constexpr int NTHR = 128 * NHG;
    constexpr int DT = DHP / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Qt = smem;
    float* Kt = Qt + NHG * TS;
    float* Vt = Kt + NHG * TS;
    float* Gt = Vt + NHG * TS;
    float* St = Gt + NHG * TS;          // [NHG][4][64]: max, 1/sum, D, key-mask term ; then [NHG][64][2] keep words
    const int tid = threadIdx.x, w = tid >> 6, hw = w >> 1, t = w & 1, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int groups = (H + NHG - 1) / NHG;
    const int n = blockIdx.x / groups, gidx = blockIdx.x - n * groups;
    const int hd0 = gidx * NHG, nh = min(NHG, H - hd0), hd = hd0 + hw;
    const bool active = hw < nh;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const int64_t row0 = (int64_t)n * S;
    const float* base = qkv + row0 * ld + hd0 * dh;
    {
        float* const tl[4] = {Qt, Kt, Vt, Gt};
        const float* const gp[4] = {base, base + A, base + 2 * A, dctx + row0 * A + hd0 * dh};
        const int64_t lds4[4] = {ld, ld, ld, (int64_t)A};
        heads_in_n<DHP, 64, 4>(tl, gp, lds4, S, dh, nh, tid, vec, NTHR);
    }
    float* Qs = Qt + hw * TS; float* Ks = Kt + hw * TS; float* Vs = Vt + hw * TS; float* Gs = Gt + hw * TS;
    float* Ms = St + hw * 256; float* Rl = Ms + 64; float* Dsh = Ms + 128; float* Mk = Ms + 192;      // Rl = 1 / row_sum
    unsigned* Kb = reinterpret_cast<unsigned*>(St + NHG * 256) + hw * 128;     // [query][key tile]
    const int64_t qrow0 = ((int64_t)n * H + hd) * S;
    if (active) {
        const bool ok = lane < S;
        if (t == 0) {
            tile_zero_pad<64>(Qs, S, dh, lane); tile_zero_pad<64>(Ks, S, dh, lane);
            Ms[lane] = ok ? row_max[qrow0 + lane] : 0.f;
            Rl[lane] = ok ? 1.0f / row_sum[qrow0 + lane] : 1.f;
        } else {
            tile_zero_pad<64>(Vs, S, dh, lane); tile_zero_pad<64>(Gs, S, dh, lane);
            Mk[lane] = ok ? (mask[(int64_t)n * S + lane] ? 0.f : dc.mask_fill) : -INFINITY;
        }
    }
    __syncthreads();
        const float sq = sqrtf((float)dh), rsq = 1.0f / sq;
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    f32x16 dq[DT], ds[2], pdn[2];
    if (active) {
        // ---------------- T orientation: row = key (both tiles), col = query (tile t; lane's query = t*32 + l31)
        const int q = t * 32 + l31;
        const float m = Ms[q], rsum = Rl[q];
        f32x16 st[2];
        float pr[2][16];
        float dpart = 0.f;
        uint64_t keep = ~0ull;
        if (dc.p > 0.f) {
            keep = keep_word64(dc, drop_threshold(dc.p), qrow0 + q, S, h);
            if (h == 0) { Kb[q * 2] = (uint32_t)keep; Kb[q * 2 + 1] = (uint32_t)(keep >> 32); }     // bit j = keep key kt*32 + j
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            st[kt] = tile_xyT(Ks + kt * HT, Qs + t * HT, dh, l31, h);
            const f32x16 dpt = tile_xyT(Vs + kt * HT, Gs + t * HT, dh, l31, h);   // dPd^T[key][query] = V[key].dO[query]
            const uint32_t kw = (uint32_t)(keep >> (32 * kt));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = div_by(st[kt][r], sq, rsq) + Mk[kt * 32 + krow(r, h)];
                const float p = expf(x - m) * rsum;
                const float pd = (dc.p > 0.f) ? (((kw >> krow(r, h)) & 1u) ? p * scale : 0.f) : p;
                pr[kt][r] = p;
                st[kt][r] = pd * dpt[r];                           // Pd * dPd
                dpart += st[kt][r];
            }
        }
        const float D = dpart + __shfl_xor(dpart, 32, 64);
        if (h == 0) Dsh[q] = D;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[kt][r] = div_by(st[kt][r] - pr[kt][r] * D, sq, rsq);      // dS^T
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {                          // dQ^T[d][query], kept in registers
            dq[dt] = zT_tile(Ks, st[0], dt, l31, h, LD);
            zT_tile_acc(dq[dt], Ks + HT, st[1], dt, l31, h, LD);
        }
    }
    __syncthreads();                                           // Dsh / Kb of BOTH query tiles visible
    if (active) {
        // ---------------- N orientation: row = query (both tiles), col = key (tile t; lane's key = t*32 + l31)
        const float mk = Mk[t * 32 + l31];
#pragma unroll
        for (int qi = 0; qi < 2; ++qi) {
            ds[qi] = tile_xyT(Qs + qi * HT, Ks + t * HT, dh, l31, h);
            const f32x16 dpn = tile_xyT(Gs + qi * HT, Vs + t * HT, dh, l31, h);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = qi * 32 + krow(r, h);
                const float x = div_by(ds[qi][r], sq, rsq) + mk;
                const float p = expf(x - Ms[i]) * Rl[i];
                float pd = p;
                if (dc.p > 0.f) pd = ((Kb[i * 2 + t] >> l31) & 1u) ? p * scale : 0.f;
                pdn[qi][r] = pd;
                ds[qi][r] = div_by(pd * dpn[r] - p * Dsh[i], sq, rsq);       // dS[query][key]
            }
        }
    }
    __syncthreads();                                           // K and V tiles are dead: reuse them as staging
    float* dbase = dqkv + row0 * ld + hd0 * dh;
    if (active) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            put_T(Ks + t * HT, dq[dt], dt, l31, h, LD, dh);
            f32x16 o = zT_tile(Qs, ds[0], dt, l31, h, LD);        // dK^T[d][key]
            zT_tile_acc(o, Qs + HT, ds[1], dt, l31, h, LD);
            put_T(Vs + t * HT, o, dt, l31, h, LD, dh);
        }
    }
    __syncthreads();
    heads_out<DHP, 64>(dbase, ld, Kt, S, dh, nh, tid, vec, NTHR);
    heads_out<DHP, 64>(dbase + A, ld, Vt, S, dh, nh, tid, vec, NTHR);
    __syncthreads();
    if (active) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            f32x16 o = zT_tile(Gs, pdn[0], dt, l31, h, LD);       // dV^T[d][key]
            zT_tile_acc(o, Gs + HT, pdn[1], dt, l31, h, LD);
            put_T(Ks + t * HT, o, dt, l31, h, LD, dh);
        }
    }
    __syncthreads();
    heads_out<DHP, 64>(dbase + 2 * A, ld, Kt, S, dh, nh, tid, vec, NTHR);
}

bool mhsa_mfma_supported(int S, int dh) { return S >= 1 && S <= 32 && dh >= 1 && dh <= 64; }
bool mhsa_mfma2_supported(int S, int dh) { return S > 32 && S <= 64 && dh >= 1 && dh <= 64; }

static bool vec_ok(const void* a, const void* b, int H, int dh, int hpg = HPG) {
    // 16-byte segments: 4 heads start on a 16-B boundary iff the row pitch and the q/k/v offsets do
    const int A = H * dh;
    return (A % 4 == 0) && (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) &&
           (H % hpg == 0 || (dh % 4 == 0));
}

template <typename T>
static int launch_fwd(const T* qkv, const int64_t* mask, T* ctx, float* row_max, float* row_sum, int64_t N, int S, int H, int dh, DropoutCfg dc,
                      int vec, hipStream_t s) {
    const int dhp = dh <= 32 ? 32 : 64;
    const int hpg = dhp == 32 ? 4 : 2;
    const dim3 grid((unsigned)(N * ((H + hpg - 1) / hpg)));
    const size_t lds = ((size_t)3 * hpg * 32 * (dh | 1) + 32) * sizeof(float);
    if (dhp == 32) {
        hipLaunchKernelGGL((mhsa_fwd_mfma_kernel<32, T, 4>), grid, dim3(256), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc, vec);
    } else {
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_fwd_mfma_kernel<64, T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
        hipLaunchKernelGGL((mhsa_fwd_mfma_kernel<64, T, 2>), grid, dim3(128), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc, vec);
    }
    return check_launch("mhsa_fwd_mfma");
}
template <typename T>
static int launch_bwd(const T* qkv, const int64_t* mask, const T* dctx, const float* row_max, const float* row_sum, T* dqkv, int64_t N, int S,
                      int H, int dh, DropoutCfg dc, int vec, hipStream_t s) {
    const int dhp = dh <= 32 ? 32 : 64;
    const int hpg = dhp == 32 ? 4 : 2;
    const dim3 grid((unsigned)(N * ((H + hpg - 1) / hpg)));
    const size_t lds = ((size_t)4 * hpg * 32 * (dh | 1) + hpg * 128 + hpg * 32) * sizeof(float);
    if (dhp == 32) {
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_mfma_kernel<32, T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
        hipLaunchKernelGGL((mhsa_bwd_mfma_kernel<32, T, 4>), grid, dim3(256), lds, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc, vec);
    } else {
        static bool once = false;
        if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_mfma_kernel<64, T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); once = true; }
        hipLaunchKernelGGL((mhsa_bwd_mfma_kernel<64, T, 2>), grid, dim3(128), lds, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc, vec);
    }
    return check_launch("mhsa_bwd_mfma");
}

int mhsa_fwd_mfma(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
                  int dh, DropoutCfg dc, hipStream_t s) {
    return launch_fwd<float>(qkv, mask, ctx, row_max, row_sum, N, S, H, dh, dc, vec_ok(qkv, ctx, H, dh, dh <= 32 ? 4 : 2) ? 4 : 0, s);
}

int mhsa_bwd_mfma(const float* qkv, const int64_t* mask, const float* dctx, const float* row_max, const float* row_sum,
                  float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s) {
    return launch_bwd<float>(qkv, mask, dctx, row_max, row_sum, dqkv, N, S, H, dh, dc, (vec_ok(qkv, dctx, H, dh, dh <= 32 ? 4 : 2) && vec_ok(dqkv, dqkv, H, dh, dh <= 32 ? 4 : 2)) ? 4 : 0, s);
}

// 32 < S <= 64, f32 tensors, sequences not packed.  Two heads per workgroup at head dim <= 32, one above: a wave per SIMD
// (the softmax / dropout phases are VALU-bound) and, for the 32-sequence user tower, 192 workgroups instead of 96.
static int vec_width(const void* a, const void* b, int H, int dh, int hpg) {
    const int A = H * dh;
    const uintptr_t al = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b);
    if (A % 4 == 0 && (al & 15) == 0 && (hpg * dh) % 4 == 0 && (H % hpg == 0 || dh % 4 == 0)) return 4;
    if (A % 2 == 0 && (al & 7) == 0 && (hpg * dh) % 2 == 0 && (H % hpg == 0 || dh % 2 == 0)) return 2;
    return 0;
}
int mhsa_fwd_mfma2(const float* qkv, const int64_t* mask, float* ctx, float* row_max, float* row_sum, int64_t N, int S, int H,
                   int dh, DropoutCfg dc, hipStream_t s) {
    const int dhp = dh <= 32 ? 32 : 64;
    const int hpg = dhp == 32 ? 2 : 1;
    const int vec = vec_width(qkv, ctx, H, dh, hpg);
    const dim3 grid((unsigned)(N * ((H + hpg - 1) / hpg)));
    const size_t lds = ((size_t)3 * hpg * 64 * (dh | 1) + 64) * sizeof(float);
    if (dhp == 32) hipLaunchKernelGGL((mhsa_fwd_mfma2_kernel<32, 2>), grid, dim3(256), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc, vec);
    else           hipLaunchKernelGGL((mhsa_fwd_mfma2_kernel<64, 1>), grid, dim3(128), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc, vec);
    return check_launch("mhsa_fwd_mfma2");
}

int mhsa_bwd_mfma2(const float* qkv, const int64_t* mask, const float* dctx, const float* row_max, const float* row_sum,
                   float* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s) {
    const int dhp = dh <= 32 ? 32 : 64;
    const int hpg = dhp == 32 ? 2 : 1;
    const int vec = min(vec_width(qkv, dctx, H, dh, hpg), vec_width(dqkv, dqkv, H, dh, hpg));
    const dim3 grid((unsigned)(N * ((H + hpg - 1) / hpg)));
    const size_t lds = ((size_t)4 * hpg * 64 * (dh | 1) + hpg * 256 + hpg * 128) * sizeof(float);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_mfma2_kernel<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_mfma2_kernel<64, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        once = true;
    }
    if (dhp == 32) hipLaunchKernelGGL((mhsa_bwd_mfma2_kernel<32, 2>), grid, dim3(256), lds, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc, vec);
    else           hipLaunchKernelGGL((mhsa_bwd_mfma2_kernel<64, 1>), grid, dim3(128), lds, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc, vec);
    return check_launch("mhsa_bwd_mfma2");
}

// ---------------------------------------------------------------------------------------------------------------------
// bf16 build (tensors bf16 in HBM, parity bar 1e-2), head dim a multiple of 8: the seven products run on
// v_mfma_f32_32x32x16_bf16 straight from bf16 LDS tiles — 4 MFMAs for Q.K^T at dh = 64 instead of 32 on the f32 instruction.
// One tile per operand, row-major Xr[pos][d] (pitch DHP + 8), filled with 16-byte global pieces: the score-shaped products
// (k = d) read 8 consecutive d per lane with ds_read_b128; the products that sum over positions take their A operand from
// the same tile with transposed reads (zT16r below; until round 3 from a second, transposed copy of the tile) and their B
// operand from an accumulator tile: k-slot e of MFMA kb in lane half h is register 8*kb + e, i.e. position
// 16*kb + 8*(e>>2) + 4*h + (e&3).  Softmax, dropout and the saved statistics are the f32 code of the kernels above.
typedef __bf16 hbf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 hbf16x4 __attribute__((ext_vector_type(4)));
typedef float hf32x4 __attribute__((ext_vector_type(4)));
template <int DHP> struct H16 {
    static constexpr int PR = DHP + 8;                      // pitch of the row-major tiles
    static constexpr int RT = 32 * PR;                      // elements of a tile
};

// per-head row-major tiles [pos][d] -> [Sn rows] x [nh*dh] bf16 global segment
template <int DHP>
__device__ __forceinline__ void heads_out16(unsigned short* __restrict__ g, int64_t ld, const unsigned short* Rt, int rstride, int Sn, int dh,
                                            int nh, int tid, int nthr) {
    constexpr int PR = H16<DHP>::PR;
    const int nq = (nh * dh) >> 3;
    const int pph = dh >> 3;
    for (int idx = tid; idx < Sn * nq; idx += nthr) {
        const int r = idx / nq;
        const int q = idx - r * nq;
        const int hh = q / pph, d0 = (q - hh * pph) << 3;
        *reinterpret_cast<uint4*>(g + (int64_t)r * ld + (q << 3)) = *reinterpret_cast<const uint4*>(Rt + hh * rstride + r * PR + d0);
    }
}
// acc[r] = sum_d X[row(r,h)][d] * Y[l31][d] over row-major tiles (columns past dh are zero)
template <int DHP>
__device__ __forceinline__ f32x16 xyT16(const unsigned short* Xr, const unsigned short* Yr, int ksteps, int l31, int h) {
    constexpr int PR = H16<DHP>::PR;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const unsigned short* xp = Xr + l31 * PR + 8 * h;
    const unsigned short* yp = Yr + l31 * PR + 8 * h;
#pragma unroll
    for (int ks = 0; ks < DHP / 16; ++ks) {
        if (ks < ksteps) {
            const hbf16x8 fa = *reinterpret_cast<const hbf16x8*>(xp + 16 * ks);
            const hbf16x8 fb = *reinterpret_cast<const hbf16x8*>(yp + 16 * ks);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
    }
    return acc;
}
// eight accumulator registers 8*kb .. 8*kb+7 as the bf16 B operand of MFMA kb
__device__ __forceinline__ hbf16x8 pack_regs16(const f32x16& t, int kb) {
    hbf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (__bf16)t[8 * kb + e];
    return o;
}
// out[r] = sum over positions k of Z[k][dt*32 + row(r,h)] * tile[k][l31]  (tile = an accumulator tile whose ROW index is k),
// Z taken from its ROW-MAJOR tile Zr[pos][d] by transposed LDS reads (round 3): ds_read_b64_tr_b16
// hands lane i of a 16-lane group column i of a 4-row block (lane 4q+p supplies row q, columns 4p .. 4p+3; measured lane
// map: tools/tr_probe.hip), i.e. four consecutive positions of feature dt*32 + l31 — the two 8-byte pieces of the operand
// above without a transposed copy of the tile.  The Xt tiles (3 of a head's 7 tiles in the backward, 1 of 3 in the
// forward) and the 2-byte scatter stores that filled them are gone: 36.9 instead of 70.6 KB per backward workgroup at
// head dim 64, four workgroups per CU instead of two.  (Pitch DHP + 8 keeps the b128 row reads conflict-free and leaves
// these reads 2-way conflicted: 24 of them per head.)  EXEC must be all ones: call from wave-uniform code only.
// Until round 3 the operand came as two 8-byte pieces of a transposed copy Xt[d][pos] of the tile.
typedef short hs16x4 __attribute__((ext_vector_type(4)));
typedef short hs16x8 __attribute__((ext_vector_type(8)));
template <int DHP>
__device__ __forceinline__ f32x16 zT16r(const unsigned short* Zr, const f32x16& tile, int dt, int lane) {
    constexpr int PR = H16<DHP>::PR;
    f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.f;
    const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
    const unsigned short* zp = Zr + (4 * (g >> 1) + q) * PR + dt * 32 + 16 * (g & 1) + 4 * pq;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        const hs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(zp + (16 * kb) * PR));
        const hs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) hs16x4*)(zp + (16 * kb + 8) * PR));
        const hs16x8 a8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(hbf16x8, a8), pack_regs16(tile, kb), o, 0, 0, 0);
    }
    return o;
}
// NSRC [Sn rows] x [nh*dh] global segments (q, k, v [, dO] of the same rows) -> row-major per-head tiles: the loads of one
// round are all issued before the first LDS store, so the workgroup pays one HBM round trip for its operands, not NSRC
template <int DHP, int NSRC>
__device__ __forceinline__ void heads_in16_multi(unsigned short* const (&Rt)[NSRC], int rstride, const unsigned short* const (&g)[NSRC], const int64_t (&ld)[NSRC],
                                                 int Sn, int dh, int nh, int tid, int nthr) {
    constexpr int PR = H16<DHP>::PR;
    const int nq = (nh * dh) >> 3;
    const int pph = dh >> 3;
    for (int idx0 = 0; idx0 < Sn * nq; idx0 += 2 * nthr) {
        uint4 u[2][NSRC];
        int off[2];
        bool ok[2];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int idx = idx0 + x * nthr + tid;
            ok[x] = idx < Sn * nq;
            const int r = ok[x] ? idx / nq : 0;
            const int q = ok[x] ? idx - r * nq : 0;
            const int hh = q / pph, d0 = (q - hh * pph) << 3;
            off[x] = hh * rstride + r * PR + d0;
#pragma unroll
            for (int k = 0; k < NSRC; ++k) u[x][k] = *reinterpret_cast<const uint4*>(g[k] + (int64_t)r * ld[k] + (q << 3));
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
            if (ok[x]) {
#pragma unroll
                for (int k = 0; k < NSRC; ++k) *reinterpret_cast<uint4*>(Rt[k] + off[x]) = u[x][k];
            }
    }
}
// result rows = feature d (dt*32 + row(r,h)), col = position l31 -> row-major bf16 tile [pos][d]: registers 4g .. 4g+3 are
// four consecutive features = one 8-byte store
template <int DHP>
__device__ __forceinline__ void put16(unsigned short* Tr, const f32x16& o, int dt, int l31, int h) {
    constexpr int PR = H16<DHP>::PR;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
        const uint2 v = make_uint2(apack2(o[4 * gq], o[4 * gq + 1]), apack2(o[4 * gq + 2], o[4 * gq + 3]));
        *reinterpret_cast<uint2*>(Tr + l31 * PR + dt * 32 + 8 * gq + 4 * h) = v;
    }
}

template <int DHP, int NHG>
__global__ __launch_bounds__(64 * NHG) void mhsa_fwd_mfma16_kernel(const unsigned short* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                               unsigned short* __restrict__ ctx, float* __restrict__ row_max,
                                                               float* __restrict__ row_sum, int S, int H, int dh, DropoutCfg dc) {
    constexpr int RT = H16<DHP>::RT, NTHR = 64 * NHG;
    constexpr int DT = DHP / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned short hsm16[];
    unsigned short* Qr = hsm16;                       // [NHG][RT]  (reused for the output tiles)
    unsigned short* Kr = Qr + NHG * RT;
    unsigned short* Vr = Kr + NHG * RT;
    float* Mk = reinterpret_cast<float*>(Vr + NHG * RT);
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int groups = (H + NHG - 1) / NHG;
    const int n = blockIdx.x / groups, gidx = blockIdx.x - n * groups;
    const int hd0 = gidx * NHG, nh = min(NHG, H - hd0), hd = hd0 + w;
    const bool active = w < nh;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const int64_t row0 = dc.offs ? (int64_t)dc.offs[n] : (int64_t)n * S;
    const int Sn = dc.offs ? dc.offs[n + 1] - dc.offs[n] : S;
    if (Sn == 0) return;
    // zero everything the MFMAs read (rows / positions past Sn, columns past dh), then fill
    {
        uint4* z = reinterpret_cast<uint4*>(hsm16);
        const int n16 = (NHG * 3 * RT) >> 3;
        for (int i = tid; i < n16; i += NTHR) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (tid < 32) Mk[tid] = tid < Sn ? (mask[(int64_t)n * S + tid] ? 0.f : dc.mask_fill) : -INFINITY;
    __syncthreads();
    const unsigned short* base = qkv + row0 * ld + hd0 * dh;
    {
        unsigned short* const dst[3] = {Qr, Kr, Vr};
        const unsigned short* const src[3] = {base, base + A, base + 2 * A};
        const int64_t lds_[3] = {ld, ld, ld};
        heads_in16_multi<DHP, 3>(dst, RT, src, lds_, Sn, dh, nh, tid, NTHR);
    }
    __syncthreads();
    unsigned short* Qs = Qr + w * RT; const unsigned short* Ks = Kr + w * RT; const unsigned short* Vs = Vr + w * RT;
    f32x16 st;
    if (active) {
        const float sq = sqrtf((float)dh), rsq = 1.0f / sq;
        st = xyT16<DHP>(Ks, Qs, (dh + 15) >> 4, l31, h);       // row = key, col = query (this lane's query = l31)
        float m = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = div_by(st[r], sq, rsq) + Mk[krow(r, h)];
            st[r] = x;
            m = fmaxf(m, x);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        if (m == -INFINITY) m = 0.f;
        float sum = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(st[r] - m);
            st[r] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 32, 64);
        if (sum == 0.f) sum = 1.f;
        const int64_t qrow = ((int64_t)n * H + hd) * S + l31;
        if (row_max && h == 0 && l31 < Sn) { row_max[qrow] = m; row_sum[qrow] = sum; }
        unsigned bits = 0xFFFFu;
        float scale = 1.f;
        if (dc.p > 0.f) { bits = keep_bits16(dc, drop_threshold(dc.p), qrow, S, h); scale = 1.0f / (1.0f - dc.p); }
        const float rsum = 1.0f / sum;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float p = div_by(st[r], sum, rsum);
            if (dc.p > 0.f) p = ((bits >> r) & 1u) ? p * scale : 0.f;
            st[r] = p;
        }
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const f32x16 o = zT16r<DHP>(Vs, st, dt, lane);         // O^T[d][query]
            put16<DHP>(Qs, o, dt, l31, h);                          // this wave's own Q tile: nobody else reads it
        }
    }
    __syncthreads();
    heads_out16<DHP>(ctx + row0 * A + hd0 * dh, A, Qr, RT, Sn, dh, nh, tid, NTHR);
}

template <int DHP, int NHG>
__global__ __launch_bounds__(64 * NHG) void mhsa_bwd_mfma16_kernel(const unsigned short* __restrict__ qkv, const int64_t* __restrict__ mask,
                                                               const unsigned short* __restrict__ dctx, const float* __restrict__ row_max,
                                                               const float* __restrict__ row_sum, unsigned short* __restrict__ dqkv,
                                                               int S, int H, int dh, DropoutCfg dc) {
    constexpr int RT = H16<DHP>::RT, NTHR = 64 * NHG;
    constexpr int DT = DHP / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned short hsm16[];
    unsigned short* Qr = hsm16;                       // row-major [NHG][RT] each
    unsigned short* Kr = Qr + NHG * RT;
    unsigned short* Vr = Kr + NHG * RT;
    unsigned short* Gr = Vr + NHG * RT;
    float* St = reinterpret_cast<float*>(Gr + NHG * RT);   // [NHG][4][32]: max, 1/sum, D, key-mask term ; then [NHG][32] keep words
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
    const int groups = (H + NHG - 1) / NHG;
    const int n = blockIdx.x / groups, gidx = blockIdx.x - n * groups;
    const int hd0 = gidx * NHG, nh = min(NHG, H - hd0), hd = hd0 + w;
    const bool active = w < nh;
    const int A = H * dh;
    const int64_t ld = 3 * (int64_t)A;
    const int64_t row0 = dc.offs ? (int64_t)dc.offs[n] : (int64_t)n * S;
    const int Sn = dc.offs ? dc.offs[n + 1] - dc.offs[n] : S;
    if (Sn == 0) return;
    {
        uint4* z = reinterpret_cast<uint4*>(hsm16);
        const int n16 = (NHG * 4 * RT) >> 3;
        for (int i = tid; i < n16; i += NTHR) z[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    float* Ms = St + w * 128; float* Rl = Ms + 32; float* Dsh = Ms + 64; float* Mk = Ms + 96;
    unsigned* Kb = reinterpret_cast<unsigned*>(St + NHG * 128) + w * 32;
    const int64_t qrow0 = ((int64_t)n * H + hd) * S;
    if (active && lane < 32) {
        const bool ok = lane < Sn;
        Ms[lane] = ok ? row_max[qrow0 + lane] : 0.f;
        Rl[lane] = ok ? 1.0f / row_sum[qrow0 + lane] : 1.f;
        Mk[lane] = ok ? (mask[(int64_t)n * S + lane] ? 0.f : dc.mask_fill) : -INFINITY;
    }
    __syncthreads();
    const unsigned short* base = qkv + row0 * ld + hd0 * dh;
    {
        unsigned short* const dst[4] = {Qr, Kr, Vr, Gr};
        const unsigned short* const src[4] = {base, base + A, base + 2 * A, dctx + row0 * A + hd0 * dh};
        const int64_t lds_[4] = {ld, ld, ld, (int64_t)A};
        heads_in16_multi<DHP, 4>(dst, RT, src, lds_, Sn, dh, nh, tid, NTHR);
    }
    __syncthreads();
    unsigned short* Qs = Qr + w * RT; unsigned short* Ks = Kr + w * RT; unsigned short* Vs = Vr + w * RT; const unsigned short* Gs = Gr + w * RT;
    const int ksteps = (dh + 15) >> 4;
    const float sq = sqrtf((float)dh), rsq = 1.0f / sq;
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    if (active) {
        f32x16 dq[DT], dk[DT], dv[DT];
        {
            // ---------------- T orientation: row = key, col = query (lane's query = l31)
            f32x16 st = xyT16<DHP>(Ks, Qs, ksteps, l31, h);
            const f32x16 dpt = xyT16<DHP>(Vs, Gs, ksteps, l31, h);      // dPd^T[key][query] = V[key].dO[query]
            const float m = Ms[l31], rsum = Rl[l31];
            unsigned bits = 0xFFFFu;
            if (dc.p > 0.f) {
                bits = keep_bits16(dc, drop_threshold(dc.p), qrow0 + l31, S, h);
                unsigned word = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) word |= ((bits >> r) & 1u) << krow(r, h);
                word |= __shfl_xor(word, 32, 64);
                if (h == 0) Kb[l31] = word;
            }
            float pr[16];
            float dpart = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float x = div_by(st[r], sq, rsq) + Mk[krow(r, h)];
                const float p = expf(x - m) * rsum;
                const float pd = (dc.p > 0.f) ? (((bits >> r) & 1u) ? p * scale : 0.f) : p;
                pr[r] = p;
                st[r] = pd * dpt[r];                               // Pd * dPd
                dpart += st[r];
            }
            const float D = dpart + __shfl_xor(dpart, 32, 64);
            if (h == 0) Dsh[l31] = D;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = div_by(st[r] - pr[r] * D, sq, rsq);      // dS^T
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = zT16r<DHP>(Ks, st, dt, lane);        // dQ^T[d][query]
        }
        // Dsh / Kb were written by this wave's own lanes: LDS operations of a wave complete in order
        {
            // ---------------- N orientation: row = query, col = key (lane's key = l31)
            f32x16 ds = xyT16<DHP>(Qs, Ks, ksteps, l31, h);
            const f32x16 dpn = xyT16<DHP>(Gs, Vs, ksteps, l31, h);
            f32x16 pdn;
            const float mk = Mk[l31];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = krow(r, h);
                const float x = div_by(ds[r], sq, rsq) + mk;
                const float p = expf(x - Ms[i]) * Rl[i];
                float pd = p;
                if (dc.p > 0.f) pd = ((Kb[i] >> l31) & 1u) ? p * scale : 0.f;
                pdn[r] = pd;
                ds[r] = div_by(pd * dpn[r] - p * Dsh[i], sq, rsq);           // dS[query][key]
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                dk[dt] = zT16r<DHP>(Qs, ds, dt, lane);             // dK^T[d][key]
                dv[dt] = zT16r<DHP>(Gs, pdn, dt, lane);            // dV^T[d][key]
            }
        }
        // every read of this head's tiles is done (they are this wave's alone): the row-major tiles become the staging area
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            put16<DHP>(Qs, dq[dt], dt, l31, h);
            put16<DHP>(Ks, dk[dt], dt, l31, h);
            put16<DHP>(Vs, dv[dt], dt, l31, h);
        }
    }
    __syncthreads();
    unsigned short* dbase = dqkv + row0 * ld + hd0 * dh;
    heads_out16<DHP>(dbase, ld, Qr, RT, Sn, dh, nh, tid, NTHR);
    heads_out16<DHP>(dbase + A, ld, Kr, RT, Sn, dh, nh, tid, NTHR);
    heads_out16<DHP>(dbase + 2 * A, ld, Vr, RT, Sn, dh, nh, tid, NTHR);
}

// bf16 tensors in HBM (raw 16-bit storage), f32 tiles and arithmetic; 16-byte pieces need A % 8 == 0 and aligned bases
static bool vec_ok_h(const void* a, const void* b, int H, int dh) {
    const int A = H * dh;
    return (A % 8 == 0) && (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0) && (((dh <= 32 ? 4 : 2) * dh) % 8 == 0);
}
// the bf16-MFMA kernels take head dims that are multiples of 8 on 16-byte aligned tensors (NR_ATTN_BF16_MFMA=0: f32 arithmetic)
static bool mfma16_ok(const void* a, const void* b, int H, int dh) {
    static const bool on = !(getenv("NR_ATTN_BF16_MFMA") && atoi(getenv("NR_ATTN_BF16_MFMA")) == 0);
    return on && dh % 8 == 0 && dh <= 64 && (((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b)) & 15) == 0);
}
int mhsa_fwd_mfma_h(const unsigned short* qkv, const int64_t* mask, unsigned short* ctx, float* row_max, float* row_sum, int64_t N, int S,
                    int H, int dh, DropoutCfg dc, hipStream_t s) {
    if (N == 0) return NR_OK;
    NR_UNSUPPORTED(!mhsa_mfma_supported(S, dh), "mhsa (bf16 tensors): S=%d, head dim %d outside the matrix-core kernel (S <= 32, dh <= 64)", S, dh);
    ProfScope prof(FAM_ATTN, s, 4.0 * N * H * S * S * dh);
    if (mfma16_ok(qkv, ctx, H, dh)) {
        const int dhp = dh <= 32 ? 32 : 64, hpg = 2;
        const dim3 grid((unsigned)(N * ((H + hpg - 1) / hpg)));
        if (dhp == 32) {
            const size_t lds = (size_t)hpg * 3 * H16<32>::RT * 2 + 128;
            hipLaunchKernelGGL((mhsa_fwd_mfma16_kernel<32, 2>), grid, dim3(128), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc);
        } else {
            const size_t lds = (size_t)hpg * 3 * H16<64>::RT * 2 + 128;
            hipLaunchKernelGGL((mhsa_fwd_mfma16_kernel<64, 2>), grid, dim3(128), lds, s, qkv, mask, ctx, row_max, row_sum, S, H, dh, dc);
        }
        return check_launch("mhsa_fwd_mfma16");
    }
    return launch_fwd<unsigned short>(qkv, mask, ctx, row_max, row_sum, N, S, H, dh, dc, vec_ok_h(qkv, ctx, H, dh) ? 1 : 0, s);
}
int mhsa_bwd_mfma_h(const unsigned short* qkv, const int64_t* mask, const unsigned short* dctx, const float* row_max, const float* row_sum,
                    unsigned short* dqkv, int64_t N, int S, int H, int dh, DropoutCfg dc, hipStream_t s) {
    if (N == 0) return NR_OK;
    NR_UNSUPPORTED(!mhsa_mfma_supported(S, dh), "mhsa bwd (bf16 tensors): S=%d, head dim %d outside the matrix-core kernel (S <= 32, dh <= 64)", S, dh);
    ProfScope prof(FAM_ATTN, s, 10.0 * N * H * S * S * dh);
    if (mfma16_ok(qkv, dctx, H, dh) && mfma16_ok(dqkv, dqkv, H, dh)) {
        const int dhp = dh <= 32 ? 32 : 64, hpg = 2;
        const dim3 grid((unsigned)(N * ((H + hpg - 1) / hpg)));
        static bool once = false;
        if (!once) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_mfma16_kernel<32, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mhsa_bwd_mfma16_kernel<64, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            once = true;
        }
        if (dhp == 32) {
            const size_t lds = (size_t)hpg * 4 * H16<32>::RT * 2 + (size_t)hpg * 160 * 4;
            hipLaunchKernelGGL((mhsa_bwd_mfma16_kernel<32, 2>), grid, dim3(128), lds, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc);
        } else {
            const size_t lds = (size_t)hpg * 4 * H16<64>::RT * 2 + (size_t)hpg * 160 * 4;
            hipLaunchKernelGGL((mhsa_bwd_mfma16_kernel<64, 2>), grid, dim3(128), lds, s, qkv, mask, dctx, row_max, row_sum, dqkv, S, H, dh, dc);
        }
        return check_launch("mhsa_bwd_mfma16");
    }
    return launch_bwd<unsigned short>(qkv, mask, dctx, row_max, row_sum, dqkv, N, S, H, dh, dc, (vec_ok_h(qkv, dctx, H, dh) && vec_ok_h(dqkv, dqkv, H, dh)) ? 1 : 0, s);
}

}  // namespace nr

#ifdef NR_ATTN_STAMP
extern "C" int nr_debug_attn_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nr::attn_stamps), sizeof(unsigned long long) * n);
}
#endif
