// Per-sequence HBM-bound kernels, one wave per sequence / impression:
//   * single-query additive pooling (news & user vectors), fwd + bwd
//   * candidate scoring (+ softmax cross-entropy), fwd + bwd, with optional row gather
//     from the news-vector cache (the evaluation "segmented gather-dot")
//   * unmasked sum / mean over the history (Sum/Avg user encoders), fwd + bwd
// A wave walks the sequence row by row; each row (D floats) is read once as 16-B vectors
// (lane l holds columns 4*(l + 64*it)..+3), dot products are finished with wave reductions,
// per-position scalars (scores, probabilities, logits) live one per lane (C <= 64; pooling: SL = 1 or 2 per lane, S <= 128).
#include "common.h"
#include "seqops.h"
#include "rowops.h"

namespace nr {

constexpr float MASK_FILL = -1e5f;
constexpr int WPB = 4;                 // waves (sequences) per 256-thread block

// row pointer of position s in sequence n: either dense [nseq,S,D] or gathered from a table
// (a bad index is clamped into the table instead of faulting the GPU, and raises the library's index flag: nr_index_status)
__device__ __forceinline__ const float* row_ptr(const float* dense, const float* table, const int64_t* idx, int64_t nrows,
                                                int64_t n, int S, int s, int D, int* oob) {
    if (idx) {
        int64_t r = idx[n * S + s];
        if (r < 0 || r >= nrows) {
            if (oob) *oob = 1;
            r = r < 0 ? 0 : nrows - 1;
        }
        return table + r * (int64_t)D;
    }
    return dense + (n * S + s) * (int64_t)D;
}

template <int IT>
__device__ __forceinline__ void load_row(float4 (&r)[IT], const float* p, int D, int lane) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * 4;
        r[it] = (c < D) ? *reinterpret_cast<const float4*>(p + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int IT>
__device__ __forceinline__ void store_row(float* p, const float4 (&r)[IT], int D, int lane) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) *reinterpret_cast<float4*>(p + c) = r[it];
    }
}
template <int IT>
__device__ __forceinline__ float dot_rows(const float4 (&a)[IT], const float4 (&b)[IT]) {
    float acc = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        acc = fmaf(a[it].x, b[it].x, acc); acc = fmaf(a[it].y, b[it].y, acc);
        acc = fmaf(a[it].z, b[it].z, acc); acc = fmaf(a[it].w, b[it].w, acc);
    }
    return wave_sum(acc);
}
template <int IT>
__device__ __forceinline__ void axpy_rows(float4 (&o)[IT], float a, const float4 (&x)[IT]) {
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        o[it].x = fmaf(a, x[it].x, o[it].x); o[it].y = fmaf(a, x[it].y, o[it].y);
        o[it].z = fmaf(a, x[it].z, o[it].z); o[it].w = fmaf(a, x[it].w, o[it].w);
    }
}

// ------------------------------------------------------------------ pooling forward
// out[n] = softmax_s( q.key[n,s]/sqrt(D) + (1-mask[n,s])*(-1e5) ) . value[n,s]
// One 256-thread block per sequence: wave w owns positions s = w, w+4, ... and keeps PB of its rows in
// flight at a time (a sequence per wave left ~7 waves per CU on the NRMS shape: latency-, not HBM-bound).
// Scores meet in LDS; every wave then forms the softmax redundantly (one score per lane) and
// accumulates prob * value over its own rows; the four partial rows are summed through LDS.
constexpr int PB = 4;                  // rows in flight per wave

template <int IT, int SL>
__global__ __launch_bounds__(256) void pool_fwd_kernel(const float* __restrict__ key, const float* __restrict__ value,
                                                       const float* __restrict__ table, const int64_t* __restrict__ idx,
                                                       int64_t nrows,
                                                       const float* __restrict__ query, const int64_t* __restrict__ mask,
                                                       float* __restrict__ out, float* __restrict__ prob, int64_t nseq,
                                                       int S, int D, const int* __restrict__ offs, int* __restrict__ oob) {
    __shared__ float sc[64 * SL];
    __shared__ __attribute__((aligned(16))) float part[WPB][IT * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n = blockIdx.x;
    // packed layout (offs != NULL, dense key/value only): sequence n owns rows offs[n] .. offs[n+1]; S stays the original
    // length for the mask / prob indices.  The dropped positions had probability exactly 0 in the padded layout.
    const int Sn = offs ? offs[n + 1] - offs[n] : S;
    const int64_t rb = offs ? (int64_t)offs[n] - n * S : 0;       // dense row of (n, s) = n*S + s + rb
    key += rb * D; value += rb * D;
    if (Sn == 0) {                                                // a title with no rows: zero vector, zero probabilities
        for (int c = threadIdx.x; c < D; c += 256) out[n * (int64_t)D + c] = 0.f;
        if (prob) for (int c = threadIdx.x; c < S; c += 256) prob[n * S + c] = 0.f;
        return;
    }
    float4 q[IT];
    load_row<IT>(q, query, D, lane);
    const float sq = sqrtf((float)D);
    for (int s0 = w; s0 < Sn; s0 += WPB * PB) {
        float4 k[PB][IT];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) load_row<IT>(k[b], row_ptr(key, table, idx, nrows, n, S, s, D, oob), D, lane);
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) {
                float x = dot_rows<IT>(q, k[b]) / sq;
                if (mask) x += mask[n * S + s] ? 0.f : MASK_FILL;
                if (lane == 0) sc[s] = x;
            }
        }
    }
    __syncthreads();
    // positions lane (and lane + 64 when SL == 2) of the softmax live in this lane
    const float myscore = lane < Sn ? sc[lane] : -INFINITY;
    const float myscore2 = (SL == 2 && lane + 64 < Sn) ? sc[(SL - 1) * 64 + lane] : -INFINITY;
    const float m = wave_max(SL == 2 ? fmaxf(myscore, myscore2) : myscore);
    const float e = lane < Sn ? expf(myscore - m) : 0.f;
    const float e2 = (SL == 2 && lane + 64 < Sn) ? expf(myscore2 - m) : 0.f;
    const float sum = wave_sum(SL == 2 ? e + e2 : e);
    const float p = e / sum, p2 = e2 / sum;
    if (prob && w == 0 && lane < S) prob[n * S + lane] = p;
    if (SL == 2 && prob && w == 0 && lane + 64 < S) prob[n * S + lane + 64] = p2;
    float4 acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = w; s0 < Sn; s0 += WPB * PB) {
        float4 v[PB][IT];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) load_row<IT>(v[b], row_ptr(value, table, idx, nrows, n, S, s, D, oob), D, lane);
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) axpy_rows<IT>(acc, __shfl((SL == 2 && s >= 64) ? p2 : p, s & 63, 64), v[b]);
        }
    }
    // fixed-order cross-wave sum (deterministic): wave w writes its partial, the block adds 0+1+2+3
#pragma unroll
    for (int it = 0; it < IT; ++it) *reinterpret_cast<float4*>(&part[w][(it * 64 + lane) * 4]) = acc[it];
    __syncthreads();
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        const float4 a0 = *reinterpret_cast<const float4*>(&part[0][c]), a1 = *reinterpret_cast<const float4*>(&part[1][c]);
        const float4 a2 = *reinterpret_cast<const float4*>(&part[2][c]), a3 = *reinterpret_cast<const float4*>(&part[3][c]);
        float4 r;
        r.x = ((a0.x + a1.x) + a2.x) + a3.x; r.y = ((a0.y + a1.y) + a2.y) + a3.y;
        r.z = ((a0.z + a1.z) + a2.z) + a3.z; r.w = ((a0.w + a1.w) + a2.w) + a3.w;
        *reinterpret_cast<float4*>(out + n * (int64_t)D + c) = r;
    }
}

// ------------------------------------------------------------------ pooling backward
// given dout[n,D] and saved prob[n,S]:
//   dp_s = dout.value_s ; dscore_s = p_s (dp_s - sum_t p_t dp_t) / sqrt(D)
//   dvalue_s = p_s dout            (direct path)
//   dkey_s   = dscore_s q          -> through tanh: dz_s = dkey_s * (1 - key_s^2)   (tanh_key != 0)
//   dq_part[n] = sum_s dscore_s key_s
// key_is_value (AttnUserEncoder: key = value = x, no projection): dvalue_s += dkey_s, dkeypre unused.
// Same block-per-sequence split as the forward kernel.
template <int IT, int SL>
__global__ __launch_bounds__(256) void pool_bwd_kernel(const float* __restrict__ key, const float* __restrict__ value,
                                                       const float* __restrict__ query, const float* __restrict__ prob,
                                                       const float* __restrict__ dout, float* __restrict__ dkeypre,
                                                       float* __restrict__ dvalue, float* __restrict__ dq_part,
                                                       int64_t nseq, int S, int D, int tanh_key, int key_is_value,
                                                       const int* __restrict__ offs, const float* __restrict__ dout1, int64_t split) {
    __shared__ float sdp[64 * SL];
    __shared__ __attribute__((aligned(16))) float part[WPB][IT * 256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t n = blockIdx.x;
    const int Sn = offs ? offs[n + 1] - offs[n] : S;              // packed layout: see pool_fwd_kernel
    const int64_t rb = offs ? (int64_t)offs[n] - n * S : 0;
    key += rb * D; value += rb * D; dvalue += rb * D;
    if (dkeypre) dkeypre += rb * D;
    if (Sn == 0) {
        for (int c = threadIdx.x; c < D; c += 256) dq_part[n * (int64_t)D + c] = 0.f;
        return;
    }
    float4 q[IT], g[IT];
    load_row<IT>(q, query, D, lane);
    // (two upstream gradients: sequences 0 .. split-1 read dout, the rest dout1 — candidates | history of a train step)
    load_row<IT>(g, (dout1 && n >= split) ? dout1 + (n - split) * (int64_t)D : dout + n * (int64_t)D, D, lane);
    const float sq = sqrtf((float)D);
    const float p = lane < Sn ? prob[n * S + lane] : 0.f;
    const float p2 = (SL == 2 && lane + 64 < Sn) ? prob[n * S + lane + 64] : 0.f;
    for (int s0 = w; s0 < Sn; s0 += WPB * PB) {
        float4 v[PB][IT];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) load_row<IT>(v[b], value + (n * S + s) * (int64_t)D, D, lane);
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) {
                const float dp = dot_rows<IT>(g, v[b]);
                if (lane == 0) sdp[s] = dp;
            }
        }
    }
    __syncthreads();
    const float mydp = lane < Sn ? sdp[lane] : 0.f;
    const float mydp2 = (SL == 2 && lane + 64 < Sn) ? sdp[(SL - 1) * 64 + lane] : 0.f;
    const float tot = wave_sum(SL == 2 ? p * mydp + p2 * mydp2 : p * mydp);
    const float dscore = p * (mydp - tot) / sq, dscore2 = p2 * (mydp2 - tot) / sq;
    float4 dq[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) dq[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = w; s0 < Sn; s0 += WPB * PB) {
        float4 k[PB][IT];
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s < Sn) load_row<IT>(k[b], key + (n * S + s) * (int64_t)D, D, lane);
        }
#pragma unroll
        for (int b = 0; b < PB; ++b) {
            const int s = s0 + b * WPB;
            if (s >= Sn) continue;
            const float ps = __shfl((SL == 2 && s >= 64) ? p2 : p, s & 63, 64);
            const float dsc = __shfl((SL == 2 && s >= 64) ? dscore2 : dscore, s & 63, 64);
            float4 dv[IT], dk[IT];
            axpy_rows<IT>(dq, dsc, k[b]);
#pragma unroll
            for (int it = 0; it < IT; ++it) {
                dv[it] = make_float4(ps * g[it].x, ps * g[it].y, ps * g[it].z, ps * g[it].w);
                dk[it] = make_float4(dsc * q[it].x, dsc * q[it].y, dsc * q[it].z, dsc * q[it].w);
                if (tanh_key) {
                    dk[it].x *= 1.f - k[b][it].x * k[b][it].x; dk[it].y *= 1.f - k[b][it].y * k[b][it].y;
                    dk[it].z *= 1.f - k[b][it].z * k[b][it].z; dk[it].w *= 1.f - k[b][it].w * k[b][it].w;
                }
                if (key_is_value) { dv[it].x += dk[it].x; dv[it].y += dk[it].y; dv[it].z += dk[it].z; dv[it].w += dk[it].w; }
            }
            store_row<IT>(dvalue + (n * S + s) * (int64_t)D, dv, D, lane);
            if (!key_is_value) store_row<IT>(dkeypre + (n * S + s) * (int64_t)D, dk, D, lane);
        }
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) *reinterpret_cast<float4*>(&part[w][(it * 64 + lane) * 4]) = dq[it];
    __syncthreads();
    for (int c = threadIdx.x * 4; c < D; c += 1024) {
        const float4 a0 = *reinterpret_cast<const float4*>(&part[0][c]), a1 = *reinterpret_cast<const float4*>(&part[1][c]);
        const float4 a2 = *reinterpret_cast<const float4*>(&part[2][c]), a3 = *reinterpret_cast<const float4*>(&part[3][c]);
        float4 r;
        r.x = ((a0.x + a1.x) + a2.x) + a3.x; r.y = ((a0.y + a1.y) + a2.y) + a3.y;
        r.z = ((a0.z + a1.z) + a2.z) + a3.z; r.w = ((a0.w + a1.w) + a2.w) + a3.w;
        *reinterpret_cast<float4*>(dq_part + n * (int64_t)D + c) = r;
    }
}

// ------------------------------------------------------------------ scoring (+ cross entropy)
// logits[b,c] = cdd[b,c].user[b] / sqrt(D)   (cdd dense or gathered by idx from table)
// loss_b[b] = -log_softmax(logits[b])[label[b]]   (optional)
template <int IT>
__global__ __launch_bounds__(256) void score_fwd_kernel(const float* __restrict__ cdd, const float* __restrict__ table,
                                                        const int64_t* __restrict__ idx, int64_t nrows,
                                                        const float* __restrict__ user,
                                                        const int64_t* __restrict__ label, float* __restrict__ logits,
                                                        float* __restrict__ loss_b, int64_t B, int C, int D, int* __restrict__ oob) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (b >= B) return;
    float4 u[IT];
    load_row<IT>(u, user + b * (int64_t)D, D, lane);
    const float sq = sqrtf((float)D);
    float mylogit = -INFINITY;
    for (int c = 0; c < C; ++c) {
        float4 v[IT];
        load_row<IT>(v, row_ptr(cdd, table, idx, nrows, b, C, c, D, oob), D, lane);
        const float x = dot_rows<IT>(v, u) / sq;
        if (lane == (c & 63)) mylogit = x;
        if ((c & 63) == 63 || c == C - 1) {                  // flush 64 logits at a time: any C (the loss path needs C <= 64)
            const int base = c & ~63;
            if (base + lane <= c) logits[b * C + base + lane] = mylogit;
        }
    }
    if (loss_b) {
        const float m = wave_max(mylogit);
        const float e = lane < C ? expf(mylogit - m) : 0.f;
        const float sum = wave_sum(e);
        const int lab = (int)label[b];
        const float xl = __shfl(mylogit, lab, 64);
        if (lane == 0) loss_b[b] = -((xl - m) - logf(sum));
    }
}

// d_logit[b,c] = (softmax(logits[b])[c] - [c == label]) * gscale,  gscale = grad_out / B
// d_cdd[b,c] = d_logit/sqrt(D) * user[b] ; d_user[b] = sum_c d_logit/sqrt(D) * cdd[b,c]
template <int IT>
__global__ __launch_bounds__(256) void score_bwd_kernel(const float* __restrict__ cdd, const float* __restrict__ user,
                                                        const int64_t* __restrict__ label, const float* __restrict__ logits,
                                                        const float* __restrict__ grad_out,
                                                        const float* __restrict__ dlogits, float* __restrict__ d_cdd,
                                                        float* __restrict__ d_user, int64_t B, int C, int D) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (b >= B) return;
    const float sq = sqrtf((float)D);
    float dl;
    if (dlogits) {                      // plain _compute_logits backward: upstream d(logits) given
        dl = lane < C ? dlogits[b * C + lane] / sq : 0.f;
    } else {                            // fused mean cross-entropy backward
        const float x = lane < C ? logits[b * C + lane] : -INFINITY;
        const float m = wave_max(x);
        const float e = lane < C ? expf(x - m) : 0.f;
        const float sum = wave_sum(e);
        const float gs = grad_out[0] / (float)B;
        dl = lane < C ? (e / sum - (lane == (int)label[b] ? 1.f : 0.f)) * gs / sq : 0.f;
    }
    float4 u[IT], du[IT];
    load_row<IT>(u, user + b * (int64_t)D, D, lane);
#pragma unroll
    for (int it = 0; it < IT; ++it) du[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
        const float d = __shfl(dl, c, 64);
        float4 v[IT], dv[IT];
        load_row<IT>(v, cdd + (b * C + c) * (int64_t)D, D, lane);
        axpy_rows<IT>(du, d, v);
#pragma unroll
        for (int it = 0; it < IT; ++it) dv[it] = make_float4(d * u[it].x, d * u[it].y, d * u[it].z, d * u[it].w);
        store_row<IT>(d_cdd + (b * C + c) * (int64_t)D, dv, D, lane);
    }
    store_row<IT>(d_user + b * (int64_t)D, du, D, lane);
}

// ------------------------------------------------------------------ unmasked sum / mean over positions
template <int IT>
__global__ __launch_bounds__(256) void seq_reduce_kernel(const float* __restrict__ x, const float* __restrict__ table,
                                                         const int64_t* __restrict__ idx, int64_t nrows,
                                                         float* __restrict__ out, int64_t nseq, int S, int D, int mean, int* __restrict__ oob) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (n >= nseq) return;
    float4 acc[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < S; ++s) {
        float4 v[IT];
        load_row<IT>(v, row_ptr(x, table, idx, nrows, n, S, s, D, oob), D, lane);
#pragma unroll
        for (int it = 0; it < IT; ++it) { acc[it].x += v[it].x; acc[it].y += v[it].y; acc[it].z += v[it].z; acc[it].w += v[it].w; }
    }
    if (mean) {
        const float fs = (float)S;
#pragma unroll
        for (int it = 0; it < IT; ++it) { acc[it].x /= fs; acc[it].y /= fs; acc[it].z /= fs; acc[it].w /= fs; }
    }
    store_row<IT>(out + n * (int64_t)D, acc, D, lane);
}
// dx[n,s,:] = dout[n,:] * scale
template <int IT>
__global__ __launch_bounds__(256) void seq_broadcast_kernel(const float* __restrict__ dout, float* __restrict__ dx,
                                                            int64_t nseq, int S, int D, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * WPB + (threadIdx.x >> 6);
    if (n >= nseq) return;
    float4 g[IT];
    load_row<IT>(g, dout + n * (int64_t)D, D, lane);
#pragma unroll
    for (int it = 0; it < IT; ++it) { g[it].x *= scale; g[it].y *= scale; g[it].z *= scale; g[it].w *= scale; }
    for (int s = 0; s < S; ++s) store_row<IT>(dx + (n * S + s) * (int64_t)D, g, D, lane);
}

// ------------------------------------------------------------------ host launchers
#define NR_SEQ_DISPATCH(KERNEL, GRID, ...)                                                          \
    do {                                                                                            \
        if (D <= 256)       hipLaunchKernelGGL(KERNEL<1>, GRID, dim3(256), 0, s, __VA_ARGS__);      \
        else if (D <= 512)  hipLaunchKernelGGL(KERNEL<2>, GRID, dim3(256), 0, s, __VA_ARGS__);      \
        else if (D <= 1024) hipLaunchKernelGGL(KERNEL<4>, GRID, dim3(256), 0, s, __VA_ARGS__);      \
        else                hipLaunchKernelGGL(KERNEL<8>, GRID, dim3(256), 0, s, __VA_ARGS__);      \
    } while (0)

#define NR_POOL_DISPATCH(KERNEL, GRID, ...)                                                         \
    do {                                                                                            \
        if (S <= 64) {                                                                              \
            if (D <= 256)       hipLaunchKernelGGL((KERNEL<1, 1>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else if (D <= 512)  hipLaunchKernelGGL((KERNEL<2, 1>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else if (D <= 1024) hipLaunchKernelGGL((KERNEL<4, 1>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else                hipLaunchKernelGGL((KERNEL<8, 1>), GRID, dim3(256), 0, s, __VA_ARGS__); \
        } else {                                                                                    \
            if (D <= 256)       hipLaunchKernelGGL((KERNEL<1, 2>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else if (D <= 512)  hipLaunchKernelGGL((KERNEL<2, 2>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else if (D <= 1024) hipLaunchKernelGGL((KERNEL<4, 2>), GRID, dim3(256), 0, s, __VA_ARGS__); \
            else                hipLaunchKernelGGL((KERNEL<8, 2>), GRID, dim3(256), 0, s, __VA_ARGS__); \
        }                                                                                           \
    } while (0)

static int check_seq_dims(const char* what, int S, int D, int smax = 64) {
    NR_UNSUPPORTED(S < 1 || S > smax, "%s: %d positions per sequence not in [1,%d]", what, S, smax);
    NR_UNSUPPORTED(D % 4 != 0 || D > 2048 || D < 4, "%s: D=%d must be a multiple of 4 in [4,2048]", what, D);
    return NR_OK;
}

int pool_fwd(const float* key, const float* value, const float* table, const int64_t* idx, int64_t nrows, const float* query,
             const int64_t* mask, float* out, float* prob, int64_t nseq, int S, int D, hipStream_t s, const int* offs, int64_t rows) {
    if (nseq == 0) return NR_OK;
    if (int rc = check_seq_dims("pooling", S, D, 128)) return rc;
    NR_UNSUPPORTED(offs && idx, "pooling: packed sequences and table gather are exclusive");
    ProfScope prof(FAM_POOL, s, (double)(offs ? rows : nseq * S) * (2.0 * D * 4) + (double)nseq * D * 4);
    const dim3 grid((unsigned)nseq);
    NR_POOL_DISPATCH(pool_fwd_kernel, grid, key, value, table, idx, nrows, query, mask, out, prob, nseq, S, D, offs, idx ? index_flag_ptr() : nullptr);
    return check_launch("pool_fwd");
}

int pool_bwd(const float* key, const float* value, const float* query, const float* prob, const float* dout, float* dkeypre,
             float* dvalue, float* dq_part, int64_t nseq, int S, int D, int tanh_key, int key_is_value, hipStream_t s,
             const int* offs, int64_t rows, const float* dout1, int64_t split) {
    if (nseq == 0) return NR_OK;
    if (int rc = check_seq_dims("pooling bwd", S, D, 128)) return rc;
    ProfScope prof(FAM_POOL, s, (double)(offs ? rows : nseq * S) * (4.0 * D * 4));
    const dim3 grid((unsigned)nseq);
    NR_POOL_DISPATCH(pool_bwd_kernel, grid, key, value, query, prob, dout, dkeypre, dvalue, dq_part, nseq, S, D, tanh_key, key_is_value, offs, dout1, split);
    return check_launch("pool_bwd");
}

int score_fwd(const float* cdd, const float* table, const int64_t* idx, int64_t nrows, const float* user, const int64_t* label, float* logits,
              float* loss_b, int64_t B, int C, int D, hipStream_t s) {
    if (B == 0) return NR_OK;
    if (int rc = check_seq_dims("scoring", C, D, loss_b ? 64 : 65536)) return rc;       // plain logits: any C; fused CE: one logit per lane
    ProfScope prof(FAM_SCORE, s, (double)B * ((C + 1.0) * D * 4 + C * 4 + (idx ? C * 8 : 0)));
    const dim3 grid(cdiv(B, WPB));
    NR_SEQ_DISPATCH(score_fwd_kernel, grid, cdd, table, idx, nrows, user, label, logits, loss_b, B, C, D, idx ? index_flag_ptr() : nullptr);
    return check_launch("score_fwd");
}

int score_bwd(const float* cdd, const float* user, const int64_t* label, const float* logits, const float* grad_out,
              const float* dlogits, float* d_cdd, float* d_user, int64_t B, int C, int D, hipStream_t s) {
    if (B == 0) return NR_OK;
    if (int rc = check_seq_dims("scoring bwd", C, D)) return rc;
    ProfScope prof(FAM_SCORE, s, (double)B * (2.0 * (C + 1.0) * D * 4));
    const dim3 grid(cdiv(B, WPB));
    NR_SEQ_DISPATCH(score_bwd_kernel, grid, cdd, user, label, logits, grad_out, dlogits, d_cdd, d_user, B, C, D);
    return check_launch("score_bwd");
}

int seq_reduce(const float* x, const float* table, const int64_t* idx, int64_t nrows, float* out, int64_t nseq, int S, int D, int mean,
               hipStream_t s) {
    if (nseq == 0) return NR_OK;
    NR_UNSUPPORTED(D % 4 != 0 || D > 2048 || D < 4, "seq_reduce: D=%d must be a multiple of 4 in [4,2048]", D);
    ProfScope prof(FAM_POOL, s, (double)nseq * (S + 1.0) * D * 4);
    const dim3 grid(cdiv(nseq, WPB));
    NR_SEQ_DISPATCH(seq_reduce_kernel, grid, x, table, idx, nrows, out, nseq, S, D, mean, idx ? index_flag_ptr() : nullptr);
    return check_launch("seq_reduce");
}

int seq_broadcast(const float* dout, float* dx, int64_t nseq, int S, int D, float scale, hipStream_t s) {
    if (nseq == 0) return NR_OK;
    NR_UNSUPPORTED(D % 4 != 0 || D > 2048 || D < 4, "seq_broadcast: D=%d must be a multiple of 4 in [4,2048]", D);
    ProfScope prof(FAM_POOL, s, (double)nseq * (S + 1.0) * D * 4);
    const dim3 grid(cdiv(nseq, WPB));
    NR_SEQ_DISPATCH(seq_broadcast_kernel, grid, dout, dx, nseq, S, D, scale);
    return check_launch("seq_broadcast");
}

}  // namespace nr
