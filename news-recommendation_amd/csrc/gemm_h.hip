// bf16-in-HBM GEMM for the bf16 build of the BERT-sized layers (parity bar 1e-2):
//
//   C[m,n] = epi( sum_k A[m,k] * B[n,k] + bias[n] )      A [M,K], B [N,K]: bf16, k contiguous; f32 accumulate
//
// Every product of the layer is brought to this one "NT" form: forward x . W^T as is; dX = dY . W with a
// transposed bf16 copy of W as B; dW = dY^T . X with transposed bf16 copies of dY and X (rowops: transpose_cast).
// v_mfma_f32_32x32x16_bf16, workgroup tile 128 x 128 x 64, 4 waves 2x2 (wave tile 64 x 64), two LDS stages.
// Operand tiles travel HBM -> LDS directly (global_load_lds, 16 B per lane: no staging registers, no conversion):
// one wave instruction fills 1 KiB = 8 tile rows of 128 B.  The LDS image is lane-linear, so the bank swizzle sits on
// the SOURCE address: 16-byte chunk c of tile row r lives at chunk c ^ ((r >> 1) & 7) — the 16 lanes of one
// ds_read_b128 phase (16 consecutive rows, same logical chunk) then cover all 64 banks exactly once.
// K must be a multiple of 64 (callers zero-pad the reduction dimension of transposed copies); M / N edges read a
// clamped row and are not stored.
// Measured (tools/gemm_h_bench.py, random operands): 775-800 TFLOP/s at 4096^3 / 8192^3, 540-750 at the BERT-base
// shapes (R = 14 080 rows, K = 768 / 3072; one output).  That is the ceiling of this structure (two barriers per k-tile,
// two workgroups per CU: the other workgroup's MFMAs are the only thing that overlaps a workgroup's wait for its fills).
// Tried and dropped: BK = 32 with three stages, counted vmcnt + raw s_barrier and three workgroups per CU (two k-tiles
// in flight per workgroup): 646 / 690 TFLOP/s at the cubes — half the MFMA work per barrier costs more than the deeper
// queue buys.  Past ~0.9 PFLOP/s needs the 256 x 256, 8-wave, multi-phase schedule (next round).
#include "common.h"
#include "gemm.h"
#include <cstdlib>
#include <mutex>

namespace nr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

constexpr int HM = 128, HN = 128, HK = 64;
constexpr int H_TILE_BYTES = HM * HK * 2;            // 16 KiB per operand tile
constexpr int H_STAGE_BYTES = 2 * H_TILE_BYTES;      // A + B
constexpr int H_LDS_BYTES = 4 * 64 * 68 * 4;          // the epilogue's four [64][68] f32 tiles (> the two operand stages)

__device__ __forceinline__ unsigned short f2bf(float v) {
    const f32x2v t = {v, 0.f};
    return (unsigned short)(__builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2v)) & 0xffffu);   // RNE (v_cvt_pk_bf16_f32)
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    const f32x2v t = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bf16x2v));
}
__device__ __forceinline__ float bf2f(unsigned short v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }

// 256 bytes of zeros per GPU: the source of a reduction-major tile's k-rows past K
static __device__ __attribute__((aligned(256))) unsigned short g_zero_page[128];
static const unsigned short* zero_page_ptr() {
    static std::mutex mu;
    static const unsigned short* per_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!per_dev[dev]) {
        unsigned short* ptr = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(g_zero_page)) == hipSuccess && ptr) per_dev[dev] = ptr;
        else (void)hipGetLastError();
    }
    return per_dev[dev];
}

// Reduction-major operands (A_RM: A[k][m], B_RM: B[k][n] — dY and X of dW = dY^T X, the weight of dX = dY W) need no
// transposed copy in HBM: their 64 (k) x 128 tile goes to LDS as it lies in memory (256-byte rows, 16-byte chunk ch of
// row r at chunk ch ^ (((r & 3) << 2) | ((r >> 2) & 3)), applied to the SOURCE address as for the k-contiguous image) and
// the MFMA fragments come out of it through ds_read_b64_tr_b16: per 16-lane group a 4 (k) x 16 (column) block, delivered
// column-major, lane 4q+p supplying the address of row q, columns 4p .. 4p+3 (measured lane map: tools/tr_probe.hip) —
// two reads make the eight consecutive k of one lane's 32x32x16 operand.  k-rows past K are read from a page of zeros,
// so the reduction length need not be padded.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ int rm_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

template <bool A_RM, bool B_RM>
__global__ __launch_bounds__(256, 2) void gemm_h_kernel(const GemmHParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char hsm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int tiles_n = (p.N + HN - 1) / HN;
    const int tiles_m = (p.M + HM - 1) / HM;
    const int ntile = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {                                       // XCD-aware, bijective, slice-major over the whole grid (see gemm_f32_kernel)
        const int G = ntile * p.ksplit;
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int ks = bid / ntile;
    bid -= ks * ntile;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * HM, n0 = tn * HN;

    const int kt_total = (p.K + HK - 1) / HK;
    const int kt_per = (kt_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);

    // per-lane source pointers of this wave's 4 + 4 fills per k-tile: fill c covers tile rows 8c .. 8c+7,
    // lane -> row 8c + (lane >> 3), LDS chunk lane & 7 = logical chunk ^ ((row >> 1) & 7)
    // (reduction-major: fill c covers k-rows 4c .. 4c+3 of 256 B, lane -> k-row 4c + (lane >> 4), LDS chunk lane & 15 =
    // logical chunk ^ rm_swz(k-row); columns past the edge read the last valid chunk and are never stored)
    const unsigned short* ga[4];
    const unsigned short* gb[4];
    const unsigned short* zsrc = p.zero + (lane & 15) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = wid + 4 * j;
        if (A_RM) {
            const int krow = 4 * c + (lane >> 4);
            const int col = min(m0 + (((lane & 15) ^ rm_swz(krow)) << 3), p.M - 8);
            ga[j] = p.A + (int64_t)krow * p.lda + col;
        } else {
            const int row = 8 * c + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            ga[j] = p.A + (int64_t)min(m0 + row, p.M - 1) * p.lda + chunk * 8;
        }
        if (B_RM) {
            const int krow = 4 * c + (lane >> 4);
            const int col = min(n0 + (((lane & 15) ^ rm_swz(krow)) << 3), p.N - 8);
            gb[j] = p.B + (int64_t)krow * p.ldb + col;
        } else {
            const int row = 8 * c + (lane >> 3);
            const int chunk = (lane & 7) ^ ((row >> 1) & 7);
            gb[j] = p.B + (int64_t)min(n0 + row, p.N - 1) * p.ldb + chunk * 8;
        }
    }
    // Only the LAST k-tile of a reduction-major operand can hold rows past K: `tail` selects the zero page for those lanes there, so the
    // loop proper carries no predicate (a per-lane pointer select in the loop became exec-masked branches around every fill).
    auto stage = [&](int buf, int kt, bool tail) {
        unsigned char* As = hsm + buf * H_STAGE_BYTES;
        unsigned char* Bs = As + H_TILE_BYTES;
        const int k0 = kt * HK;
        const int64_t ka = A_RM ? (int64_t)k0 * p.lda : (int64_t)k0, kb = B_RM ? (int64_t)k0 * p.ldb : (int64_t)k0;      // wave-uniform
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wid + 4 * j;
            const unsigned short* src = ga[j] + ka;
            if (A_RM && tail && k0 + 4 * c + (lane >> 4) >= p.K) src = zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(As + c * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wid + 4 * j;
            const unsigned short* src = gb[j] + kb;
            if (B_RM && tail && k0 + 4 * c + (lane >> 4) >= p.K) src = zsrc;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(Bs + c * 1024), 16, 0, 0);
        }
    };
    const bool ragged = (A_RM || B_RM) && (p.K % HK) != 0;         // then the last k-tile of the whole product is partial

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment addresses: row (wr*64 + i*32 + l31) of A / (wc*64 + j*32 + l31) of B, logical chunk 2*s + h of k-step s
    const int sw = (l31 >> 1) & 7;                        // tile-row offsets are multiples of 16: swizzle depends on l31 only
    const int arow = (wr * 64 + l31) * 128, brow = (wc * 64 + l31) * 128;
    // transposed reads of a reduction-major tile: byte offset of this lane's address for 32-column block i, k-quad t of a
    // k-step (k-step s adds 16 rows = 4096 bytes; rm_swz of row 16s + 8h + 4t + q does not depend on s)
    int tra[2][2], trb[2][2];
    {
        const int g = lane >> 4, q = (lane & 15) >> 2, pq = lane & 3;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int row = 8 * (g >> 1) + 4 * t + q;
            const int sz = rm_swz(row);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ca = wr * 64 + i * 32 + 16 * (g & 1) + 4 * pq, cb = wc * 64 + i * 32 + 16 * (g & 1) + 4 * pq;
                tra[i][t] = row * 256 + (((ca >> 3) ^ sz) << 4) + (ca & 7) * 2;
                trb[i][t] = row * 256 + (((cb >> 3) ^ sz) << 4) + (cb & 7) * 2;
            }
        }
    }
    auto tr8 = [](const unsigned char* base, int o0, int o1) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
        const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    };
    // fragments of k-step s+1 are requested before the MFMAs of k-step s are issued (two register sets): a wave's LDS
    // latency then hides behind its own matrix work instead of only behind the partner workgroup's
    auto compute = [&](int buf) {
        const unsigned char* As = hsm + buf * H_STAGE_BYTES;
        const unsigned char* Bs = As + H_TILE_BYTES;
        bf16x8 fa[2][2], fb[2][2];
        auto frags = [&](int s, bf16x8 (&a)[2], bf16x8 (&b)[2]) {
            const int co = ((2 * s + h) ^ sw) << 4;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (A_RM) a[i] = tr8(As + s * 4096, tra[i][0], tra[i][1]);
                else      a[i] = *reinterpret_cast<const bf16x8*>(As + arow + i * 32 * 128 + co);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (B_RM) b[j] = tr8(Bs + s * 4096, trb[j][0], trb[j][1]);
                else      b[j] = *reinterpret_cast<const bf16x8*>(Bs + brow + j * 32 * 128 + co);
            }
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s + 1 < 4) frags(s + 1, fa[(s + 1) & 1], fb[(s + 1) & 1]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s & 1][i], fb[s & 1][j], acc[i][j], 0, 0, 0);
        }
    };

    if (kt_begin < kt_end) {
        const bool last_is_tail = ragged && kt_end == kt_total;
        stage(0, kt_begin, last_is_tail && kt_begin == kt_end - 1);
        __syncthreads();                                   // drains the fills (vmcnt(0)) and publishes them
        int cur = 0;
        for (int kt = kt_begin; kt < kt_end - 2; ++kt) {
            stage(cur ^ 1, kt + 1, false);                 // next tile flies while this one is multiplied
            compute(cur);
            __syncthreads();
            cur ^= 1;
        }
        if (kt_end - kt_begin >= 2) {
            stage(cur ^ 1, kt_end - 1, last_is_tail);
            compute(cur);
            __syncthreads();
            cur ^= 1;
        }
        compute(cur);
    }

    // ---- epilogue through LDS: each wave parks its 64 x 64 f32 tile in its own [64][68] region, then walks it row by
    // row so that every global access is a 16-byte piece of a full 128-B line (bf16 rows of 64 columns) — the accumulator
    // layout itself would store 2-byte elements 64 B at a time.  The epilogue math runs in this row-contiguous phase.
    constexpr int EP = 68;
    __syncthreads();                                       // every wave is done reading the operand stages
    float* tile = reinterpret_cast<float*>(hsm) + wid * 64 * EP;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                tile[(i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * EP + j * 32 + l31] = acc[i][j][r];
    __syncthreads();
    const bool partial = p.ksplit > 1;
    float* C = partial ? p.C + (int64_t)ks * p.split_stride : p.C;
    const int64_t ldc = partial ? p.N : p.ldc;
    const int epi = partial ? EPI_NONE : p.epi;
    const bool ub = !partial && p.bias != nullptr;
    unsigned short* Ch = partial ? nullptr : p.Ch;
    unsigned short* ChT = partial ? nullptr : p.ChT;
    const int mw = m0 + wr * 64, nw = n0 + wc * 64;
    const int cc = (lane & 7) * 8;                          // 8 columns per lane, 8 rows per wave instruction
    const int n = nw + cc;
    const bool nfull = p.vec && n + 7 < p.N;
    float bv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bv[e] = (ub && n + e < p.N) ? p.bias[n + e] : 0.f;
#pragma unroll 2
    for (int it = 0; it < 8; ++it) {
        const int rr = it * 8 + (lane >> 3);
        const int m = mw + rr;
        if (m >= p.M || n >= p.N) continue;                 // (their LDS rows keep raw accumulators: the transposed pass masks them)
        float v[8];
        {
            const float4 t0 = *reinterpret_cast<const float4*>(tile + rr * EP + cc), t1 = *reinterpret_cast<const float4*>(tile + rr * EP + cc + 4);
            v[0] = t0.x + bv[0]; v[1] = t0.y + bv[1]; v[2] = t0.z + bv[2]; v[3] = t0.w + bv[3];
            v[4] = t1.x + bv[4]; v[5] = t1.y + bv[5]; v[6] = t1.z + bv[6]; v[7] = t1.w + bv[7];
        }
        if (epi == EPI_GELU) {
            if (p.out2h) {
                unsigned short* zp = p.out2h + (int64_t)m * p.ldout2h + n;
                if (nfull) *reinterpret_cast<uint4*>(zp) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
                else for (int e = 0; e < 8; ++e) if (n + e < p.N) zp[e] = f2bf(v[e]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = gelu_f(v[e]);
        } else if (epi == EPI_ADD_AUX) {
            const float* ap = p.aux + (int64_t)m * p.ldaux + n;
            if (nfull) {
                const float4 a0 = *reinterpret_cast<const float4*>(ap), a1 = *reinterpret_cast<const float4*>(ap + 4);
                v[0] += a0.x; v[1] += a0.y; v[2] += a0.z; v[3] += a0.w; v[4] += a1.x; v[5] += a1.y; v[6] += a1.z; v[7] += a1.w;
            } else for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] += ap[e];
        } else if (epi == EPI_MUL_GELU_GRAD) {
            const unsigned short* ap = p.auxh + (int64_t)m * p.ldauxh + n;
            if (nfull) {
                const uint4 u = *reinterpret_cast<const uint4*>(ap);
                const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] *= gelu_grad_f(__builtin_bit_cast(float, w[e] << 16));
                    v[2 * e + 1] *= gelu_grad_f(__builtin_bit_cast(float, w[e] & 0xffff0000u));
                }
            } else for (int e = 0; e < 8; ++e) if (n + e < p.N) v[e] *= gelu_grad_f(bf2f(ap[e]));
        } else if (epi == EPI_TANH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = tanhf(v[e]);
        } else if (epi == EPI_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
        }
        if (C) {
            float* cp = C + (int64_t)m * ldc + n;
            if (nfull) {
                *reinterpret_cast<float4*>(cp) = make_float4(v[0], v[1], v[2], v[3]);
                *reinterpret_cast<float4*>(cp + 4) = make_float4(v[4], v[5], v[6], v[7]);
            } else for (int e = 0; e < 8; ++e) if (n + e < p.N) cp[e] = v[e];
        }
        if (Ch) {
            unsigned short* hp = Ch + (int64_t)m * p.ldch + n;
            if (nfull) *reinterpret_cast<uint4*>(hp) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
            else for (int e = 0; e < 8; ++e) if (n + e < p.N) hp[e] = f2bf(v[e]);
        }
        if (ChT) {                                         // park the finished values for the transposed pass below
            *reinterpret_cast<float4*>(tile + rr * EP + cc) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4*>(tile + rr * EP + cc + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
    }
    if (ChT) {
        // transposed copy [n][m]: lane = (column of an 8-column group, 8-row chunk) -> one wave instruction writes 8 output
        // rows x 128 contiguous bytes; rows m >= M of the tile are written as zeros up to the padded pitch
        __syncthreads();
        const int c8 = lane >> 3, mq = lane & 7;
        const int mbase = mw + mq * 8;
#pragma unroll 2
        for (int g8 = 0; g8 < 8; ++g8) {
            const int col = g8 * 8 + c8;
            const int nn = nw + col;
            if (nn >= p.N || mbase >= p.ldcht) continue;
            float e[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) e[k] = (mbase + k < p.M) ? tile[(mq * 8 + k) * EP + col] : 0.f;
            *reinterpret_cast<uint4*>(ChT + (int64_t)nn * p.ldcht + mbase) = make_uint4(pack2(e[0], e[1]), pack2(e[2], e[3]), pack2(e[4], e[5]), pack2(e[6], e[7]));
        }
    }
}

// sum split-K slabs in a fixed order (deterministic) into C [M,N] (ldc)
__global__ __launch_bounds__(256) void splitk_sum_kernel(const float* __restrict__ slabs, int ks, int64_t stride, float* __restrict__ C,
                                                         int64_t ldc, int M, int N) {
    const int64_t total = (int64_t)M * N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / N), n = (int)(idx - (int64_t)m * N);
        float v = 0.f;
        for (int s = 0; s < ks; ++s) v += slabs[(int64_t)s * stride + idx];
        C[(int64_t)m * ldc + n] = v;
    }
}

int choose_ksplit_h(int M, int N, int K) {
    static const int forced = getenv("NR_GEMM_H_KSPLIT") ? atoi(getenv("NR_GEMM_H_KSPLIT")) : 0;     // tools/gemm_h_bench.py sweep
    if (forced > 0) return std::max(1, std::min(forced, K / HK));
    const int tiles = cdiv(M, HM) * cdiv(N, HN);
    const int kt = (K + HK - 1) / HK;
    if (tiles >= 256 || kt < 8) return 1;
    // one full round of the chip (256 CUs x 2 resident workgroups) and no more: swept on MI355X at the BERT-base dW
    // shapes (K = 14 080): 144 tiles x 3 = 432 workgroups 712 TFLOP/s (x2 545, x4 577, x13 504); 108 tiles x 4 667; 36 x 13 417-730
    int ks = std::max(1, 512 / tiles);
    ks = std::min(ks, std::max(1, kt / 4));
    return std::max(ks, 1);
}

size_t gemm_h_workspace_bytes(int M, int N, int ksplit) { return ksplit > 1 ? (size_t)ksplit * M * N * sizeof(float) : 0; }

bool gemm_h_supported(int M, int N, int K) { return M >= 1 && N >= 1 && K >= HK && K % HK == 0; }       // the NT form

int gemm_h(GemmHParams p, float* workspace, size_t ws_bytes, hipStream_t s) {
    NR_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "gemm_h: negative dims");
    if (p.M == 0 || p.N == 0) return NR_OK;
    NR_CHECK_ARG(p.A && p.B && (p.C || p.Ch), "gemm_h: null operand");
    NR_CHECK_ARG(!p.ChT || (p.ldcht % 64 == 0 && p.ldcht >= p.M && (reinterpret_cast<uintptr_t>(p.ChT) & 15) == 0), "gemm_h: transposed output needs a 64-multiple pitch >= M");
    NR_UNSUPPORTED(p.a_rm && !p.b_rm, "gemm_h: layout (A reduction-major, B k-contiguous) not built");
    // a k-contiguous operand is read 64 k at a time with no edge handling; reduction-major tiles take their rows past K from zeros
    NR_UNSUPPORTED(p.K < 1 || ((!p.a_rm || !p.b_rm) && p.K % HK != 0), "gemm_h: K=%d must be a positive multiple of %d unless both operands are reduction-major", p.K, HK);
    NR_UNSUPPORTED((p.a_rm && (p.M % 8 != 0 || p.M < 8)) || (p.b_rm && (p.N % 8 != 0 || p.N < 8)), "gemm_h: a reduction-major operand needs M / N to be a multiple of 8");
    if (p.a_rm || p.b_rm) {
        p.zero = zero_page_ptr();
        if (!p.zero) { set_error("gemm_h: the per-GPU zero page is unavailable (hipGetSymbolAddress failed)"); return NR_ERR_LAUNCH; }
    }
    auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    // 16-byte epilogue accesses need aligned bases and row pitches; anything else takes the element-wise path
    p.vec = (!p.C || (p.ldc % 4 == 0 && al16(p.C))) && (!p.Ch || (p.ldch % 8 == 0 && al16(p.Ch))) && (!p.aux || (p.ldaux % 4 == 0 && al16(p.aux))) &&
            (!p.auxh || (p.ldauxh % 8 == 0 && al16(p.auxh))) && (!p.out2h || (p.ldout2h % 8 == 0 && al16(p.out2h))) && (p.N % 8 == 0);
    NR_CHECK_ARG(p.lda % 8 == 0 && p.ldb % 8 == 0 && ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B)) & 15) == 0,
                 "gemm_h: operands must be 16-byte aligned with row pitches that are multiples of 8 elements");
    if (p.ksplit < 1) p.ksplit = 1;
    float* final_c = p.C;
    const int64_t final_ldc = p.ldc;
    if (p.ksplit > 1) {
        NR_CHECK_ARG(p.C && !p.Ch && !p.ChT && p.epi == EPI_NONE && !p.bias, "gemm_h: split-K writes plain f32 results only");
        const size_t need = gemm_h_workspace_bytes(p.M, p.N, p.ksplit);
        if (!workspace || ws_bytes < need) { set_error("gemm_h: split-K workspace too small (%zu < %zu)", ws_bytes, need); return NR_ERR_WORKSPACE; }
        p.C = workspace;
        p.split_stride = (int64_t)p.M * p.N;
    }
    ProfScope prof(FAM_GEMM, s, 2.0 * p.M * p.N * p.K);
    static bool once = false;
    if (!once) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, H_LDS_BYTES);
        once = true;
    }
    const int tiles = cdiv(p.M, HM) * cdiv(p.N, HN);
    if (p.a_rm)      hipLaunchKernelGGL((gemm_h_kernel<true, true>), dim3(tiles * p.ksplit), dim3(256), H_LDS_BYTES, s, p);
    else if (p.b_rm) hipLaunchKernelGGL((gemm_h_kernel<false, true>), dim3(tiles * p.ksplit), dim3(256), H_LDS_BYTES, s, p);
    else             hipLaunchKernelGGL((gemm_h_kernel<false, false>), dim3(tiles * p.ksplit), dim3(256), H_LDS_BYTES, s, p);
    if (int rc = check_launch("gemm_h")) return rc;
    if (p.ksplit > 1) {
        const int64_t total = (int64_t)p.M * p.N;
        const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_sum_kernel, dim3(grid), dim3(256), 0, s, (const float*)workspace, p.ksplit, p.split_stride, final_c, final_ldc, p.M, p.N);
        return check_launch("splitk_sum");
    }
    return NR_OK;
}

}  // namespace nr
