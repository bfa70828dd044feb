// fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 fma chain).
//
//   C[m,n] = epi( sum_k A(m,k) * B(n,k) + bias[n] )
//
// Operand layouts (template flags):
//   A_KC : A(m,k) = A[m*lda + k]   (k contiguous; activations / dY for dX)
//  !A_KC : A(m,k) = A[k*lda + m]   (reduction-major; dY^T for dW)
//   B_KC : B(n,k) = B[n*ldb + k]   (nn.Linear weight [N,K])
//  !B_KC : B(n,k) = B[k*ldb + n]   (weight used transposed for dX; X for dW)
//
// Block: 256 threads = 4 waves (2x2).  Each wave owns a (32*WM) x (32*NJ) sub-tile = WM x NJ MFMA
// tiles of 32x32, so the workgroup tile is (64*WM) x (64*NJ) x 32:
//   WM=2 NJ=1  128 x  64   activations x weights with N in {300, 900, 1200} (64 wastes least columns)
//   WM=2 NJ=2  128 x 128   N a multiple of 128
//   WM=5 NJ=1  320 x  64   dW = dY^T X: M is 300/900/1200 (one/three/four tiles, 6.7 % padding instead of
//                          up to 28 % with 128-row tiles) and it issues 4.8 instead of 6 global loads per 32 MFMAs
// One LDS buffer + register prefetch of the next k-tile (more workgroups per CU beat double buffering
// here), LDS fragments software-pipelined by hand one k-quad ahead of the MFMAs.  Measured on MI355X
// (tools/peaks.hip): a wave's global loads cost matrix-pipe time by themselves (2 dwordx4 loads per
// 32 MFMAs: 156 -> 129 TFLOP/s with no data dependence at all), so loads per MFMA is the quantity the
// tile shapes minimise.
// K-contiguous tiles live in LDS as [row][32+4] and are read with ds_read_b128
// (lane half h takes k = 8*kq+4*h .. +3: the k permutation is free because
// both operands use the same one); reduction-major tiles live as [k][cols+4]
// and are read with conflict-free ds_read_b32.
//
// conv mode (kernel_size=3, padding=1 Conv1d as an implicit GEMM): the
// im2col row of token (m,t) is the 3E contiguous floats starting at
// x[(r-1)*E] (r = m*T+t), with the first/last E zeroed at t==0 / t==T-1.
#include "common.h"
#include "gemm.h"
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace nr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 32;
constexpr int KC_LD = BK + 4;        // floats per LDS row of a k-contiguous tile

template <int COLS> struct RedMajor { static constexpr int LD = COLS + 4; };

__device__ __forceinline__ bool conv_masked(int r, int c, int T, int E) {
    // r: token row (m*T+t); c: im2col column in [0,3E)
    const int t = r % T;
    return (t == 0 && c < E) || (t == T - 1 && c >= 2 * E);
}

// ---- global -> register staging -------------------------------------------------
// Per-thread loader state is computed ONCE per tile (row pointers, validity, conv border flags);
// per k-tile only `k < K` and the pointer bump remain.
//
// k-contiguous operand: tile ROWS x 32 floats, thread loads float4 (row = t/8 + 32*i, k = (t%8)*4)
template <int ROWS>
struct LoaderKC {
    const float* ptr[ROWS / 32];   // &base[row*ld + c4]
    unsigned valid, first, last;   // per pass i, bit i: row valid / conv t==0 / conv t==T-1
    int c4;
    const float* safe;             // a readable address for predicated-off lanes (loads stay unconditional: a
                                   // branch around every load costs more than the wasted 16 bytes)
    __device__ __forceinline__ void init(const float* base, int64_t ld, int row0, int nrows, int tid, int convT, const float* safe_) {
        c4 = (tid & 7) * 4;
        safe = safe_;
        const int r0 = tid >> 3;
        valid = first = last = 0;
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            const int r = row0 + r0 + 32 * i;
            const bool ok = r < nrows;
            ptr[i] = base + (int64_t)(ok ? r : row0) * ld + c4;
            if (ok) valid |= 1u << i;
            if (convT > 0) {
                const int t = r % convT;
                if (t == 0) first |= 1u << i;
                if (t == convT - 1) last |= 1u << i;
            }
        }
    }
    // full k-tile, no conv: rows past the edge read a clamped (valid) row; their products land in
    // output rows/cols that are never stored, so no predication is needed
    __device__ __forceinline__ void load_full(float4 (&reg)[ROWS / 32], int k0) const {
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) reg[i] = *reinterpret_cast<const float4*>(ptr[i] + k0);
    }
    __device__ __forceinline__ void load(float4 (&reg)[ROWS / 32], int k0, int K, int convE) const {
        const int k = k0 + c4;
        const bool kok = k < K;
        const bool seg0 = k < convE, seg2 = k >= 2 * convE;          // convE == 0 -> both false/true-but-unflagged
#pragma unroll
        for (int i = 0; i < ROWS / 32; ++i) {
            bool ok = kok && ((valid >> i) & 1u);
            if (((first >> i) & 1u) && seg0) ok = false;
            if (((last >> i) & 1u) && seg2) ok = false;
            const float4 v = *reinterpret_cast<const float4*>(ok ? ptr[i] + k0 : safe);
            reg[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
};
template <int ROWS>
__device__ __forceinline__ void store_kc(float* lds, const float4 (&reg)[ROWS / 32], int tid) {
    const int c4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i)
        *reinterpret_cast<float4*>(lds + (r0 + 32 * i) * KC_LD + c4) = reg[i];
}

// reduction-major operand: tile 32 (k) x COLS floats = 8*COLS/256 float4 per thread, element
// e = tid + 256*i -> k-row e / (COLS/4), column 4*(e % (COLS/4))   (COLS = 64, 128 or 320)
template <int COLS>
struct LoaderRM {
    static constexpr int TPR = COLS / 4;     // float4 per k-row
    static constexpr int NV = 8 * COLS / 256;
    const float* base;
    int64_t ld;
    uint32_t off[NV];                        // byte offset of element i inside a k-tile: (kr*ld + column) * 4, column clamped
    unsigned cvalid;                         // bit i: column in range
    int col0, convE;
    const float* safe;
    __device__ __forceinline__ void init(const float* base_, int64_t ld_, int col0_, int ncols, int tid, int convE_, const float* safe_) {
        base = base_; ld = ld_; col0 = col0_; convE = convE_; safe = safe_;
        cvalid = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + 256 * i, kr = e / TPR, c = col0 + (e - kr * TPR) * 4;
            const bool ok = c < ncols;
            if (ok) cvalid |= 1u << i;
            off[i] = (uint32_t)(kr * (int)ld + (ok ? c : col0)) * 4u;
        }
    }
    __device__ __forceinline__ void load_full(float4 (&reg)[NV], int k0, int) const {
        const char* kb = reinterpret_cast<const char*>(base + (int64_t)k0 * ld);      // uniform
#pragma unroll
        for (int i = 0; i < NV; ++i) reg[i] = *reinterpret_cast<const float4*>(kb + off[i]);
    }
    __device__ __forceinline__ void load(float4 (&reg)[NV], int k0, int K, int convT, int tid) const {
        const char* kb = reinterpret_cast<const char*>(base + (int64_t)k0 * ld);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int e = tid + 256 * i, kr = e / TPR, c = col0 + (e - kr * TPR) * 4;
            const int k = k0 + kr;
            bool ok = ((cvalid >> i) & 1u) && (k < K);
            if (convT > 0) {
                const int t = k % convT;
                if ((t == 0 && c < convE) || (t == convT - 1 && c >= 2 * convE)) ok = false;
            }
            const float4 v = *reinterpret_cast<const float4*>(ok ? kb + off[i] : reinterpret_cast<const char*>(safe));
            reg[i] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
};
template <int COLS>
__device__ __forceinline__ void store_rm(float* lds, const float4 (&reg)[8 * COLS / 256], int tid) {
    constexpr int TPR = COLS / 4;
#pragma unroll
    for (int i = 0; i < 8 * COLS / 256; ++i) {
        const int e = tid + 256 * i, kr = e / TPR, c4 = (e - kr * TPR) * 4;
        *reinterpret_cast<float4*>(lds + kr * RedMajor<COLS>::LD + c4) = reg[i];
    }
}

// epilogue math on a value whose aux operands are already in registers
template <int EPI>
__device__ __forceinline__ float epi_math(float v, float a, float a2) {
    if (EPI == EPI_RELU) return fmaxf(v, 0.f);
    if (EPI == EPI_GELU) return gelu_f(v);
    if (EPI == EPI_TANH) return tanhf(v);
    if (EPI == EPI_ADD_AUX) return v + a;
    if (EPI == EPI_MUL_GELU_GRAD) return v * gelu_grad_f(a);
    if (EPI == EPI_ADD_AUX_RELUMASK) return a2 > 0.f ? v + a : 0.f;
    if (EPI == EPI_RELUMASK) return a2 > 0.f ? v : 0.f;
    return v;
}
template <int EPI> struct EpiTraits {
    static constexpr bool AUX = EPI == EPI_ADD_AUX || EPI == EPI_MUL_GELU_GRAD || EPI == EPI_ADD_AUX_RELUMASK;
    static constexpr bool AUX2 = EPI == EPI_ADD_AUX_RELUMASK || EPI == EPI_RELUMASK;
};

__device__ __forceinline__ float epilogue(float v, int epi, int m, int n, const GemmParams& p) {
    const float a = (epi == EPI_ADD_AUX || epi == EPI_MUL_GELU_GRAD || epi == EPI_ADD_AUX_RELUMASK) ? p.aux[(int64_t)m * p.ldaux + n] : 0.f;
    const float a2 = (epi == EPI_ADD_AUX_RELUMASK || epi == EPI_RELUMASK) ? p.aux2[(int64_t)m * p.ldaux2 + n] : 0.f;
    switch (epi) {
        case EPI_RELU: return epi_math<EPI_RELU>(v, a, a2);
        case EPI_GELU:
            if (p.out2) p.out2[(int64_t)m * p.ldout2 + n] = v;
            return epi_math<EPI_GELU>(v, a, a2);
        case EPI_TANH: return epi_math<EPI_TANH>(v, a, a2);
        case EPI_ADD_AUX: return epi_math<EPI_ADD_AUX>(v, a, a2);
        case EPI_MUL_GELU_GRAD: return epi_math<EPI_MUL_GELU_GRAD>(v, a, a2);
        case EPI_ADD_AUX_RELUMASK: return epi_math<EPI_ADD_AUX_RELUMASK>(v, a, a2);
        case EPI_RELUMASK: return epi_math<EPI_RELUMASK>(v, a, a2);
    }
    return v;
}

// write one wave's 64 x (32*NJ) accumulator block.  C/D map of the 32x32 MFMA:
// col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).  Addressing is wave-uniform base
// (SGPR pair) + one 32-bit per-lane offset, so the 32*NJ stores need no per-lane 64-bit pointers.
// Aux operands of a 32x32 tile are loaded as one batch BEFORE its stores (the compiler cannot prove
// C does not alias aux, so interleaving them would serialise every element behind a round trip).
__device__ __forceinline__ float& at_off(float* base, uint32_t off_bytes) {           // SGPR base + u32 VGPR offset
    return *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + off_bytes);
}
__device__ __forceinline__ float ld_off(const float* base, uint32_t off_bytes) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + off_bytes);
}
template <int EPI, int WM, int NJ, bool CHECK>
__device__ __forceinline__ void store_block(const f32x16 (&acc)[WM][NJ], float* C, int mw, int nw, int l31, int h,
                                            bool use_bias, const GemmParams& p) {
    const int64_t ldc = p.ldc, ldaux = p.ldaux, ldaux2 = p.ldaux2, ldout2 = p.ldout2;
    const uint32_t vc = (uint32_t)(4 * h * (int)ldc + l31) * 4u;
    const uint32_t va = (uint32_t)(4 * h * (int)ldaux + l31) * 4u;
    const uint32_t va2 = (uint32_t)(4 * h * (int)ldaux2 + l31) * 4u;
    const uint32_t vo2 = (uint32_t)(4 * h * (int)ldout2 + l31) * 4u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nb = nw + j * 32;                       // uniform
        const int n = nb + l31;
        if (CHECK && n >= p.N) continue;
        const float bv = use_bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int mb = mw + i * 32;                   // uniform
            const int ml = mb + 4 * h;                    // first row of this lane
            float* cb = C + mb * ldc + nb;
            float a[16], a2[16];
            if (EpiTraits<EPI>::AUX) {
                const float* ab = p.aux + mb * ldaux + nb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    a[r] = (!CHECK || ml + ro < p.M) ? ld_off(ab + ro * ldaux, va) : 0.f;
                }
            }
            if (EpiTraits<EPI>::AUX2) {
                const float* ab = p.aux2 + mb * ldaux2 + nb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    a2[r] = (!CHECK || ml + ro < p.M) ? ld_off(ab + ro * ldaux2, va2) : 0.f;
                }
            }
            float* ob = (EPI == EPI_GELU && p.out2) ? p.out2 + mb * ldout2 + nb : nullptr;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (CHECK && ml + ro >= p.M) continue;
                const float v = acc[i][j][r] + bv;
                if (EPI == EPI_GELU && ob) at_off(ob + ro * ldout2, vo2) = v;
                at_off(cb + ro * ldc, vc) = epi_math<EPI>(v, EpiTraits<EPI>::AUX ? a[r] : 0.f, EpiTraits<EPI>::AUX2 ? a2[r] : 0.f);
            }
        }
    }
}

// the same block store with the epilogue chosen at run time (wave-uniform branches instead of one instantiation per
// epilogue): the bf16 kernel's k-loop is short on registers, and 16 inlined epilogue variants pushed it into scratch
template <int WM, int NJ, bool CHECK>
__device__ __forceinline__ void store_block_rt(const f32x16 (&acc)[WM][NJ], float* C, int mw, int nw, int l31, int h,
                                            bool use_bias, int epi, const GemmParams& p) {
    const int64_t ldc = p.ldc, ldaux = p.ldaux, ldaux2 = p.ldaux2, ldout2 = p.ldout2;
    const bool has_aux = epi == EPI_ADD_AUX || epi == EPI_MUL_GELU_GRAD || epi == EPI_ADD_AUX_RELUMASK;
    const bool has_aux2 = epi == EPI_ADD_AUX_RELUMASK || epi == EPI_RELUMASK;
    const bool has_out2 = epi == EPI_GELU && p.out2 != nullptr;
    const uint32_t vc = (uint32_t)(4 * h * (int)ldc + l31) * 4u;
    const uint32_t va = (uint32_t)(4 * h * (int)ldaux + l31) * 4u;
    const uint32_t va2 = (uint32_t)(4 * h * (int)ldaux2 + l31) * 4u;
    const uint32_t vo2 = (uint32_t)(4 * h * (int)ldout2 + l31) * 4u;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int nb = nw + j * 32;
        const int n = nb + l31;
        if (CHECK && n >= p.N) continue;
        const float bv = use_bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int mb = mw + i * 32;
            const int ml = mb + 4 * h;
            float* cb = C + mb * ldc + nb;
            float a[16], a2[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { a[r] = 0.f; a2[r] = 0.f; }
            if (has_aux) {
                const float* ab = p.aux + mb * ldaux + nb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    a[r] = (!CHECK || ml + ro < p.M) ? ld_off(ab + ro * ldaux, va) : 0.f;
                }
            }
            if (has_aux2) {
                const float* ab = p.aux2 + mb * ldaux2 + nb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    a2[r] = (!CHECK || ml + ro < p.M) ? ld_off(ab + ro * ldaux2, va2) : 0.f;
                }
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[i][j][r] + bv;
            if (has_out2) {
                float* ob = p.out2 + mb * ldout2 + nb;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    if (!CHECK || ml + ro < p.M) at_off(ob + ro * ldout2, vo2) = v[r];
                }
            }
            switch (epi) {
                case EPI_RELU:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_RELU>(v[r], a[r], a2[r]);
                    break;
                case EPI_GELU:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_GELU>(v[r], a[r], a2[r]);
                    break;
                case EPI_TANH:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_TANH>(v[r], a[r], a2[r]);
                    break;
                case EPI_ADD_AUX:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_ADD_AUX>(v[r], a[r], a2[r]);
                    break;
                case EPI_MUL_GELU_GRAD:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_MUL_GELU_GRAD>(v[r], a[r], a2[r]);
                    break;
                case EPI_ADD_AUX_RELUMASK:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_ADD_AUX_RELUMASK>(v[r], a[r], a2[r]);
                    break;
                case EPI_RELUMASK:
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = epi_math<EPI_RELUMASK>(v[r], a[r], a2[r]);
                    break;
                default: break;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ro = (r & 3) + 8 * (r >> 2);
                if (!CHECK || ml + ro < p.M) at_off(cb + ro * ldc, vc) = v[r];
            }
        }
    }
}

template <bool A_KC, bool B_KC, int WM, int NJ>
__global__ __launch_bounds__(256, (WM * NJ <= 2) ? 4 : 2) void gemm_f32_kernel(const GemmParams p) {
    constexpr int BM = 64 * WM, BN = 64 * NJ;
    constexpr int A_ELEMS = A_KC ? BM * KC_LD : BK * RedMajor<BM>::LD;
    constexpr int NVA = 8 * BM / 256, NVB = 8 * BN / 256;     // float4 per thread per k-tile
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // one buffer each (see header)
    float* Bs = smem + A_ELEMS;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: keeps tile bases in SGPRs
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD/L2, so give each XCD a
    // contiguous run of tiles (n fastest) -> the A row-panel is re-read from L2.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    // Split-K: the remap runs over the WHOLE grid, slice-major, so an XCD owns whole slices and the operand panels of
    // a slice are fetched into one L2 instead of eight (PMC, dW launches of the NRMS step: 605 -> 211 MB of HBM traffic
    // per launch at the same duration).  The tall tile keeps the per-slice remap: its 19 workgroups per slice then
    // stall on the same misses (454 -> 499 us) — it is latency-, not bandwidth-bound.
    int bid = blockIdx.x;
    int ks;
    if (WM > 2) {
        ks = bid / ntile;
        bid -= ks * ntile;
        const int q = ntile >> 3, r = ntile & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    } else {
        const int G = ntile * p.ksplit;
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
        ks = bid / ntile;
        bid -= ks * ntile;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // k range of this split
    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_per = (kt_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);

    f32x16 acc[WM][NJ];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[NVA];
    float4 rb[NVB];
    const int convA = (p.conv_on == 1) ? p.conv_T : 0;
    const int convB = (p.conv_on == 2) ? p.conv_T : 0;
    LoaderKC<BM> lkA; LoaderRM<BM> lrA; LoaderKC<BN> lkB; LoaderRM<BN> lrB;
    // conv views start one token row (conv_E floats) before the tensor: base + conv_E is the first readable float
    const float* safeA = p.A + (convA ? p.conv_E : 0);
    const float* safeB = p.B + (convB ? p.conv_E : 0);
    if (A_KC) lkA.init(p.A, p.lda, m0, p.M, tid, convA, safeA); else lrA.init(p.A, p.lda, m0, p.M, tid, 0, safeA);
    if (B_KC) lkB.init(p.B, p.ldb, n0, p.N, tid, 0, safeB);     else lrB.init(p.B, p.ldb, n0, p.N, tid, convB ? p.conv_E : 0, safeB);

    // Full k-tiles of a non-conv GEMM load unconditionally: rows/columns past the edge read a clamped
    // (valid) address and only ever reach output elements that are not stored.
    const bool plain = p.conv_on == 0;      // uniform
    auto gloadA = [&](int kt, bool fullk) {
        const int k0 = kt * BK;
        if (fullk && plain) { if (A_KC) lkA.load_full(ra, k0); else lrA.load_full(ra, k0, tid); }
        else { if (A_KC) lkA.load(ra, k0, p.K, convA ? p.conv_E : 0); else lrA.load(ra, k0, p.K, 0, tid); }
    };
    auto gloadB = [&](int kt, bool fullk) {
        const int k0 = kt * BK;
        if (fullk && plain) { if (B_KC) lkB.load_full(rb, k0); else lrB.load_full(rb, k0, tid); }
        else { if (B_KC) lkB.load(rb, k0, p.K, 0); else lrB.load(rb, k0, p.K, convB, tid); }
    };
    auto lstore = [&]() {
        if (A_KC) store_kc<BM>(As, ra, tid); else store_rm<BM>(As, ra, tid);
        if (B_KC) store_kc<BN>(Bs, rb, tid); else store_rm<BN>(Bs, rb, tid);
    };

    // LDS -> register fragments of one k-quad (8 k) and the MFMAs that consume them.  The fragments
    // are software-pipelined by hand (reads of quad q+1 are issued before the MFMAs of quad q): the
    // compiler does not overlap them on its own and a wave then idles for the LDS latency 2x per tile.
    struct Frags { float a[WM][4]; float b[NJ][4]; };
    auto fload = [&](Frags& f, int kq) {
        const int kb = kq * 8 + 4 * h;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            const int row = wr * (32 * WM) + i * 32 + l31;
            if (A_KC) {
                const float4 v = *reinterpret_cast<const float4*>(As + row * KC_LD + kb);
                f.a[i][0] = v.x; f.a[i][1] = v.y; f.a[i][2] = v.z; f.a[i][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.a[i][e] = As[(kb + e) * RedMajor<BM>::LD + row];
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = wc * (BN / 2) + j * 32 + l31;
            if (B_KC) {
                const float4 v = *reinterpret_cast<const float4*>(Bs + col * KC_LD + kb);
                f.b[j][0] = v.x; f.b[j][1] = v.y; f.b[j][2] = v.z; f.b[j][3] = v.w;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) f.b[j][e] = Bs[(kb + e) * RedMajor<BN>::LD + col];
            }
        }
    };
    auto fmma = [&](const Frags& f) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][e], f.b[j][e], acc[i][j], 0, 0, 0);
    };

    // bias gradient riding on the dW GEMM: the n-tile-0 workgroups also sum their A tile over k
    constexpr int NCS = (BM + 255) / 256;
    float cs[NCS];
#pragma unroll
    for (int j = 0; j < NCS; ++j) cs[j] = 0.f;
    const bool do_csum = !A_KC && p.csum != nullptr && tn == 0;
    auto csum_tile = [&]() {
        if (A_KC) return;
#pragma unroll
        for (int j = 0; j < NCS; ++j) {
            const int c = tid + 256 * j;
            if (c < BM) {
                float a = 0.f;
#pragma unroll 8
                for (int k = 0; k < BK; ++k) a += As[k * RedMajor<BM>::LD + c];
                cs[j] += a;
            }
        }
    };

    // tiles [kt_begin, kt_full) are complete (4 quads); at most one ragged tile follows
    const int kt_full = min(kt_end, p.K / BK);
    if (kt_begin < kt_end) {
        gloadA(kt_begin, false);
        gloadB(kt_begin, false);
        lstore();
    }
    __syncthreads();

    for (int kt = kt_begin; kt < kt_full; ++kt) {
        const bool more = kt + 1 < kt_end;
        const bool nfull = kt + 1 < kt_full;
        Frags f0, f1;
        fload(f0, 0);
        fload(f1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) gloadA(kt + 1, nfull);     // issued behind the first LDS reads: the MFMAs start at once
        fmma(f0);
        __builtin_amdgcn_sched_barrier(0);
        fload(f0, 2);
        if (more) gloadB(kt + 1, nfull);
        __builtin_amdgcn_sched_barrier(0);
        fmma(f1);
        __builtin_amdgcn_sched_barrier(0);
        fload(f1, 3);
        __builtin_amdgcn_sched_barrier(0);
        fmma(f0);
        fmma(f1);
        if (do_csum) csum_tile();
        __syncthreads();                     // everyone is done reading the single buffer
        if (more) lstore();
        __syncthreads();
    }
    if (kt_begin < kt_end && kt_full < kt_end) {   // ragged last tile (an empty split-K slice owns none): only the quads that hold k < K (the loader zero-fills)
        const int nkq = (p.K - kt_full * BK + 7) >> 3;
        for (int kq = 0; kq < nkq; ++kq) {
            Frags f;
            fload(f, kq);
            fmma(f);
        }
        if (do_csum) csum_tile();            // rows k >= K of the tile were zero-filled by the loader
    }
    if (do_csum) {
        float* dst = p.ksplit > 1 ? p.csum_part + (int64_t)ks * p.M : p.csum;
#pragma unroll
        for (int j = 0; j < NCS; ++j) {
            const int c = tid + 256 * j;
            if (c < BM && m0 + c < p.M) dst[m0 + c] = cs[j];
        }
    }

    // ---- epilogue (one switch per wave, straight-line stores inside)
    const bool partial = p.ksplit > 1;
    float* Cout = partial ? p.C + (int64_t)ks * p.split_stride : p.C;
    const int mw = m0 + wr * (32 * WM), nw = n0 + wc * (BN / 2);
    const bool full = (mw + 32 * WM <= p.M) && (nw + BN / 2 <= p.N);
    const bool ub = !partial && p.bias != nullptr;
    const int epi = partial ? EPI_NONE : p.epi;
#define NR_STORE(E)                                                                        \
    do {                                                                                   \
        if (full) store_block<E, WM, NJ, false>(acc, Cout, mw, nw, l31, h, ub, p);         \
        else      store_block<E, WM, NJ, true>(acc, Cout, mw, nw, l31, h, ub, p);          \
    } while (0)
    if (WM > 2) {                            // the tall tile only ever writes split-K slabs / plain dW
        NR_STORE(EPI_NONE);
    } else {
        switch (epi) {
            case EPI_NONE: NR_STORE(EPI_NONE); break;
            case EPI_RELU: NR_STORE(EPI_RELU); break;
            case EPI_GELU: NR_STORE(EPI_GELU); break;
            case EPI_TANH: NR_STORE(EPI_TANH); break;
            case EPI_ADD_AUX: NR_STORE(EPI_ADD_AUX); break;
            case EPI_MUL_GELU_GRAD: NR_STORE(EPI_MUL_GELU_GRAD); break;
            case EPI_ADD_AUX_RELUMASK: NR_STORE(EPI_ADD_AUX_RELUMASK); break;
            case EPI_RELUMASK: NR_STORE(EPI_RELUMASK); break;
        }
    }
#undef NR_STORE
}

// =================================================================================================
// bf16-input variant (v_mfma_f32_32x32x16_bf16, f32 accumulate): same operands in HBM (fp32), same
// epilogues and split-K; the tiles are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS.
// For the BERT-sized GEMMs whose parity bar is 1e-2.  Workgroup tile 128 x 128 x 32,
// 4 waves 2x2, wave tile 64 x 64.  Both operands live in LDS k-contiguous as [row][32] bf16
// (64-byte rows, 16-byte chunks XOR-swizzled by row: hb_swz);
// reduction-major operands are transposed while being stored: a thread loads two consecutive k-rows
// of the same 4 columns and writes 4 packed (k, k+1) pairs, lanes laid out so the 64 writes of one
// instruction hit 64 different banks.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int HB_LD = 32;            // bf16 per LDS row (64 B, unpadded: see hb_swz)

__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

typedef float f32x4 __attribute__((ext_vector_type(4)));   // staging registers as plain vector values (arrays of HIP's
                                                            // float4 struct ended up in scratch in this kernel)
__device__ __forceinline__ f32x4 ldg4(const void* p) { return *reinterpret_cast<const f32x4*>(p); }

// k-contiguous 128 x 32 tile: thread loads f32x4 (row = t/8 + 32*i, k = (t%8)*4), zero past the row / k range
struct LoaderKCH {
    const float* ptr[4];
    unsigned valid, first, last;   // per pass i, bit i: row in range / conv t == 0 / conv t == T-1
    int c4;
    const float* safe;
    __device__ __forceinline__ void init(const float* base, int64_t ld, int row0, int nrows, int tid, int convT, const float* safe_) {
        c4 = (tid & 7) * 4;
        safe = safe_;
        valid = first = last = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = row0 + (tid >> 3) + 32 * i;
            const bool ok = r < nrows;
            ptr[i] = base + (int64_t)(ok ? r : row0) * ld + c4;
            if (ok) valid |= 1u << i;
            if (convT > 0) {
                const int t = r % convT;
                if (t == 0) first |= 1u << i;
                if (t == convT - 1) last |= 1u << i;
            }
        }
    }
    // convE > 0: the row is an im2col view of 3 token rows; its first / last convE floats are zero at the title borders
    __device__ __forceinline__ void load(f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3, int k0, int K, int convE) const {
        const int k = k0 + c4;
        const bool kok = k < K;
        const unsigned dead = (k < convE ? first : 0u) | ((convE > 0 && k >= 2 * convE) ? last : 0u);
        const unsigned live = kok ? (valid & ~dead) : 0u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const bool o0 = live & 1u, o1 = live & 2u, o2 = live & 4u, o3 = live & 8u;
        const f32x4 v0 = ldg4(o0 ? ptr[0] + k0 : safe), v1 = ldg4(o1 ? ptr[1] + k0 : safe);
        const f32x4 v2 = ldg4(o2 ? ptr[2] + k0 : safe), v3 = ldg4(o3 ? ptr[3] + k0 : safe);
        r0 = o0 ? v0 : z; r1 = o1 ? v1 : z; r2 = o2 ? v2 : z; r3 = o3 ? v3 : z;
    }
};

// 16-byte chunk q (0..3) of row r sits at chunk q ^ hb_swz(r).  With unpadded 64-byte rows and this swizzle the
// ds_read_b128 fragment reads and the ds_write_b64 staging stores are conflict-free in the bank model of
// MI355X_MICROARCH.md (lane groups of 16, 64 banks) and measured so (SQ_LDS_BANK_CONFLICT = 0); the round-1 layout
// (80-byte rows, swizzle (r >> 3) & 2) cost 2x on both.
__device__ __forceinline__ int hb_swz(int row) { return (row >> 2) & 3; }

__device__ __forceinline__ void store_kc_bf16(uint32_t* lds, f32x4 r0, f32x4 r1, f32x4 r2, f32x4 r3, int tid) {
    const int q = (tid & 7) >> 1, half = (tid & 1) * 2, row0 = tid >> 3;
    const f32x4 r[4] = {r0, r1, r2, r3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = row0 + 32 * i;
        uint2 v;
        v.x = pack_bf16(r[i][0], r[i][1]);
        v.y = pack_bf16(r[i][2], r[i][3]);
        *reinterpret_cast<uint2*>(lds + row * (HB_LD / 2) + 4 * (q ^ hb_swz(row)) + half) = v;
    }
}

// reduction-major 32 (k) x 128 tile: wave w owns columns 32*w .. +31; lane = 8*c + a -> columns 32*w + 4*a .. +3 of the
// k pair 8*i + c (k = 16*i + 2*c, +1), i = 0, 1: every wave instruction reads eight full 128-byte lines
struct LoaderRMH {
    const float* base;
    int64_t ld;
    uint32_t off;
    bool cvalid;
    int c, col;
    const float* safe;
    __device__ __forceinline__ void init(const float* base_, int64_t ld_, int col0, int ncols, int tid, const float* safe_) {
        base = base_; ld = ld_; safe = safe_;
        const int w = tid >> 6, lane = tid & 63, a = lane & 7;
        c = lane >> 3;
        col = col0 + 32 * w + 4 * a;
        cvalid = col < ncols;
        off = (uint32_t)(2 * c * (int)ld + (cvalid ? col : col0)) * 4u;
    }
    // r0/r1: rows k, k+1 of the k pair c; r2/r3: of the k pair 8 + c.  convT > 0: the operand is the im2col view,
    // k = token row: columns < convE are zero at t == 0, columns >= 2*convE at t == T-1
    __device__ __forceinline__ bool conv_dead(int k, int convT, int convE) const {
        if (convT <= 0) return false;
        const int t = k % convT;
        return (t == 0 && col < convE) || (t == convT - 1 && col >= 2 * convE);
    }
    __device__ __forceinline__ void load(f32x4& r0, f32x4& r1, f32x4& r2, f32x4& r3, int k0, int K, int convT, int convE) const {
        const uint32_t rowb = (uint32_t)ld * 4u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const char* kb0 = reinterpret_cast<const char*>(base + (int64_t)k0 * ld);             // uniform
        const char* kb1 = reinterpret_cast<const char*>(base + (int64_t)(k0 + 16) * ld);
        const char* sf = reinterpret_cast<const char*>(safe);
        const int k = k0 + 2 * c;
        const bool o0 = cvalid && k < K && !conv_dead(k, convT, convE), o1 = cvalid && k + 1 < K && !conv_dead(k + 1, convT, convE);
        const bool o2 = cvalid && k + 16 < K && !conv_dead(k + 16, convT, convE), o3 = cvalid && k + 17 < K && !conv_dead(k + 17, convT, convE);
        const f32x4 v0 = ldg4(o0 ? kb0 + off : sf), v1 = ldg4(o1 ? kb0 + off + rowb : sf);
        const f32x4 v2 = ldg4(o2 ? kb1 + off : sf), v3 = ldg4(o3 ? kb1 + off + rowb : sf);
        r0 = o0 ? v0 : z; r1 = o1 ? v1 : z; r2 = o2 ? v2 : z; r3 = o3 ? v3 : z;
    }
};
__device__ __forceinline__ void store_rm_bf16(uint32_t* lds, f32x4 r0, f32x4 r1, f32x4 r2, f32x4 r3, int tid) {
    const int w = tid >> 6, lane = tid & 63, a = lane & 7, c = lane >> 3;
    const int row = 32 * w + 4 * a;                       // rows row .. row+3 (row a multiple of 4) share the swizzle
    const int sw = hb_swz(row);
    uint32_t* d0 = lds + row * (HB_LD / 2) + 4 * ((c >> 2) ^ sw) + (c & 3);
    uint32_t* d1 = lds + row * (HB_LD / 2) + 4 * ((2 + (c >> 2)) ^ sw) + (c & 3);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        d0[j * (HB_LD / 2)] = pack_bf16(r0[j], r1[j]);
        d1[j * (HB_LD / 2)] = pack_bf16(r2[j], r3[j]);
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const GemmParams p) {
    constexpr int BM = 128, BN = 128, WM = 2, NJ = 2;
    constexpr int STAGE = (BM + BN) * HB_LD / 2;          // dwords per pipeline stage
    __shared__ __attribute__((aligned(16))) uint32_t smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, h = lane >> 5;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {                                       // slice-major over the whole grid (see gemm_f32_kernel)
        const int G = ntile * p.ksplit;
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int ks = bid / ntile;
    bid -= ks * ntile;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_per = (kt_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);

    f32x16 acc[WM][NJ];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 a0, a1, a2, a3, b0, b1, b2, b3;                 // the next k-tile on its way HBM -> LDS
    LoaderKCH lkA, lkB; LoaderRMH lrA, lrB;
    // conv views (gemm.hip header): they start one token row (conv_E floats) before the tensor
    const int convA = (A_KC && p.conv_on == 1) ? p.conv_T : 0;
    const int convB = (!B_KC && p.conv_on == 2) ? p.conv_T : 0;
    const float* safeA = p.A + (convA ? p.conv_E : 0);
    const float* safeB = p.B + (convB ? p.conv_E : 0);
    if (A_KC) lkA.init(p.A, p.lda, m0, p.M, tid, convA, safeA); else lrA.init(p.A, p.lda, m0, p.M, tid, safeA);
    if (B_KC) lkB.init(p.B, p.ldb, n0, p.N, tid, 0, safeB);     else lrB.init(p.B, p.ldb, n0, p.N, tid, safeB);

    // bias gradient riding on dW (A reduction-major): exact fp32 column sums of the A tile taken from the
    // staging registers before they are rounded; the 16 lanes that share a column meet in the epilogue
    const bool do_csum = !A_KC && p.csum != nullptr && tn == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};                      // columns 32*w + 4*a .. +3, partial over this lane's k pairs

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        if (A_KC) lkA.load(a0, a1, a2, a3, k0, p.K, convA ? p.conv_E : 0); else lrA.load(a0, a1, a2, a3, k0, p.K, 0, 0);
        if (B_KC) lkB.load(b0, b1, b2, b3, k0, p.K, 0);                    else lrB.load(b0, b1, b2, b3, k0, p.K, convB, p.conv_E);
    };
    auto lstore = [&](int buf) {
        uint32_t* As = smem + buf * STAGE;
        uint32_t* Bs = As + BM * HB_LD / 2;
        if (A_KC) store_kc_bf16(As, a0, a1, a2, a3, tid);
        else {
            store_rm_bf16(As, a0, a1, a2, a3, tid);
            if (do_csum) cs += (a0 + a1) + (a2 + a3);
        }
        if (B_KC) store_kc_bf16(Bs, b0, b1, b2, b3, tid); else store_rm_bf16(Bs, b0, b1, b2, b3, tid);
    };

    // two LDS stages, one barrier per k-tile: while tile kt is multiplied out of stage `cur`, tile kt+1 travels
    // HBM -> registers -> stage cur^1 (last read before the previous barrier)
    if (kt_begin < kt_end) {
        gload(kt_begin);
        lstore(0);
    }
    __syncthreads();
    const int frag_sw = hb_swz(l31);                      // tile bases are multiples of 32 rows
    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) gload(kt + 1);
        const uint32_t* As = smem + cur * STAGE;
        const uint32_t* Bs = As + BM * HB_LD / 2;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int q4 = 4 * ((2 * s + h) ^ frag_sw);
            bf16x8 fa[WM], fb[NJ];
#pragma unroll
            for (int i = 0; i < WM; ++i)
                fa[i] = *reinterpret_cast<const bf16x8*>(As + (wr * 64 + i * 32 + l31) * (HB_LD / 2) + q4);
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                fb[j] = *reinterpret_cast<const bf16x8*>(Bs + (wc * 64 + j * 32 + l31) * (HB_LD / 2) + q4);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    if (do_csum) {
        float* dst = p.ksplit > 1 ? p.csum_part + (int64_t)ks * p.M : p.csum;
        const int a = lane & 7, c = lane >> 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float v = cs[j];
            v += __shfl_xor(v, 8);
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            const int col = m0 + 32 * wid + 4 * a + j;
            if (c == 0 && col < p.M) dst[col] = v;
        }
    }

    const bool partial = p.ksplit > 1;
    float* Cout = partial ? p.C + (int64_t)ks * p.split_stride : p.C;
    const int mw = m0 + wr * 64, nw = n0 + wc * 64;
    const bool full = (mw + 64 <= p.M) && (nw + 64 <= p.N);
    const bool ub = !partial && p.bias != nullptr;
    const int epi = partial ? EPI_NONE : p.epi;
    if (full) store_block_rt<WM, NJ, false>(acc, Cout, mw, nw, l31, h, ub, epi, p);
    else      store_block_rt<WM, NJ, true>(acc, Cout, mw, nw, l31, h, ub, epi, p);
}

template <bool A_KC, bool B_KC>
static int launch_bf16(const GemmParams& p, hipStream_t s) {
    const int tiles = cdiv(p.M, 128) * cdiv(p.N, 128);
    hipLaunchKernelGGL((gemm_bf16_kernel<A_KC, B_KC>), dim3(tiles * p.ksplit), dim3(256), 0, s, p);
    return check_launch("gemm_bf16");
}

// =================================================================================================
// fp32 products on the bf16 matrix pipe ("x3"): every fp32 operand is split on its way into LDS into three bf16 pieces
//   a = a0 + a1 + a2,  a0 = bf16(a), a1 = bf16(a - a0), a2 = bf16(a - a0 - a1)        (both subtractions exact in fp32)
// which together carry all 24 significand bits, and a*b is accumulated in fp32 from the six piece products of weight
// >= 2^-16:  a0*b2 + a2*b0 + a1*b1 + a0*b1 + a1*b0 + a0*b0  (smallest first).  Each bf16 x bf16 product is exact in
// fp32; the three dropped products (a1*b2, a2*b1, a2*b2) are below 2^-23 |a*b|, the size of fp32's own rounding of
// the product.  v_mfma_f32_32x32x16_bf16 retires 16x the multiply-adds per cycle of v_mfma_f32_32x32x2_f32, so six
// of them per k-step still run 2.7x faster than the fp32 instruction while the result keeps fp32 accuracy (the parity
// tests hold this path to the same 1e-5 as the fp32 MFMA kernel).  Same operands in HBM (fp32), layouts, conv views,
// epilogues, split-K and fused column sums as gemm_f32_kernel; loaders / LDS layout of gemm_bf16_kernel above, with
// three planes per operand.  Workgroup tile (64*WM) x (64*NJ) x 32, 4 waves 2x2, ONE LDS stage with the next k-tile
// prefetched into registers: 46-61 KB per workgroup leaves room for 2-3 workgroups per CU, whose waves keep the matrix
// pipe busy while this one splits and stores its next tile.
// LDS image of the x3 kernel: rows of exactly 32 bf16 (64 B, no padding), 16-byte chunk q of row r stored at chunk
// q ^ ((r >> 2) & 3).  By the bank model of MI355X_MICROARCH.md (LDS table) that makes the ds_read_b128 fragment reads
// and the ds_write_b64 staging stores conflict-free (padded 80-byte rows with the swizzle (r >> 3) & 2, the first layout, cost 2x on both:
// SQ_LDS_BANK_CONFLICT was half of SQ_LDS_IDX_ACTIVE).
constexpr int X3_LD = 32;
__device__ __forceinline__ int x3_swz(int row) { return (row >> 2) & 3; }

template <int ROWS>
struct LoaderKCX {
    static constexpr int NP = ROWS / 32;
    const float* base;             // first row of the tile (uniform)
    uint32_t off[NP];              // byte offset of this thread's float4 in pass i (row clamped into the tensor)
    unsigned valid, first, last;
    int c4;
    const float* safe;
    __device__ __forceinline__ void init(const float* base_, int64_t ld, int row0, int nrows, int tid, int convT, const float* safe_) {
        c4 = (tid & 7) * 4;
        safe = safe_;
        base = base_ + (int64_t)row0 * ld;
        valid = first = last = 0;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = row0 + (tid >> 3) + 32 * i;
            const bool ok = r < nrows;
            off[i] = (uint32_t)((ok ? r - row0 : 0) * (int)ld + c4) * 4u;
            if (ok) valid |= 1u << i;
            if (convT > 0) {
                const int t = r % convT;
                if (t == 0) first |= 1u << i;
                if (t == convT - 1) last |= 1u << i;
            }
        }
    }
    // a full k-tile of a plain (non-conv) operand: rows past the edge read a clamped (valid) row, whose products only
    // reach output rows / columns that are never stored -> no predication, wave-uniform base + 32-bit lane offset
    __device__ __forceinline__ void load_full(f32x4 (&r)[NP], int k0) const {
        const char* kb = reinterpret_cast<const char*>(base + k0);                  // uniform
#pragma unroll
        for (int i = 0; i < NP; ++i) r[i] = ldg4(kb + off[i]);
    }
    __device__ __forceinline__ void load(f32x4 (&r)[NP], int k0, int K, int convE) const {
        const int k = k0 + c4;
        const bool kok = k < K;
        const unsigned dead = (k < convE ? first : 0u) | ((convE > 0 && k >= 2 * convE) ? last : 0u);
        const unsigned live = kok ? (valid & ~dead) : 0u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const char* kb = reinterpret_cast<const char*>(base + k0);                  // uniform
        const char* sf = reinterpret_cast<const char*>(safe);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const bool o = (live >> i) & 1u;
            const f32x4 v = ldg4(o ? kb + off[i] : sf);
            r[i] = o ? v : z;
        }
    }
};

// two floats -> their three packed bf16 pieces
#ifndef NR_X3_EXP
#define NR_X3_EXP 0
#endif
#ifndef NR_X3_WGS
#define NR_X3_WGS 2
#endif
#ifndef NR_X3_IGLP
#define NR_X3_IGLP 0                                       // -1: no iglp_opt in gemm_x3_kernel's k-loop (A/B builds)
#endif
#if NR_X3_EXP == 4
// diagnostic build only (never shipped: the default build has none of this): per-workgroup cycle stamps of the x3 kernel
__device__ unsigned long long x3_stamps[8192 * 32];
#define X3_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) x3_stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define X3_STAMP_RT(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) x3_stamps[blockIdx.x * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define X3_STAMP(i) do {} while (0)
#define X3_STAMP_RT(i) do {} while (0)
#endif
__device__ __forceinline__ void split3(float x, float y, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
#if NR_X3_EXP == 1
    p0 = __builtin_amdgcn_perm(__float_as_uint(y), __float_as_uint(x), 0x07060302u);
    p1 = p0; p2 = p0;
    return;
#endif
    p0 = pack_bf16(x, y);
    float rx = x - __uint_as_float(p0 << 16), ry = y - __uint_as_float(p0 & 0xffff0000u);
    p1 = pack_bf16(rx, ry);
    rx -= __uint_as_float(p1 << 16);
    ry -= __uint_as_float(p1 & 0xffff0000u);
    p2 = pack_bf16(rx, ry);
}

template <int ROWS>
__device__ __forceinline__ void store_kc_x3(uint32_t* lds, const f32x4 (&r)[ROWS / 32], int tid) {
    constexpr int PL = ROWS * X3_LD / 2;                  // dwords per plane
    const int q = (tid & 7) >> 1, half = (tid & 1) * 2, row0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int row = row0 + 32 * i;
        uint2 v0, v1, v2;
        split3(r[i][0], r[i][1], v0.x, v1.x, v2.x);
        split3(r[i][2], r[i][3], v0.y, v1.y, v2.y);
        uint32_t* d = lds + row * (X3_LD / 2) + 4 * (q ^ x3_swz(row)) + half;
        *reinterpret_cast<uint2*>(d) = v0;
        *reinterpret_cast<uint2*>(d + PL) = v1;
        *reinterpret_cast<uint2*>(d + 2 * PL) = v2;
        // keep each pass's stores behind its own split: left alone the compiler sinks all stores to the end of the phase,
        // where the four waves then queue on the LDS write path (~85 B/clk) after having left it idle during the split
        __builtin_amdgcn_sched_barrier(0);
    }
}

// reduction-major 32 (k) x COLS tile, COLS = 64 | 128 | 256: one wave instruction covers 32 columns x 8 k pairs (lane =
// 8*c + a: columns 4*a .. +3 of the group, k pair c of the half).  The COLS/16 (column group, k half) units are dealt
// round-robin to the waves: unit u = w + 4*i -> column group u % (COLS/32), k half u / (COLS/32).
template <int COLS>
struct LoaderRMX {
    static constexpr int NP = COLS / 64;                  // units (k pairs) per thread
    static constexpr int NG = COLS / 32;
    const float* base;
    int64_t ld;
    int w, a, c, col0, ncols;
    const float* safe;
    __device__ __forceinline__ void init(const float* base_, int64_t ld_, int col0_, int ncols_, int tid, const float* safe_) {
        base = base_; ld = ld_; safe = safe_; col0 = col0_; ncols = ncols_;
        w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
        a = lane & 7;
        c = lane >> 3;
    }
    __device__ __forceinline__ int lrow(int i) const { return 32 * ((w + 4 * i) % NG) + 4 * a; }      // tile-local first column
    __device__ __forceinline__ int kp(int i) const { return 8 * ((w + 4 * i) / NG) + c; }             // k pair inside the k-tile
    __device__ __forceinline__ bool conv_dead(int k, int col, int convT, int convE) const {
        if (convT <= 0) return false;
        const int t = k % convT;
        return (t == 0 && col < convE) || (t == convT - 1 && col >= 2 * convE);
    }
    __device__ __forceinline__ void load_full(f32x4 (&r)[2 * NP], int k0) const {
        const uint32_t rowb = (uint32_t)ld * 4u;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int khalf = (w + 4 * i) / NG;                                                      // uniform
            const char* kb = reinterpret_cast<const char*>(base + (int64_t)(k0 + 16 * khalf) * ld);
            const int col = col0 + lrow(i);
            const uint32_t off = (uint32_t)(2 * c * (int)ld + (col < ncols ? col : col0)) * 4u;
            r[2 * i] = ldg4(kb + off);
            r[2 * i + 1] = ldg4(kb + off + rowb);
        }
    }
    __device__ __forceinline__ void load(f32x4 (&r)[2 * NP], int k0, int K, int convT, int convE) const {
        const uint32_t rowb = (uint32_t)ld * 4u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const char* sf = reinterpret_cast<const char*>(safe);
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int khalf = (w + 4 * i) / NG;                                                      // uniform
            const char* kb = reinterpret_cast<const char*>(base + (int64_t)(k0 + 16 * khalf) * ld);
            const int col = col0 + lrow(i);
            const bool cvalid = col < ncols;
            const uint32_t off = (uint32_t)(2 * c * (int)ld + (cvalid ? col : col0)) * 4u;
            const int k = k0 + 2 * kp(i);
            const bool o0 = cvalid && k < K && !conv_dead(k, col, convT, convE), o1 = cvalid && k + 1 < K && !conv_dead(k + 1, col, convT, convE);
            const f32x4 v0 = ldg4(o0 ? kb + off : sf), v1 = ldg4(o1 ? kb + off + rowb : sf);
            r[2 * i] = o0 ? v0 : z;
            r[2 * i + 1] = o1 ? v1 : z;
        }
    }
};
template <int COLS>
__device__ __forceinline__ void store_rm_x3(uint32_t* lds, const f32x4 (&r)[COLS / 32], const LoaderRMX<COLS>& L) {
    constexpr int PL = COLS * X3_LD / 2;
#pragma unroll
    for (int i = 0; i < COLS / 64; ++i) {
        const int lrow = L.lrow(i), kp = L.kp(i);
        const int sw = x3_swz(lrow);                      // rows lrow .. lrow+3 (lrow a multiple of 4) share the swizzle
        uint32_t* d = lds + lrow * (X3_LD / 2) + 4 * ((kp >> 2) ^ sw) + (kp & 3);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            uint32_t p0, p1, p2;
            split3(r[2 * i][j], r[2 * i + 1][j], p0, p1, p2);
            d[j * (X3_LD / 2)] = p0;
            d[j * (X3_LD / 2) + PL] = p1;
            d[j * (X3_LD / 2) + 2 * PL] = p2;
        }
    }
}

// WR x WC waves (WR * WC = 4), each owning a 64 x 64 block (2 x 2 MFMA tiles): per k-step of 16 a wave reads 12 LDS
// fragments (2 row blocks + 2 column blocks, 3 pieces each) for its 24 MFMAs.  LDS bandwidth is what this kernel is
// bounded by next to the matrix pipe (a 64 x 32 wave block needs 9 fragments per 12 MFMAs and ran LDS-bound, 95-115
// TFLOP/s), hence the square wave block and the tile shapes 256 x 64 / 128 x 128 / 64 x 256 around it.
template <bool A_KC, bool B_KC, int WR, int WC>
__global__ __launch_bounds__(256, NR_X3_WGS) void gemm_x3_kernel(const GemmParams p) {
    constexpr int BM = 64 * WR, BN = 64 * WC;
    constexpr int PA = BM * X3_LD / 2, PB = BN * X3_LD / 2;           // dwords per plane
    constexpr int NA = BM / 32, NB = BN / 32;                          // staging f32x4 per thread
    extern __shared__ __attribute__((aligned(16))) uint32_t smem_x3[]; // A planes 0..2, B planes 0..2

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid / WC, wc = wid % WC;
    const int l31 = lane & 31, h = lane >> 5;

    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {                                       // slice-major over the whole grid (see gemm_f32_kernel)
        const int G = ntile * p.ksplit;
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int ks = bid / ntile;
    bid -= ks * ntile;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_per = (kt_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 ra[NA], rb[NB];                                  // the next k-tile on its way HBM -> LDS
    LoaderKCX<BM> lkA; LoaderKCX<BN> lkB; LoaderRMX<BM> lrA; LoaderRMX<BN> lrB;
    const int convA = (A_KC && p.conv_on == 1) ? p.conv_T : 0;
    const int convB = (!B_KC && p.conv_on == 2) ? p.conv_T : 0;
    const float* safeA = p.A + (convA ? p.conv_E : 0);
    const float* safeB = p.B + (convB ? p.conv_E : 0);
    if (A_KC) lkA.init(p.A, p.lda, m0, p.M, tid, convA, safeA); else lrA.init(p.A, p.lda, m0, p.M, tid, safeA);
    if (B_KC) lkB.init(p.B, p.ldb, n0, p.N, tid, 0, safeB);     else lrB.init(p.B, p.ldb, n0, p.N, tid, safeB);

    // bias gradient riding on dW (A reduction-major): exact fp32 column sums of the A tile from the staging registers,
    // one partial per unit of the loader
    const bool do_csum = !A_KC && p.csum != nullptr && tn == 0;
    f32x4 cs[A_KC ? 1 : BM / 64];
#pragma unroll
    for (int i = 0; i < (A_KC ? 1 : BM / 64); ++i) cs[i] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool plain = p.conv_on == 0;
    const int kt_fullk = p.K / BK;                         // k-tiles below this index are complete
    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        if (plain && kt < kt_fullk) {                      // uniform: no predication, clamped rows (see load_full)
            if (A_KC) lkA.load_full(ra, k0); else lrA.load_full(ra, k0);
            if (B_KC) lkB.load_full(rb, k0); else lrB.load_full(rb, k0);
        } else {
            if (A_KC) lkA.load(ra, k0, p.K, convA ? p.conv_E : 0); else lrA.load(ra, k0, p.K, 0, 0);
            if (B_KC) lkB.load(rb, k0, p.K, 0);                    else lrB.load(rb, k0, p.K, convB, p.conv_E);
        }
    };
    auto lstore = [&]() {
        uint32_t* As = smem_x3;
        uint32_t* Bs = As + 3 * PA;
        if constexpr (A_KC) store_kc_x3<BM>(As, ra, tid);
        else {
            store_rm_x3<BM>(As, ra, lrA);
            if (do_csum) {
#pragma unroll
                for (int i = 0; i < BM / 64; ++i) cs[i] += ra[2 * i] + ra[2 * i + 1];
            }
        }
        if constexpr (B_KC) store_kc_x3<BN>(Bs, rb, tid); else store_rm_x3<BN>(Bs, rb, lrB);
    };

    struct Frags { bf16x8 a[2][3]; bf16x8 b[2][3]; };
    const int frag_sw = x3_swz(l31);                      // tile bases are multiples of 32 rows
    auto fload = [&](Frags& f, int s) {
        const uint32_t* As = smem_x3;
        const uint32_t* Bs = As + 3 * PA;
        const int q4 = 4 * ((2 * s + h) ^ frag_sw);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                f.a[i][t] = *reinterpret_cast<const bf16x8*>(As + t * PA + (wr * 64 + i * 32 + l31) * (X3_LD / 2) + q4);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 3; ++t)
                f.b[j][t] = *reinterpret_cast<const bf16x8*>(Bs + t * PB + (wc * 64 + j * 32 + l31) * (X3_LD / 2) + q4);
    };
    auto fmma = [&](const Frags& f) {
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // smallest products first
#if NR_X3_EXP == 5
        // diagnostic: every fragment read stays (kept alive below), one of the six MFMAs per product
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 3; ++t) { asm volatile("" :: "v"(f.a[i][t])); asm volatile("" :: "v"(f.b[i][t])); }
#endif
#pragma unroll
        for (int t = ((NR_X3_EXP == 2 || NR_X3_EXP == 5) ? 5 : 0); t < 6; ++t)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][TA[t]], f.b[j][TB[t]], acc[i][j], 0, 0, 0);
    };

    // ONE LDS stage with the next k-tile prefetched into registers (the single-buffer loop of gemm_f32_kernel): 49-61 KB
    // of LDS and <= 168 VGPRs leave room for 2-3 workgroups per CU.  With K = 300 (ten k-tiles) a workgroup is mostly
    // prologue and epilogue, which only other resident workgroups can cover: a two-stage, one-workgroup-per-CU pipeline
    // measured slower on every NRMS shape (and no faster at 4096^3, where the k-loop waits on L2 either way).
    const int nkt = kt_end - kt_begin;
    X3_STAMP(0);
    X3_STAMP_RT(1);
    if (nkt > 0) {
        gload(kt_begin);
        lstore();
    }
    __syncthreads();
    X3_STAMP(2);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
#if NR_X3_WGS >= 3
        Frags f0;                                          // 168 VGPRs: one fragment set at a time
        fload(f0, 0);
        if (more) gload(kt + 1);
        fmma(f0);
        fload(f0, 1);
        fmma(f0);
#else
        Frags f0, f1;
        fload(f0, 0);
#if NR_X3_EXP == 8
        if (more && kt == kt_begin) gload(kt + 1);         // diagnostic: no global loads after the second k-tile
#else
        if (more) gload(kt + 1);
#endif
#if NR_X3_EXP == 6
        f1 = f0;                                           // diagnostic: half of the fragment reads, all MFMAs
#else
        fload(f1, 1);
#endif
        fmma(f0);
        fmma(f1);
#if NR_X3_IGLP >= 0
        // the compiler's own DS / MFMA interleaving pattern for this block; together with -mllvm -enable-post-misched=0
        // (Makefile) 2.3 % on the twelve NRMS products, three alternating rounds (1 082 -> 1 057 us summed; either alone 0.7 / 1.2 %)
        __builtin_amdgcn_iglp_opt(NR_X3_IGLP);
#endif
#endif
        if (kt - kt_begin < 6) X3_STAMP(3 + 4 * (kt - kt_begin));
        __syncthreads();                                  // everyone is done reading the single stage
        if (kt - kt_begin < 6) X3_STAMP(4 + 4 * (kt - kt_begin));
#if NR_X3_EXP == 7
        // diagnostic: no split and no LDS stores after the first k-tile (the loaded registers are kept alive)
        if (more) {
#pragma unroll
            for (int i = 0; i < NA; ++i) asm volatile("" :: "v"(ra[i]));
#pragma unroll
            for (int i = 0; i < NB; ++i) asm volatile("" :: "v"(rb[i]));
        }
#else
        if (more) lstore();
#endif
        if (kt - kt_begin < 6) X3_STAMP(5 + 4 * (kt - kt_begin));
        __syncthreads();
        if (kt - kt_begin < 6) X3_STAMP(6 + 4 * (kt - kt_begin));
    }
    X3_STAMP(28);
    if (do_csum) {
        // lanes that share 4 columns (same a, all c) meet by shuffles, the units of one column group (two k halves, on
        // one or two waves) through LDS: two addends per slot on top of 0 -> order-independent
        float* dst = p.ksplit > 1 ? p.csum_part + (int64_t)ks * p.M : p.csum;
        float* red = reinterpret_cast<float*>(smem_x3);
        for (int x = tid; x < BM; x += 256) red[x] = 0.f;
        __syncthreads();
        if constexpr (!A_KC) {
#pragma unroll
            for (int i = 0; i < BM / 64; ++i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v = cs[i][j];
                    v += __shfl_xor(v, 8);
                    v += __shfl_xor(v, 16);
                    v += __shfl_xor(v, 32);
                    if ((lane >> 3) == 0) atomicAdd(&red[lrA.lrow(i) + j], v);
                }
            }
        }
        __syncthreads();
        for (int x = tid; x < BM; x += 256)
            if (m0 + x < p.M) dst[m0 + x] = red[x];
    }

    const bool partial = p.ksplit > 1;
    float* Cout = partial ? p.C + (int64_t)ks * p.split_stride : p.C;
    const int mw = m0 + wr * 64, nw = n0 + wc * 64;
    const bool full = (mw + 64 <= p.M) && (nw + 64 <= p.N);
    const bool ub = !partial && p.bias != nullptr;
    const int epi = partial ? EPI_NONE : p.epi;
#if NR_X3_EXP == 3
    if (acc[0][0][0] != 12345.678f) return;               // timing experiment: the k-loop without its epilogue
#endif
    if (full) store_block_rt<2, 2, false>(acc, Cout, mw, nw, l31, h, ub, epi, p);
    else      store_block_rt<2, 2, true>(acc, Cout, mw, nw, l31, h, ub, epi, p);
    X3_STAMP(29);
    X3_STAMP_RT(30);
}
#if NR_X3_EXP == 4
}  // namespace nr
extern "C" int nr_debug_x3_stamps(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(nr::x3_stamps), sizeof(unsigned long long) * n);
}
namespace nr {
#endif

template <bool A_KC, bool B_KC, int WR, int WC>
static int launch_x3(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 64 * WR, BN = 64 * WC;
    constexpr size_t lds = 3 * (BM + BN) * X3_LD * 2;
    static bool once = [] {                               // > 64 KB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_x3_kernel<A_KC, B_KC, WR, WC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        return true;
    }();
    (void)once;
    const int tiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_x3_kernel<A_KC, B_KC, WR, WC>), dim3(tiles * p.ksplit), dim3(256), lds, s, p);
    return check_launch("gemm_x3");
}
// shape: 0 = 256 x 64, 1 = 128 x 128, 2 = 64 x 256
template <bool A_KC, bool B_KC>
static int launch_x3_tile(const GemmParams& p, int shape, hipStream_t s) {
    if (shape == 0) return launch_x3<A_KC, B_KC, 4, 1>(p, s);
    if (shape == 2) return launch_x3<A_KC, B_KC, 1, 4>(p, s);
    return launch_x3<A_KC, B_KC, 2, 2>(p, s);
}
static int x3_tile_shape(int M, int N) {
    // the shape that pads least (ties: the square one)
    int64_t best = -1; int shape = 1;
    const int bm[3] = {256, 128, 64}, bn[3] = {64, 128, 256}, order[3] = {1, 0, 2};
    for (int o = 0; o < 3; ++o) {
        const int t = order[o];
        const int64_t area = (int64_t)cdiv(M, bm[t]) * bm[t] * cdiv(N, bn[t]) * bn[t];
        if (best < 0 || area < best) { best = area; shape = t; }
    }
    return shape;
}

// ---- the same arithmetic for products with few tiles (the user tower: 1 600 rows): ONE 64 x 64 block per workgroup,
// shared by its four waves, which split every 64-deep k-tile between them (wave w takes the k-step 16*w .. 16*w+15) and
// meet through LDS at the end.  25 x 5 tiles of a [1600, 300] product are 125 workgroups whose per-k-tile chain is a
// quarter of the fp32 instruction kernel's (24 bf16 MFMAs = 768 cycles per wave and 64 k, against 32 x 64 cycles), and
// three of them share a CU.  LDS image: rows of 64 bf16 (128 B), 16-byte chunk c of row r at c ^ ((r >> 1) & 7)
// (conflict-free b128 fragment reads and b64 staging stores by the same bank model).
constexpr int XS_BK = 64;
constexpr int XS_LDW = 32;                                   // dwords per LDS row
__device__ __forceinline__ int xs_swz(int row) { return (row >> 1) & 7; }

constexpr int XS_NT = 512;                                   // threads per workgroup: 8 waves
constexpr int XS_NW = XS_NT / 64;
struct LoaderKCS {                      // 64 rows x 64 k floats: thread loads f32x4 (row = t/16 + 32*i, k = (t%16)*4), i = 0, 1
    const float* base;
    uint32_t off[2];
    unsigned valid, first, last;
    int c4;
    const float* safe;
    __device__ __forceinline__ void init(const float* base_, int64_t ld, int row0, int nrows, int tid, int convT, const float* safe_) {
        c4 = (tid & 15) * 4;
        safe = safe_;
        base = base_ + (int64_t)row0 * ld;
        valid = first = last = 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = row0 + (tid >> 4) + 32 * i;
            const bool ok = r < nrows;
            off[i] = (uint32_t)((ok ? r - row0 : 0) * (int)ld + c4) * 4u;
            if (ok) valid |= 1u << i;
            if (convT > 0) {
                const int t = r % convT;
                if (t == 0) first |= 1u << i;
                if (t == convT - 1) last |= 1u << i;
            }
        }
    }
    __device__ __forceinline__ void load(f32x4 (&r)[2], int k0, int K, int convE) const {
        const int k = k0 + c4;
        const bool kok = k < K;
        const unsigned dead = (k < convE ? first : 0u) | ((convE > 0 && k >= 2 * convE) ? last : 0u);
        const unsigned live = kok ? (valid & ~dead) : 0u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const char* kb = reinterpret_cast<const char*>(base + k0);
        const char* sf = reinterpret_cast<const char*>(safe);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool o = (live >> i) & 1u;
            const f32x4 v = ldg4(o ? kb + off[i] : sf);
            r[i] = o ? v : z;
        }
    }
};
__device__ __forceinline__ void store_kc_xs(uint32_t* lds, const f32x4 (&r)[2], int tid) {
    constexpr int PL = 64 * XS_LDW;
    const int chunk = (tid & 15) >> 1, half = (tid & 1) * 2, row0 = tid >> 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = row0 + 32 * i;
        uint2 v0, v1, v2;
        split3(r[i][0], r[i][1], v0.x, v1.x, v2.x);
        split3(r[i][2], r[i][3], v0.y, v1.y, v2.y);
        uint32_t* d = lds + row * XS_LDW + 4 * (chunk ^ xs_swz(row)) + half;
        *reinterpret_cast<uint2*>(d) = v0;
        *reinterpret_cast<uint2*>(d + PL) = v1;
        *reinterpret_cast<uint2*>(d + 2 * PL) = v2;
    }
}
// reduction-major 64 (k) x 64 columns: lane = 8*c + a of wave w -> columns 32*(w&1) + 4*a .. +3, k pair 8*(w>>1) + c
struct LoaderRMS {
    const float* base;
    int64_t ld;
    int w, a, c, col0, ncols;
    const float* safe;
    __device__ __forceinline__ void init(const float* base_, int64_t ld_, int col0_, int ncols_, int tid, const float* safe_) {
        base = base_; ld = ld_; safe = safe_; col0 = col0_; ncols = ncols_;
        w = __builtin_amdgcn_readfirstlane(tid >> 6);
        const int lane = tid & 63;
        a = lane & 7;
        c = lane >> 3;
    }
    __device__ __forceinline__ int lrow() const { return 32 * (w & 1) + 4 * a; }
    __device__ __forceinline__ int kp() const { return 8 * (w >> 1) + c; }
    __device__ __forceinline__ bool conv_dead(int k, int col, int convT, int convE) const {
        if (convT <= 0) return false;
        const int t = k % convT;
        return (t == 0 && col < convE) || (t == convT - 1 && col >= 2 * convE);
    }
    __device__ __forceinline__ void load(f32x4 (&r)[2], int k0, int K, int convT, int convE) const {
        const uint32_t rowb = (uint32_t)ld * 4u;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        const char* sf = reinterpret_cast<const char*>(safe);
        const int col = col0 + lrow();
        const bool cvalid = col < ncols;
        const uint32_t off = (uint32_t)(2 * c * (int)ld + (cvalid ? col : col0)) * 4u;
        const char* kb = reinterpret_cast<const char*>(base + (int64_t)(k0 + 16 * (w >> 1)) * ld);      // uniform
        const int k = k0 + 2 * kp();
        const bool o0 = cvalid && k < K && !conv_dead(k, col, convT, convE), o1 = cvalid && k + 1 < K && !conv_dead(k + 1, col, convT, convE);
        const f32x4 v0 = ldg4(o0 ? kb + off : sf), v1 = ldg4(o1 ? kb + off + rowb : sf);
        r[0] = o0 ? v0 : z;
        r[1] = o1 ? v1 : z;
    }
};
__device__ __forceinline__ void store_rm_xs(uint32_t* lds, const f32x4 (&r)[2], const LoaderRMS& L) {
    constexpr int PL = 64 * XS_LDW;
    const int lrow = L.lrow(), kp = L.kp();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = lrow + j;
        uint32_t p0, p1, p2;
        split3(r[0][j], r[1][j], p0, p1, p2);
        uint32_t* d = lds + row * XS_LDW + 4 * ((kp >> 2) ^ xs_swz(row)) + (kp & 3);
        d[0] = p0;
        d[PL] = p1;
        d[2 * PL] = p2;
    }
}

// EIGHT waves: wave w multiplies the k-step 16*(w>>1) .. +15 of the row half w & 1 (a 32 x 64 strip: 12 MFMAs per k-tile) and
// stages an eighth of the tile (16 floats per thread).  Four waves per block ran every k-tile as 24 MFMAs + a 32-float
// split per thread back to back, ~2 500 cycles with nothing else on the CU; halving both is what the small products
// (125 workgroups on 256 CUs) respond to.
// (forward form: 128 registers, so that two 8-wave workgroups share a CU and the 375 / 475-workgroup forward products of the
// user tower run as one round, 19.0 -> 15.5 and 21.0 -> 19.1 us; the other forms need 143-155 registers and lose more to
// spills at 128 than the second workgroup gives them: 25.5 -> 35.3 us for dX at K = 1 200)
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(XS_NT, (A_KC && B_KC) ? 4 : 1) void gemm_x3s_kernel(const GemmParams p) {
    constexpr int PL = 64 * XS_LDW;                        // dwords per plane
    __shared__ __attribute__((aligned(16))) uint32_t smem[6 * PL];   // A planes 0..2, B planes 0..2 (49 152 B); reused by the final exchange
    uint32_t* As = smem;
    uint32_t* Bs = smem + 3 * PL;

    X3_STAMP(0);
    X3_STAMP_RT(1);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int kstep = wid >> 1, rh = wid & 1;              // this wave's k-step inside a k-tile, and its row half

    const int tiles_n = (p.N + 63) / 64;
    const int tiles_m = (p.M + 63) / 64;
    const int ntile = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {                                       // slice-major over the whole grid (see gemm_f32_kernel)
        const int G = ntile * p.ksplit;
        const int q = G >> 3, r = G & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int ks = bid / ntile;
    bid -= ks * ntile;
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * 64, n0 = tn * 64;

    const int kt_total = (p.K + XS_BK - 1) / XS_BK;
    const int kt_per = (kt_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);

    f32x16 acc[2];                                         // tiles (rh, 0) and (rh, 1), partial over this wave's k-steps
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // THREE register sets: the loads of tiles t+1 .. t+3 are in flight while tile t is multiplied.  Tile t travels through
    // set t % 3; indices past the last tile are clamped (a redundant, unused load) so that the loop body has no branch
    // around a load and every wait stays a counted one.
    f32x4 ra[3][2], rb[3][2];
    LoaderKCS lkA, lkB; LoaderRMS lrA, lrB;
    const int convA = (A_KC && p.conv_on == 1) ? p.conv_T : 0;
    const int convB = (!B_KC && p.conv_on == 2) ? p.conv_T : 0;
    const float* safeA = p.A + (convA ? p.conv_E : 0);
    const float* safeB = p.B + (convB ? p.conv_E : 0);
    if (A_KC) lkA.init(p.A, p.lda, m0, p.M, tid, convA, safeA); else lrA.init(p.A, p.lda, m0, p.M, tid, safeA);
    if (B_KC) lkB.init(p.B, p.ldb, n0, p.N, tid, 0, safeB);     else lrB.init(p.B, p.ldb, n0, p.N, tid, safeB);

    const bool do_csum = !A_KC && p.csum != nullptr && tn == 0;
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};

    const int nkt = kt_end - kt_begin;
    auto gload = [&](auto set, int t) {
        constexpr int S = decltype(set)::value;
        const int k0 = (kt_begin + min(t, nkt - 1)) * XS_BK;
        if (A_KC) lkA.load(ra[S], k0, p.K, convA ? p.conv_E : 0); else lrA.load(ra[S], k0, p.K, 0, 0);
        if (B_KC) lkB.load(rb[S], k0, p.K, 0);                    else lrB.load(rb[S], k0, p.K, convB, p.conv_E);
    };
    auto lstore = [&](auto set) {
        constexpr int S = decltype(set)::value;
        if constexpr (A_KC) store_kc_xs(As, ra[S], tid);
        else {
            store_rm_xs(As, ra[S], lrA);
            if (do_csum) cs += ra[S][0] + ra[S][1];
        }
        if constexpr (B_KC) store_kc_xs(Bs, rb[S], tid); else store_rm_xs(Bs, rb[S], lrB);
    };
    auto compute = [&]() {
        bf16x8 fa[3], fb[2][3];
        {
            const int row = rh * 32 + l31;
            const int q4 = 4 * ((2 * kstep + h) ^ xs_swz(row));
#pragma unroll
            for (int t = 0; t < 3; ++t) fa[t] = *reinterpret_cast<const bf16x8*>(As + t * PL + row * XS_LDW + q4);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = j * 32 + l31;
            const int q4 = 4 * ((2 * kstep + h) ^ xs_swz(row));
#pragma unroll
            for (int t = 0; t < 3; ++t) fb[j][t] = *reinterpret_cast<const bf16x8*>(Bs + t * PL + row * XS_LDW + q4);
        }
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};      // smallest products first
#pragma unroll
        for (int t = 0; t < 6; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[TA[t]], fb[j][TB[t]], acc[j], 0, 0, 0);
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    using S2 = std::integral_constant<int, 2>;
    // one tile: reload this tile's (now free) set with tile t+3, multiply tile t out of LDS, then stage tile t+1
    auto step = [&](auto set, auto next, int t) {
        gload(set, t + 3);
        compute();
        if (t < 6) X3_STAMP(3 + 4 * t);
        __syncthreads();
        if (t < 6) X3_STAMP(4 + 4 * t);
        if (t + 1 < nkt) lstore(next);
        if (t < 6) X3_STAMP(5 + 4 * t);
        __syncthreads();
        if (t < 6) X3_STAMP(6 + 4 * t);
    };
    if (nkt > 0) {
        gload(S0{}, 0);
        gload(S1{}, 1);
        gload(S2{}, 2);
        lstore(S0{});
    }
    __syncthreads();
    X3_STAMP(2);
    int t = 0;
    for (; t + 2 < nkt; t += 3) {
        step(S0{}, S1{}, t);
        step(S1{}, S2{}, t + 1);
        step(S2{}, S0{}, t + 2);
    }
    if (t < nkt) { step(S0{}, S1{}, t); ++t; }
    if (t < nkt) { step(S1{}, S2{}, t); ++t; }

    X3_STAMP(28);
    float* fsm = reinterpret_cast<float*>(smem);
    if (do_csum) {
        // a column's k pairs are spread over the lanes' c and the four waves of its column half: shuffles, then one LDS
        // slot per (wave pair, column), summed in a fixed order
        float* dst = p.ksplit > 1 ? p.csum_part + (int64_t)ks * p.M : p.csum;
        if constexpr (!A_KC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v = cs[j];
                v += __shfl_xor(v, 8);
                v += __shfl_xor(v, 16);
                v += __shfl_xor(v, 32);
                if ((lane >> 3) == 0) fsm[(wid >> 1) * 64 + lrA.lrow() + j] = v;
            }
        }
        __syncthreads();
        if (tid < 64 && m0 + tid < p.M) dst[m0 + tid] = ((fsm[tid] + fsm[64 + tid]) + fsm[128 + tid]) + fsm[192 + tid];
        __syncthreads();
    }

    // the partial strips meet: wave w < 4 finishes the 32 x 32 tile (row half w & 1, column half w >> 1); every wave hands
    // the tiles it does not finish to their owners through LDS (12 slots of 4 KB = the staging area), which add the four
    // k-step partials in k-step order -> deterministic
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (kstep != j) {
            const int slot = (2 * j + rh) * 3 + (kstep < j ? kstep : kstep - 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) fsm[slot * 1024 + r * 64 + lane] = acc[j][r];
        }
    }
    __syncthreads();
    if (wid >= 4) return;
    const int jo = wid >> 1;                               // this owner's column half (= its own k-step index)
    f32x16 out[1][1];
    {
        const f32x16 own = jo == 0 ? acc[0] : acc[1];
        f32x16 sum;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) {
            f32x16 part;
            if (s_ == jo) part = own;
            else {
                const int slot = wid * 3 + (s_ < jo ? s_ : s_ - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) part[r] = fsm[slot * 1024 + r * 64 + lane];
            }
            if (s_ == 0) sum = part;
            else {
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[r] += part[r];
            }
        }
        out[0][0] = sum;
    }

    const bool partial = p.ksplit > 1;
    float* Cout = partial ? p.C + (int64_t)ks * p.split_stride : p.C;
    const int mw = m0 + rh * 32, nw = n0 + jo * 32;
    const bool full = (mw + 32 <= p.M) && (nw + 32 <= p.N);
    const bool ub = !partial && p.bias != nullptr;
    const int epi = partial ? EPI_NONE : p.epi;
    if (full) store_block_rt<1, 1, false>(out, Cout, mw, nw, l31, h, ub, epi, p);
    else      store_block_rt<1, 1, true>(out, Cout, mw, nw, l31, h, ub, epi, p);
    X3_STAMP(29);
    X3_STAMP_RT(30);
}

template <bool A_KC, bool B_KC>
static int launch_x3s(const GemmParams& p, hipStream_t s) {
    const int tiles = cdiv(p.M, 64) * cdiv(p.N, 64);
    hipLaunchKernelGGL((gemm_x3s_kernel<A_KC, B_KC>), dim3(tiles * p.ksplit), dim3(XS_NT), 0, s, p);
    return check_launch("gemm_x3s");
}
// products the small-tile split kernel takes: enough 64 x 64 tiles to matter, at least two of its k-tiles
static bool x3s_eligible(int M, int N, int K, int ksplit) {
    static const int min_wgs = getenv("NR_X3S_MIN_WGS") ? atoi(getenv("NR_X3S_MIN_WGS")) : 48;        // A/B switch for tools
    return K >= 128 && (int64_t)cdiv(M, 64) * cdiv(N, 64) * ksplit >= min_wgs;
}

static std::atomic<int> g_gemm_compute{2};       // default: fp32 results from split operands on the bf16 matrix pipe
void set_gemm_compute(int mode) { g_gemm_compute.store(mode); }
int get_gemm_compute() { return g_gemm_compute.load(); }

// sum split-K slabs (fixed order -> deterministic), add bias, apply epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, const float* __restrict__ slabs) {
    const int64_t total = (int64_t)p.M * p.N;
    if (p.csum) {
        for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += (int64_t)gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int s = 0; s < p.ksplit; ++s) v += p.csum_part[(int64_t)s * p.M + m];
            p.csum[m] = v;
        }
    }
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.N), n = (int)(idx - (int64_t)m * p.N);
        float v = 0.f;
        for (int s = 0; s < p.ksplit; ++s) v += slabs[(int64_t)s * p.split_stride + (int64_t)m * p.N + n];
        if (p.bias) v += p.bias[n];
        v = epilogue(v, p.epi, m, n, p);
        p.C[(int64_t)m * p.ldc + n] = v;
    }
}

// the plain case (dW: no bias, no epilogue, 16-byte rows): four outputs per thread, four slabs in flight.  The scalar
// kernel above walks its slabs one dependent load at a time: 47 slabs of dW[300,300] took 24.6 us for 17 MB.
__global__ __launch_bounds__(256) void splitk_reduce4_kernel(const GemmParams p, const float* __restrict__ slabs) {
    const int n4 = p.N >> 2;
    const int64_t total = (int64_t)p.M * n4;
    if (p.csum) {
        for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < p.M; m += (int64_t)gridDim.x * blockDim.x) {
            float v = 0.f;
            for (int s = 0; s < p.ksplit; ++s) v += p.csum_part[(int64_t)s * p.M + m];
            p.csum[m] = v;
        }
    }
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / n4), n = (int)(idx - (int64_t)m * n4) << 2;
        const float* src = slabs + (int64_t)m * p.N + n;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0;
        int s = 0;
        for (; s + 4 <= p.ksplit; s += 4) {                // fixed order: the result does not depend on the launch
            a0 += ldg4(src + (int64_t)s * p.split_stride);
            a1 += ldg4(src + (int64_t)(s + 1) * p.split_stride);
            a2 += ldg4(src + (int64_t)(s + 2) * p.split_stride);
            a3 += ldg4(src + (int64_t)(s + 3) * p.split_stride);
        }
        for (; s < p.ksplit; ++s) a0 += ldg4(src + (int64_t)s * p.split_stride);
        *reinterpret_cast<f32x4*>(p.C + (int64_t)m * p.ldc + n) = (a0 + a1) + (a2 + a3);
    }
}

// slow but fully general fallback (any K/N alignment): one thread per output element
template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_naive_kernel(const GemmParams p) {
    const int64_t total = (int64_t)p.M * p.N;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int m = (int)(idx / p.N), n = (int)(idx - (int64_t)m * p.N);
        float v = 0.f;
        for (int k = 0; k < p.K; ++k) {
            float a, b;
            if (A_KC) a = (p.conv_on == 1 && conv_masked(m, k, p.conv_T, p.conv_E)) ? 0.f : p.A[(int64_t)m * p.lda + k];
            else      a = p.A[(int64_t)k * p.lda + m];
            if (B_KC) b = p.B[(int64_t)n * p.ldb + k];
            else      b = (p.conv_on == 2 && conv_masked(k, n, p.conv_T, p.conv_E)) ? 0.f : p.B[(int64_t)k * p.ldb + n];
            v = fmaf(a, b, v);
        }
        if (p.bias) v += p.bias[n];
        v = epilogue(v, p.epi, m, n, p);
        p.C[(int64_t)m * p.ldc + n] = v;
    }
}

template <bool A_KC, bool B_KC, int WM, int NJ>
static int launch_tiled(const GemmParams& p, hipStream_t s) {
    constexpr int BM = 64 * WM, BN = 64 * NJ;
    constexpr int A_ELEMS = A_KC ? BM * KC_LD : BK * RedMajor<BM>::LD;
    constexpr int B_ELEMS = B_KC ? BN * KC_LD : BK * RedMajor<BN>::LD;
    const size_t lds = (A_ELEMS + B_ELEMS) * sizeof(float);
    const int tiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, WM, NJ>), dim3(tiles * p.ksplit), dim3(256), lds, s, p);
    return check_launch("gemm_f32");
}

// dW GEMMs (both operands reduction-major, split-K): the tall 320-row tile when it pads M less than 128-row tiles
// (only for long reductions: with few k-tiles per split its 80 MFMAs per wave and k-tile serialise on too few CUs)
// and many n-tiles to share the tall A tile through L2: measured on the NRMS shapes it wins for dW[300,1200]
// (85 vs 79 TF/s) and loses for dW[900,300] / dW[300,300] (74 vs 78, 51 vs 59)
static bool tall_tile(int M, int N, int K) {
    static const bool off = getenv("NR_GEMM_NO_TALL") != nullptr;       // A/B switch for tools
    // R = 56 320 rows (the padded NRMS step): 85 vs 79 TF/s for dW[300,1200]; at the ~22 000 rows of the packed step the
    // 64-row tile wins (step 3.255 -> 3.215 ms): long reductions only
    return !off && K >= 40000 && N >= 1024 && cdiv(M, 320) * 320 < cdiv(M, 128) * 128;
}
// ... and 64-row tiles when they pad M less than 128-row ones (M = 300: 320 vs 384 rows; dW[300,300] 52 -> 66 TF/s,
// dW[900,300] 78 -> 83, dW[1200,300] 84 -> 90)
static bool short_tile(int M) { return cdiv(M, 64) * 64 < cdiv(M, 128) * 128; }
static int dw_tile_rows(int M, int N, int K) { return tall_tile(M, N, K) ? 320 : (short_tile(M) ? 64 : 128); }

// outputs that the split kernel's 16 384-element tiles pad badly (300 x 300: 384 x 384 = 1.64x the work) and 64 x 64 tiles
// do not (320 x 320 = 1.14x): the small-tile kernel takes them whatever the grid size
static bool x3_prefers_small(int M, int N) {
    static const bool off = getenv("NR_X3_SMALL_PREF") && atoi(getenv("NR_X3_SMALL_PREF")) == 0;      // A/B switch for tools
    if (off) return false;
    const int shape = x3_tile_shape(M, N);
    const int bm[3] = {256, 128, 64}, bn[3] = {64, 128, 256};
    const int64_t big = (int64_t)cdiv(M, bm[shape]) * bm[shape] * cdiv(N, bn[shape]) * bn[shape];
    const int64_t small = (int64_t)cdiv(M, 64) * 64 * cdiv(N, 64) * 64;
    return 10 * big > 13 * small;
}

size_t gemm_workspace_bytes(int M, int N, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * M * N * sizeof(float) : 0;
}

int choose_ksplit(int M, int N, int K) {
    if (g_gemm_compute.load() == 2) {
        // split-operand kernel: 9-25 tiles of 16 384 elements for the NRMS dW shapes -> slices until ~1.6 workgroups per
        // CU are resident (its occupancy), never shorter than 8 k-tiles
        static const int target = getenv("NR_X3_KS_WGS") ? atoi(getenv("NR_X3_KS_WGS")) : 416;      // A/B switch for tools (swept 200..1024 on the NRMS step: flat optimum 384-448)
        const int shape = x3_tile_shape(M, N);
        const int bm[3] = {256, 128, 64}, bn[3] = {64, 128, 256};
        const int tiles = cdiv(M, bm[shape]) * cdiv(N, bn[shape]);
        const int kt = cdiv(K, BK);
        const int ks = std::min(cdiv(target, tiles), kt / 8);
        if (ks >= 2 && tiles * ks >= 200 && !x3_prefers_small(M, N)) return ks;
        // small-tile split kernel: 64 x 64 tiles, 64-deep k-tiles; ~1.5 workgroups per CU, slices of >= 4 k-tiles
        static const int target_s = getenv("NR_X3S_KS_WGS") ? atoi(getenv("NR_X3S_KS_WGS")) : 384;
        const int tiles_s = cdiv(M, 64) * cdiv(N, 64);
        const int kts = cdiv(K, XS_BK);
        const int ks_s = std::max(1, std::min(cdiv(target_s, tiles_s), kts / 4));
        if (x3s_eligible(M, N, K, ks_s)) return ks_s;
    }
    // dW GEMMs: tiny MxN, huge K.  Two pulls (swept on MI355X): enough workgroups to fill the
    // chip (>= 512), and slices of ~32 k-tiles — shorter slices than the fill needs only add slab traffic, longer
    // ones leave the single-buffered loop latency-bound (dW[1200,300]: 68 TF/s at 512 workgroups, 85 at 2048+)
    const int tiles = cdiv(M, dw_tile_rows(M, N, K)) * cdiv(N, 64);
    const int kt = cdiv(K, BK);
    if (tiles >= 2048 || kt < 8) return 1;   // (288 tiles on 256 CUs would run as two unbalanced rounds: keep splitting)
    int ks = std::max((512 + tiles - 1) / tiles, kt / 32);
    ks = std::min(ks, std::max(1, 3072 / tiles));
    ks = std::max(std::min(ks, kt / 4), 1);
    // an even slice length (32 k-tiles = 1024 rows) starts every slice a multiple of 16 KiB apart: all of them
    // stream through the same L2 channels (dW[300,1200]: 75 TF/s at 32 tiles per slice, 85 at 33)
    int per = cdiv(kt, ks);
    if (ks > 1 && per % 2 == 0) ks = cdiv(kt, per + 1);
    return std::max(ks, 1);
}

bool gemm_fast_path(const GemmParams& p, bool a_kc, bool b_kc) {
    // vector (16 B) loads need 4-float alignment of every row start and of the contiguous extent
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool fast = al(p.A) && al(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
    fast = fast && (a_kc ? (p.K % 4 == 0) : (p.M % 4 == 0)) && (b_kc ? (p.K % 4 == 0) : (p.N % 4 == 0));
    if (p.conv_on) fast = fast && (p.conv_E % 4 == 0);
    return fast && !(!a_kc && b_kc);
}

int gemm_f32(GemmParams p, bool a_kc, bool b_kc, float* workspace, size_t ws_bytes, hipStream_t s) {
    NR_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "gemm: negative dims");
    if (p.M == 0 || p.N == 0) return NR_OK;
    NR_CHECK_ARG(p.A && p.B && p.C, "gemm: null operand");
    if (p.ksplit < 1) p.ksplit = 1;
    const bool fast = gemm_fast_path(p, a_kc, b_kc);
    if (p.csum) NR_CHECK_ARG(fast && !a_kc && (p.ksplit <= 1 || p.csum_part), "gemm: fused column sums need the tiled reduction-major path");
    // mode 2: fp32 products from three bf16 pieces per operand (gemm_x3_kernel) wherever its 16 384-element tiles fill
    // the chip; small products (the user tower: 1 600 rows) stay on the fp32 instruction and its 64-row tiles
    static const int env_shape = getenv("NR_X3_SHAPE") ? atoi(getenv("NR_X3_SHAPE")) : -1;       // A/B switches for tools
    static const int env_minwg = getenv("NR_X3_MIN_WGS") ? atoi(getenv("NR_X3_MIN_WGS")) : 200;
    const int x3_shape = env_shape >= 0 ? env_shape : x3_tile_shape(p.M, p.N);
    const int x3_bm[3] = {256, 128, 64}, x3_bn[3] = {64, 128, 256};
    const bool x3 = fast && g_gemm_compute.load() == 2 && p.K >= 64 && (a_kc || !b_kc) &&
                    (int64_t)cdiv(p.M, x3_bm[x3_shape]) * cdiv(p.N, x3_bn[x3_shape]) * p.ksplit >= env_minwg &&
                    !(p.ksplit > 1 && x3_prefers_small(p.M, p.N));
    const bool x3s = !x3 && fast && g_gemm_compute.load() == 2 && (a_kc || !b_kc) && x3s_eligible(p.M, p.N, p.K, p.ksplit);
    ProfScope prof(x3 ? FAM_GEMM_SPLIT : FAM_GEMM, s, 2.0 * p.M * p.N * p.K);
    if (!fast) {
        p.ksplit = 1;
        const int64_t total = (int64_t)p.M * p.N;
        const int grid = (int)std::min<int64_t>((total + 255) / 256, 8192);
        if (a_kc && b_kc) hipLaunchKernelGGL((gemm_naive_kernel<true, true>), dim3(grid), dim3(256), 0, s, p);
        else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_naive_kernel<true, false>), dim3(grid), dim3(256), 0, s, p);
        else if (!a_kc && !b_kc) hipLaunchKernelGGL((gemm_naive_kernel<false, false>), dim3(grid), dim3(256), 0, s, p);
        else { set_error("gemm: layout (A red-major, B k-contig) not built"); return NR_ERR_UNSUPPORTED; }
        return check_launch("gemm_naive");
    }
    float* slabs = nullptr;
    GemmParams q = p;
    if (p.ksplit > 1) {
        const size_t need = gemm_workspace_bytes(p.M, p.N, p.ksplit);
        if (!workspace || ws_bytes < need) {
            set_error("gemm: split-K workspace too small (%zu < %zu)", ws_bytes, need);
            return NR_ERR_WORKSPACE;
        }
        slabs = workspace;
        q.C = slabs; q.ldc = p.N; q.split_stride = (int64_t)p.M * p.N;
    }
    int rc;
    // bf16-input matrix cores when the caller asked for them (nr_set_gemm_compute(1)) and the product is big enough
    // to matter; small products stay on the exact fp32 kernel
    const bool half = g_gemm_compute.load() == 1 && p.M >= 128 && p.N >= 128 && p.K >= 128;
    if (half) {
        if (a_kc && b_kc)        rc = launch_bf16<true, true>(q, s);
        else if (a_kc && !b_kc)  rc = launch_bf16<true, false>(q, s);
        else                     rc = launch_bf16<false, false>(q, s);
    } else if (x3s) {
        if (a_kc && b_kc)        rc = launch_x3s<true, true>(q, s);
        else if (a_kc && !b_kc)  rc = launch_x3s<true, false>(q, s);
        else                     rc = launch_x3s<false, false>(q, s);
    } else if (x3) {
        const int shape = x3_shape;
        if (a_kc && b_kc)        rc = launch_x3_tile<true, true>(q, shape, s);
        else if (a_kc && !b_kc)  rc = launch_x3_tile<true, false>(q, shape, s);
        else                     rc = launch_x3_tile<false, false>(q, shape, s);
    } else {
    // BN=128 only when it pads no more columns than BN=64: at N=1200 (6.7 % vs 1.3 % padding) the wide tile measured
    // 9 489 vs 9 928 impressions/s on the whole step (r02 A/B), the extra columns cost more than the saved A re-reads
    const bool wide = (cdiv(p.N, 128) * 128 == cdiv(p.N, 64) * 64) && p.N >= 128;
    // few 128-row tiles (the user tower: 1600 history rows) leave most CUs idle: halve the tile
    // ... and so does a short, narrow product (N <= 320, K <= 512: out-proj / pooling projection at 300-d, +5-9 %):
    // ten k-tiles per workgroup are mostly prologue/epilogue, which 7 resident waves per SIMD overlap better
    const bool small = (int64_t)cdiv(p.M, 128) * cdiv(p.N, 64) * p.ksplit < 192 || (p.N <= 320 && p.K <= 512);
    if (a_kc && b_kc)        rc = small ? launch_tiled<true, true, 1, 1>(q, s) : wide ? launch_tiled<true, true, 2, 2>(q, s) : launch_tiled<true, true, 2, 1>(q, s);
    // dX (B reduction-major): 128x64 at 4 workgroups/CU also wins on N % 128 == 0 (768-d: 124-128 vs 117-123 TF/s)
    else if (a_kc && !b_kc)  rc = small ? launch_tiled<true, false, 1, 1>(q, s) : launch_tiled<true, false, 2, 1>(q, s);
    else if (!a_kc && !b_kc) {
        if (!p.conv_on && !wide && (p.epi == EPI_NONE || p.ksplit > 1) && tall_tile(p.M, p.N, p.K)) rc = launch_tiled<false, false, 5, 1>(q, s);
        else if (!wide && short_tile(p.M)) rc = launch_tiled<false, false, 1, 1>(q, s);
        else rc = wide ? launch_tiled<false, false, 2, 2>(q, s) : launch_tiled<false, false, 2, 1>(q, s);
    }
    else { set_error("gemm: layout (A red-major, B k-contig) not built"); return NR_ERR_UNSUPPORTED; }
    }
    if (rc != NR_OK) return rc;
    if (p.ksplit > 1) {
        GemmParams r = p;
        r.split_stride = (int64_t)p.M * p.N;
        const int64_t total = (int64_t)p.M * p.N;
        if (p.epi == EPI_NONE && !p.bias && (p.N & 3) == 0 && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(p.C) & 15) == 0 &&
            (reinterpret_cast<uintptr_t>(slabs) & 15) == 0) {
            const int grid = (int)std::min<int64_t>((total / 4 + 255) / 256, 4096);
            hipLaunchKernelGGL(splitk_reduce4_kernel, dim3(grid), dim3(256), 0, s, r, slabs);
            return check_launch("splitk_reduce4");
        }
        const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, r, slabs);
        return check_launch("splitk_reduce");
    }
    return NR_OK;
}

}  // namespace nr
