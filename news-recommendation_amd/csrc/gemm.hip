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
// Block: 256 threads = 4 waves (2x2), tile 128 x (64*NJ... BN) x 32, LDS double
// buffered with register prefetch (one barrier per k-tile).  Each wave owns a
// 64 x (BN/2) sub-tile = 2 x NJ MFMA tiles of 32x32 (acc in 32*NJ registers).
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

namespace nr {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int KC_LD = BK + 4;        // floats per LDS row of a k-contiguous tile

template <int COLS> struct RedMajor { static constexpr int LD = COLS + 4; };

__device__ __forceinline__ bool conv_masked(int r, int c, int T, int E) {
    // r: token row (m*T+t); c: im2col column in [0,3E)
    const int t = r % T;
    return (t == 0 && c < E) || (t == T - 1 && c >= 2 * E);
}

// ---- global -> register staging -------------------------------------------------
// k-contiguous operand: tile ROWS x 32 floats, thread loads float4 (row = t/8 + 32*i, k = (t%8)*4)
template <int ROWS>
__device__ __forceinline__ void load_kc(float4 (&reg)[ROWS / 32], const float* __restrict__ base, int64_t ld,
                                        int row0, int nrows, int k0, int K, int tid, int convT, int convE) {
    const int c4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i) {
        const int r = row0 + r0 + 32 * i;
        const int k = k0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = (r < nrows) && (k < K);
        if (convT > 0 && ok) ok = !conv_masked(r, k, convT, convE);
        if (ok) v = *reinterpret_cast<const float4*>(base + (int64_t)r * ld + k);
        reg[i] = v;
    }
}
template <int ROWS>
__device__ __forceinline__ void store_kc(float* lds, const float4 (&reg)[ROWS / 32], int tid) {
    const int c4 = (tid & 7) * 4;
    const int r0 = tid >> 3;
#pragma unroll
    for (int i = 0; i < ROWS / 32; ++i)
        *reinterpret_cast<float4*>(lds + (r0 + 32 * i) * KC_LD + c4) = reg[i];
}

// reduction-major operand: tile 32 (k) x COLS floats
template <int COLS>
__device__ __forceinline__ void load_rm(float4 (&reg)[COLS / 32], const float* __restrict__ base, int64_t ld,
                                        int col0, int ncols, int k0, int K, int tid, int convT, int convE) {
    constexpr int TPR = COLS / 4;          // threads per k-row
    constexpr int KPP = 256 / TPR;         // k-rows per pass
    const int c4 = (tid % TPR) * 4;
    const int kr0 = tid / TPR;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i) {
        const int k = k0 + kr0 + KPP * i;
        const int c = col0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        bool ok = (k < K) && (c < ncols);
        if (convT > 0 && ok) ok = !conv_masked(k, c, convT, convE);
        if (ok) v = *reinterpret_cast<const float4*>(base + (int64_t)k * ld + c);
        reg[i] = v;
    }
}
template <int COLS>
__device__ __forceinline__ void store_rm(float* lds, const float4 (&reg)[COLS / 32], int tid) {
    constexpr int TPR = COLS / 4;
    constexpr int KPP = 256 / TPR;
    const int c4 = (tid % TPR) * 4;
    const int kr0 = tid / TPR;
#pragma unroll
    for (int i = 0; i < COLS / 32; ++i)
        *reinterpret_cast<float4*>(lds + (kr0 + KPP * i) * RedMajor<COLS>::LD + c4) = reg[i];
}

__device__ __forceinline__ float epilogue(float v, int epi, int m, int n, const GemmParams& p) {
    switch (epi) {
        case EPI_NONE: return v;
        case EPI_RELU: return fmaxf(v, 0.f);
        case EPI_GELU:
            if (p.out2) p.out2[(int64_t)m * p.ldout2 + n] = v;
            return gelu_f(v);
        case EPI_TANH: return tanhf(v);
        case EPI_ADD_AUX: return v + p.aux[(int64_t)m * p.ldaux + n];
        case EPI_MUL_GELU_GRAD: return v * gelu_grad_f(p.aux[(int64_t)m * p.ldaux + n]);
        case EPI_ADD_AUX_RELUMASK: {
            const float g = v + p.aux[(int64_t)m * p.ldaux + n];
            return p.aux2[(int64_t)m * p.ldaux2 + n] > 0.f ? g : 0.f;
        }
        case EPI_RELUMASK: return p.aux2[(int64_t)m * p.ldaux2 + n] > 0.f ? v : 0.f;
    }
    return v;
}

template <bool A_KC, bool B_KC, int NJ>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmParams p) {
    constexpr int BN = 64 * NJ;
    constexpr int A_ELEMS = A_KC ? BM * KC_LD : BK * RedMajor<BM>::LD;
    constexpr int B_ELEMS = B_KC ? BN * KC_LD : BK * RedMajor<BN>::LD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][A_ELEMS]
    float* Bs = smem + 2 * A_ELEMS;         // [2][B_ELEMS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int l31 = lane & 31, h = lane >> 5;

    // XCD-aware tile order: blocks b and b+8 share an XCD/L2, so give each XCD a
    // contiguous run of tiles (n fastest) -> the A row-panel is re-read from L2.
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int ntile = tiles_m * tiles_n;
    int bid = blockIdx.x;
    const int ks = bid / ntile;             // split-K slice
    bid -= ks * ntile;
    {
        const int q = ntile >> 3, r = ntile & 7, x = bid & 7, loc = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + loc;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    // k range of this split
    const int kt_total = (p.K + BK - 1) / BK;
    const int kt_per = (kt_total + p.ksplit - 1) / p.ksplit;
    const int kt_begin = ks * kt_per;
    const int kt_end = min(kt_total, kt_begin + kt_per);

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[BM / 32];
    float4 rb[BN / 32];
    const int convA = (p.conv_on == 1) ? p.conv_T : 0;
    const int convB = (p.conv_on == 2) ? p.conv_T : 0;

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        if (A_KC) load_kc<BM>(ra, p.A, p.lda, m0, p.M, k0, p.K, tid, convA, p.conv_E);
        else      load_rm<BM>(ra, p.A, p.lda, m0, p.M, k0, p.K, tid, 0, 0);
        if (B_KC) load_kc<BN>(rb, p.B, p.ldb, n0, p.N, k0, p.K, tid, 0, 0);
        else      load_rm<BN>(rb, p.B, p.ldb, n0, p.N, k0, p.K, tid, convB, p.conv_E);
    };
    auto lstore = [&](int buf) {
        if (A_KC) store_kc<BM>(As + buf * A_ELEMS, ra, tid);
        else      store_rm<BM>(As + buf * A_ELEMS, ra, tid);
        if (B_KC) store_kc<BN>(Bs + buf * B_ELEMS, rb, tid);
        else      store_rm<BN>(Bs + buf * B_ELEMS, rb, tid);
    };

    if (kt_begin < kt_end) {
        gload(kt_begin);
        lstore(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = kt_begin; kt < kt_end; ++kt) {
        const bool more = kt + 1 < kt_end;
        if (more) gload(kt + 1);

        const float* Ab = As + cur * A_ELEMS;
        const float* Bb = Bs + cur * B_ELEMS;
        const int krem = min(BK, p.K - kt * BK);
        const int nkq = (krem + 7) >> 3;
        auto kstep = [&](int kq) {
            float a[2][4], b[NJ][4];
            const int kb = kq * 8 + 4 * h;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wr * 64 + i * 32 + l31;
                if (A_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(Ab + row * KC_LD + kb);
                    a[i][0] = v.x; a[i][1] = v.y; a[i][2] = v.z; a[i][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][e] = Ab[(kb + e) * RedMajor<BM>::LD + row];
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int col = wc * (BN / 2) + j * 32 + l31;
                if (B_KC) {
                    const float4 v = *reinterpret_cast<const float4*>(Bb + col * KC_LD + kb);
                    b[j][0] = v.x; b[j][1] = v.y; b[j][2] = v.z; b[j][3] = v.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[j][e] = Bb[(kb + e) * RedMajor<BN>::LD + col];
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        };
        if (nkq == BK / 8) {
#pragma unroll
            for (int kq = 0; kq < BK / 8; ++kq) kstep(kq);
        } else {
            for (int kq = 0; kq < nkq; ++kq) kstep(kq);
        }
        if (more) lstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: C/D map col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    const bool partial = p.ksplit > 1;
    float* Cout = partial ? p.C + (int64_t)ks * p.split_stride : p.C;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = n0 + wc * (BN / 2) + j * 32 + l31;
            if (n >= p.N) continue;
            const float bv = (!partial && p.bias) ? p.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wr * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m >= p.M) continue;
                float v = acc[i][j][r] + bv;
                if (!partial) v = epilogue(v, p.epi, m, n, p);
                Cout[(int64_t)m * p.ldc + n] = v;
            }
        }
    }
}

// sum split-K slabs (fixed order -> deterministic), add bias, apply epilogue
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const GemmParams p, const float* __restrict__ slabs) {
    const int64_t total = (int64_t)p.M * p.N;
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

template <bool A_KC, bool B_KC, int NJ>
static int launch_tiled(const GemmParams& p, hipStream_t s) {
    constexpr int BN = 64 * NJ;
    constexpr int A_ELEMS = A_KC ? BM * KC_LD : BK * RedMajor<BM>::LD;
    constexpr int B_ELEMS = B_KC ? BN * KC_LD : BK * RedMajor<BN>::LD;
    const size_t lds = 2 * (A_ELEMS + B_ELEMS) * sizeof(float);
    const int tiles = cdiv(p.M, BM) * cdiv(p.N, BN);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_f32_kernel<A_KC, B_KC, NJ>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_kernel<A_KC, B_KC, NJ>), dim3(tiles * p.ksplit), dim3(256), lds, s, p);
    return check_launch("gemm_f32");
}

size_t gemm_workspace_bytes(int M, int N, int ksplit) {
    return ksplit > 1 ? (size_t)ksplit * M * N * sizeof(float) : 0;
}

int choose_ksplit(int M, int N, int K) {
    // fill ~2 waves of 256 CUs when the output grid alone cannot (dW GEMMs: tiny MxN, huge K)
    const int tiles = cdiv(M, BM) * cdiv(N, 64);
    const int kt = cdiv(K, BK);
    if (tiles >= 256 || kt < 16) return 1;
    int ks = (512 + tiles - 1) / tiles;
    ks = std::min(ks, kt / 8);
    return std::max(ks, 1);
}

int gemm_f32(GemmParams p, bool a_kc, bool b_kc, float* workspace, size_t ws_bytes, hipStream_t s) {
    NR_CHECK_ARG(p.M >= 0 && p.N >= 0 && p.K >= 0, "gemm: negative dims");
    if (p.M == 0 || p.N == 0) return NR_OK;
    NR_CHECK_ARG(p.A && p.B && p.C, "gemm: null operand");
    if (p.ksplit < 1) p.ksplit = 1;
    // vector (16 B) loads need 4-float alignment of every row start and of the contiguous extent
    auto al = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
    bool fast = al(p.A) && al(p.B) && (p.lda % 4 == 0) && (p.ldb % 4 == 0);
    fast = fast && (a_kc ? (p.K % 4 == 0) : (p.M % 4 == 0)) && (b_kc ? (p.K % 4 == 0) : (p.N % 4 == 0));
    if (p.conv_on) fast = fast && (p.conv_E % 4 == 0);
    ProfScope prof(FAM_GEMM, s, 2.0 * p.M * p.N * p.K);
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
    // BN=128 only when it wastes no more columns than BN=64
    const bool wide = (cdiv(p.N, 128) * 128 == cdiv(p.N, 64) * 64) && p.N >= 128;
    int rc;
    if (a_kc && b_kc)        rc = wide ? launch_tiled<true, true, 2>(q, s) : launch_tiled<true, true, 1>(q, s);
    else if (a_kc && !b_kc)  rc = wide ? launch_tiled<true, false, 2>(q, s) : launch_tiled<true, false, 1>(q, s);
    else if (!a_kc && !b_kc) rc = wide ? launch_tiled<false, false, 2>(q, s) : launch_tiled<false, false, 1>(q, s);
    else { set_error("gemm: layout (A red-major, B k-contig) not built"); return NR_ERR_UNSUPPORTED; }
    if (rc != NR_OK) return rc;
    if (p.ksplit > 1) {
        GemmParams r = p;
        r.split_stride = (int64_t)p.M * p.N;
        const int64_t total = (int64_t)p.M * p.N;
        const int grid = (int)std::min<int64_t>((total + 255) / 256, 4096);
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, s, r, slabs);
        return check_launch("splitk_reduce");
    }
    return NR_OK;
}

}  // namespace nr
