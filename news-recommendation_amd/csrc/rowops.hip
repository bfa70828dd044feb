// HBM-bound row kernels: embedding gather / scatter-add, residual+dropout+LayerNorm
// (fwd/bwd), column sums, GELU-free elementwise helpers, Adam.
// One wave (64 lanes) owns one row; a row is read/written as 16-byte vectors,
// i.e. every wave instruction touches 1 KiB of one row (coalesced).
#include "common.h"
#include "rowops.h"
#include <mutex>

namespace nr {

constexpr int WAVES_PER_BLOCK = 4;

static inline bool aligned16(const void* a, const void* b = nullptr, const void* c = nullptr, const void* d = nullptr) {
    return ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c) |
             reinterpret_cast<uintptr_t>(d)) & 15) == 0;
}

// ------------------------------------------------------------------ index flag
static __device__ int g_index_flag;
int* index_flag_ptr() {
    static std::mutex mu;
    static int* per_dev[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    if (!per_dev[dev]) {
        // looked up again on the next call if it fails now (a first call under stream capture, say): a failure must not
        // switch the out-of-range detection off for the rest of the process.  No memset: a __device__ int starts at 0.
        int* ptr = nullptr;
        if (hipGetSymbolAddress(reinterpret_cast<void**>(&ptr), HIP_SYMBOL(g_index_flag)) == hipSuccess && ptr) per_dev[dev] = ptr;
        else (void)hipGetLastError();
    }
    return per_dev[dev];
}
int index_status(int* host_out, hipStream_t s) {
    NR_CHECK_ARG(host_out, "nr_index_status: null output");
    *host_out = 0;
    int* flag = index_flag_ptr();
    if (!flag) {        // the gathers ran without a flag: their clamped indices went unreported, say so
        set_error("nr_index_status: the per-GPU index flag is unavailable (hipGetSymbolAddress failed): out-of-range indices cannot be reported");
        return NR_ERR_LAUNCH;
    }
    if (hipMemcpyAsync(host_out, flag, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess || hipMemsetAsync(flag, 0, sizeof(int), s) != hipSuccess ||
        hipStreamSynchronize(s) != hipSuccess) {
        set_error("nr_index_status: reading the flag failed");
        return NR_ERR_LAUNCH;
    }
    return NR_OK;
}

// ------------------------------------------------------------------ gather
// out[r, :] = table[ids[r], :]   (bit-exact copy).  4 rows in flight per wave.
template <int VEC>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table, const int64_t* __restrict__ ids,
                                                          float* __restrict__ out, int64_t n, int E, int64_t V,
                                                          int* __restrict__ oob_flag) {
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    const int64_t nwave = (int64_t)gridDim.x * WAVES_PER_BLOCK;
    constexpr int RIF = 4;   // rows in flight
    for (int64_t r0 = wave * RIF; r0 < n; r0 += nwave * RIF) {
        int64_t id[RIF];
#pragma unroll
        for (int k = 0; k < RIF; ++k) {
            const int64_t r = r0 + k;
            int64_t v = (r < n) ? ids[r] : 0;
            if (v < 0 || v >= V) {            // never fault on a bad index: flag it, read row 0
                if (lane == 0 && oob_flag) *oob_flag = 1;
                v = 0;
            }
            id[k] = v;
        }
        if (VEC == 4) {
            const int nv = E >> 2;
            for (int c = lane; c < nv; c += 64) {
                float4 v[RIF];
#pragma unroll
                for (int k = 0; k < RIF; ++k)
                    v[k] = reinterpret_cast<const float4*>(table + id[k] * (int64_t)E)[c];
#pragma unroll
                for (int k = 0; k < RIF; ++k)
                    if (r0 + k < n) reinterpret_cast<float4*>(out + (r0 + k) * (int64_t)E)[c] = v[k];
            }
        } else {
            for (int c = lane; c < E; c += 64) {
#pragma unroll
                for (int k = 0; k < RIF; ++k)
                    if (r0 + k < n) out[(r0 + k) * (int64_t)E + c] = table[id[k] * (int64_t)E + c];
            }
        }
    }
}

int gather_rows(const float* table, int64_t V, int E, const int64_t* ids, int64_t n, float* out, int* oob_flag,
                hipStream_t s) {
    if (n == 0) return NR_OK;
    if (!oob_flag) oob_flag = index_flag_ptr();            // no caller flag: the library's own (nr_index_status)
    ProfScope prof(FAM_GATHER, s, 2.0 * n * E * 4 + 8.0 * n);
    const bool vec = (E % 4 == 0) && ((reinterpret_cast<uintptr_t>(table) | reinterpret_cast<uintptr_t>(out)) % 16 == 0);
    const int grid = (int)std::min<int64_t>(cdiv(n, WAVES_PER_BLOCK * 4), 8192);
    if (vec) hipLaunchKernelGGL(gather_rows_kernel<4>, dim3(grid), dim3(256), 0, s, table, ids, out, n, E, V, oob_flag);
    else     hipLaunchKernelGGL(gather_rows_kernel<1>, dim3(grid), dim3(256), 0, s, table, ids, out, n, E, V, oob_flag);
    return check_launch("gather_rows");
}

// grad_table[ids[r], :] += grad_rows[r, :]  (rows with id == padding_idx skipped: nn.Embedding(padding_idx) semantics).
// 256 contiguous bytes per atomic wave-instruction.  A few ids are in almost every title ([CLS], [SEP], frequent words) and
// the atomics onto one table row serialise in L2 (measured: two such ids doubled the kernel's time), so a workgroup first
// sums the rows of its 128-row window that share an id in LDS (up to SA_SLOTS distinct repeated ids per window; any
// further ones go straight to the table, which is always correct) and sends one row of atomics per repeated id.
constexpr int SA_WIN = 128;
constexpr int SA_SLOTS = 8;
constexpr int SA_NT = 1024;                       // 16 waves: 8 rows of the window each
__global__ __launch_bounds__(SA_NT) void scatter_add_rows_kernel(const float* __restrict__ g, const int64_t* __restrict__ ids,
                                                               float* __restrict__ gt, int64_t n, int E, int64_t V,
                                                               int64_t padding_idx) {
    extern __shared__ float sa_acc[];                 // [SA_SLOTS][E]
    __shared__ int sid[SA_WIN];                       // id of the window's row, -1 = skipped
    __shared__ int row_slot[SA_WIN];                  // LDS slot of the row's id, -1 = none (atomics go to the table)
    __shared__ int slot_id[SA_SLOTS];
    __shared__ int nslots;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int64_t nwin = (n + SA_WIN - 1) / SA_WIN;
    for (int64_t w = blockIdx.x; w < nwin; w += gridDim.x) {
        const int64_t r0 = w * SA_WIN;
        const int cnt = (int)min((int64_t)SA_WIN, n - r0);
        if (tid < SA_WIN) {
            int v = -1;
            if (tid < cnt) {
                const int64_t id = ids[r0 + tid];
                if (id != padding_idx && id >= 0 && id < V) v = (int)id;
            }
            sid[tid] = v;
        }
        if (tid == 0) nslots = 0;
        for (int c = tid; c < SA_SLOTS * E; c += SA_NT) sa_acc[c] = 0.f;
        __syncthreads();
        // the first row of every repeated id claims a slot ...
        int lead = -1;
        if (tid < cnt && sid[tid] >= 0) {
            const int my = sid[tid];
            int dup = 0;
            for (int j = cnt - 1; j >= 0; --j)
                if (sid[j] == my) { lead = j; ++dup; }
            int slot = -1;
            if (lead == tid && dup > 1) {
                slot = atomicAdd(&nslots, 1);
                if (slot < SA_SLOTS) slot_id[slot] = my; else slot = -1;
            }
            if (lead == tid) row_slot[tid] = slot;
        }
        __syncthreads();
        // ... and the other rows of that id follow it
        if (lead >= 0 && lead != tid) row_slot[tid] = row_slot[lead];
        __syncthreads();
        for (int r = wv; r < cnt; r += SA_NT / 64) {
            const int id = sid[r];
            if (id < 0) continue;
            const float* src = g + (r0 + r) * (int64_t)E;
            const int slot = row_slot[r];
            if (slot >= 0) {
                float* dst = sa_acc + slot * E;
                for (int c = lane; c < E; c += 64) atomicAdd(dst + c, src[c]);
            } else {
                float* dst = gt + (int64_t)id * E;
                for (int c = lane; c < E; c += 64) atomicAdd(dst + c, src[c]);
            }
        }
        __syncthreads();
        const int ns = min(nslots, SA_SLOTS);
        for (int c = tid; c < ns * E; c += SA_NT) {
            const int sl = c / E;
            atomicAdd(gt + (int64_t)slot_id[sl] * E + (c - sl * E), sa_acc[c]);
        }
        __syncthreads();
    }
}

int scatter_add_rows(const float* grad_rows, const int64_t* ids, int64_t n, int E, int64_t V, int64_t padding_idx,
                     float* grad_table, hipStream_t s) {
    if (n == 0) return NR_OK;
    NR_UNSUPPORTED((size_t)SA_SLOTS * E * sizeof(float) > 60 * 1024, "scatter_add_rows: row width %d > %d", E, 60 * 1024 / 4 / SA_SLOTS);
    ProfScope prof(FAM_SCATTER, s, 2.0 * n * E * 4);
    const int grid = (int)std::min<int64_t>(cdiv(n, SA_WIN), 8192);
    hipLaunchKernelGGL(scatter_add_rows_kernel, dim3(grid), dim3(SA_NT), (size_t)SA_SLOTS * E * sizeof(float), s, grad_rows, ids, grad_table, n,
                       E, V, padding_idx);
    return check_launch("scatter_add_rows");
}

// ------------------------------------------------------------------ LayerNorm
// y = LN(drop(h) + x) * gamma + beta ; optionally stores s = drop(h) + x (may alias h),
// mean and rstd.  VEC=4 needs D % 4 == 0.  D <= 64*VEC*MAXIT.
constexpr int LN_MAXIT = 8;
typedef __bf16 rbf16x2 __attribute__((ext_vector_type(2)));
typedef float rf32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    const rf32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, rbf16x2));        // RNE, v_cvt_pk_bf16_f32
}

template <int VEC, int IT>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ h, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ y, float* s_out, float* __restrict__ mean_out,
                                                     float* __restrict__ rstd_out, int64_t R, int D, float eps,
                                                     DropoutCfg dc, int64_t x_period, unsigned short* __restrict__ y_h) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (row >= R) return;
    const float* hr = h + row * D;
    const int64_t prow = dc.row_map ? (int64_t)dc.row_map[row] : row;       // Philox element index: the ORIGINAL row
    const float* xr = x ? x + (x_period > 0 ? prow % x_period : row) * D : nullptr;
    float v[IT][VEC];
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < D) {
            float t[VEC];
            if (VEC == 4) {
                const float4 a = *reinterpret_cast<const float4*>(hr + c);
                t[0] = a.x; t[1 % VEC] = a.y; t[2 % VEC] = a.z; t[3 % VEC] = a.w;
            } else t[0] = hr[c];
            if (dc.p > 0.f) {
                if (VEC == 4) {
                    const Philox4 rn = philox4x32_10(dc.seed, (uint64_t)(prow * D + c) >> 2, dc.site);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) t[e] = philox_pick(rn, e) >= thr ? t[e] * scale : 0.f;
                } else {
                    const uint64_t idx = (uint64_t)(prow * D + c);
                    const Philox4 rn = philox4x32_10(dc.seed, idx >> 2, dc.site);
                    t[0] = philox_pick(rn, (int)(idx & 3)) >= thr ? t[0] * scale : 0.f;
                }
            }
            if (xr) {
                if (VEC == 4) {
                    const float4 b = *reinterpret_cast<const float4*>(xr + c);
                    t[0] += b.x; t[1 % VEC] += b.y; t[2 % VEC] += b.z; t[3 % VEC] += b.w;
                } else t[0] += xr[c];
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) { v[it][e] = t[e]; sum += t[e]; }
        } else {
#pragma unroll
            for (int e = 0; e < VEC; ++e) v[it][e] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < D) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = v[it][e] - mean; sq += d * d; }
        }
    }
    const float var = wave_sum(sq) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * VEC;
        if (c < D) {
            float o[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) o[e] = (v[it][e] - mean) * rstd * gamma[c + e] + beta[c + e];
            if (VEC == 4) {
                *reinterpret_cast<float4*>(y + row * D + c) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
                if (y_h) *reinterpret_cast<uint2*>(y_h + row * D + c) = make_uint2(pack2_bf16(o[0], o[1 % VEC]), pack2_bf16(o[2 % VEC], o[3 % VEC]));
                if (s_out) *reinterpret_cast<float4*>(s_out + row * D + c) =
                    make_float4(v[it][0], v[it][1 % VEC], v[it][2 % VEC], v[it][3 % VEC]);
            } else {
                y[row * D + c] = o[0];
                if (y_h) y_h[row * D + c] = (unsigned short)(pack2_bf16(o[0], 0.f) & 0xffffu);
                if (s_out) s_out[row * D + c] = v[it][0];
            }
        }
    }
}

int ln_fwd(const float* h, const float* x, const float* gamma, const float* beta, float* y, float* s_out, float* mean,
           float* rstd, int64_t R, int D, float eps, DropoutCfg dc, hipStream_t s, int64_t x_period, unsigned short* y_h) {
    if (R == 0) return NR_OK;
    NR_UNSUPPORTED(D > 64 * 4 * LN_MAXIT, "layernorm: D=%d > %d", D, 64 * 4 * LN_MAXIT);
    ProfScope prof(FAM_LN, s, (double)R * D * 4 * (x ? 4 : 3));
    const int grid = cdiv(R, WAVES_PER_BLOCK);
    const bool vec = (D % 4 == 0) && aligned16(h, x, y, s_out);
#define NR_LN_FWD(V, I) hipLaunchKernelGGL((ln_fwd_kernel<V, I>), dim3(grid), dim3(256), 0, s, h, x, gamma, beta, y, s_out, mean, rstd, R, D, eps, dc, x_period, y_h)
    if (vec) {
        if (D <= 512) NR_LN_FWD(4, 2); else if (D <= 1024) NR_LN_FWD(4, 4); else NR_LN_FWD(4, 8);
    } else {
        NR_UNSUPPORTED(D > 64 * 32, "layernorm (unaligned path): D=%d > 2048", D);
        if (D <= 512) NR_LN_FWD(1, 8); else NR_LN_FWD(1, 32);
    }
#undef NR_LN_FWD
    return check_launch("ln_fwd");
}

// backward: given dy and the saved pre-LN input s (+mean, rstd):
//   ds  = rstd * (g - mean(g) - xhat * mean(g*xhat)),  g = dy*gamma   (grad wrt s; also the residual grad)
//   dh  = keep ? ds/(1-p) : 0                                          (grad wrt the dropped branch, optional)
//   dgamma_part[blk] += dy*xhat ; dbeta_part[blk] += dy               (reduced by colsum afterwards)
template <int VEC, int IT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ sin,
                                                     const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                     const float* __restrict__ gamma, float* __restrict__ ds,
                                                     float* __restrict__ dh, float* __restrict__ dgamma_part,
                                                     float* __restrict__ dbeta_part, int64_t R, int D, DropoutCfg dc,
                                                     unsigned short* __restrict__ dh_h) {
    __shared__ float red[WAVES_PER_BLOCK][2][64 * 4 * LN_MAXIT / 4];   // sized for D<=512 via chunks (see below)
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;
    float ag[IT][VEC], ab[IT][VEC], gm[IT][VEC];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            ag[it][e] = 0.f; ab[it][e] = 0.f;
            gm[it][e] = (c + e < D) ? gamma[c + e] : 0.f;
        }
    }
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = dc.p > 0.f ? 1.0f / (1.0f - dc.p) : 1.0f;
    for (int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + wid; row < R; row += (int64_t)gridDim.x * WAVES_PER_BLOCK) {
        const float mean = mean_in[row], rstd = rstd_in[row];
        const int64_t prow = dc.row_map ? (int64_t)dc.row_map[row] : row;
        float g[IT][VEC], xh[IT][VEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < D) {
                float a[VEC], b[VEC];
                if (VEC == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(dy + row * D + c);
                    const float4 u = *reinterpret_cast<const float4*>(sin + row * D + c);
                    a[0] = t.x; a[1 % VEC] = t.y; a[2 % VEC] = t.z; a[3 % VEC] = t.w;
                    b[0] = u.x; b[1 % VEC] = u.y; b[2 % VEC] = u.z; b[3 % VEC] = u.w;
                } else { a[0] = dy[row * D + c]; b[0] = sin[row * D + c]; }
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float x = (b[e] - mean) * rstd;
                    const float gg = a[e] * gm[it][e];
                    xh[it][e] = x; g[it][e] = gg;
                    s1 += gg; s2 += gg * x;
                    ag[it][e] += a[e] * x; ab[it][e] += a[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) { xh[it][e] = 0.f; g[it][e] = 0.f; }
            }
        }
        const float c1 = wave_sum(s1) / (float)D;
        const float c2 = wave_sum(s2) / (float)D;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < D) {
                float o[VEC], od[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) { o[e] = rstd * (g[it][e] - c1 - xh[it][e] * c2); od[e] = o[e]; }
                if (dh && dc.p > 0.f) {
                    if (VEC == 4) {
                        const Philox4 rn = philox4x32_10(dc.seed, (uint64_t)(prow * D + c) >> 2, dc.site);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) od[e] = philox_pick(rn, e) >= thr ? o[e] * scale : 0.f;
                    } else {
                        const uint64_t idx = (uint64_t)(prow * D + c);
                        const Philox4 rn = philox4x32_10(dc.seed, idx >> 2, dc.site);
                        od[0] = philox_pick(rn, (int)(idx & 3)) >= thr ? o[0] * scale : 0.f;
                    }
                }
                if (VEC == 4) {
                    *reinterpret_cast<float4*>(ds + row * D + c) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
                    if (dh) *reinterpret_cast<float4*>(dh + row * D + c) = make_float4(od[0], od[1 % VEC], od[2 % VEC], od[3 % VEC]);
                    if (dh_h) *reinterpret_cast<uint2*>(dh_h + row * D + c) = make_uint2(pack2_bf16(od[0], od[1 % VEC]), pack2_bf16(od[2 % VEC], od[3 % VEC]));
                } else {
                    ds[row * D + c] = o[0];
                    if (dh) dh[row * D + c] = od[0];
                    if (dh_h) dh_h[row * D + c] = (unsigned short)(pack2_bf16(od[0], 0.f) & 0xffffu);
                }
            }
        }
    }
    // cross-wave reduction of the per-column partials, 512 columns at a time
    for (int base = 0; base < D; base += 512) {
        __syncthreads();
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int c = (it * 64 + lane) * VEC;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const int cc = c + e - base;
                if (cc >= 0 && cc < 512 && c + e < D) { red[wid][0][cc] = ag[it][e]; red[wid][1][cc] = ab[it][e]; }
            }
        }
        __syncthreads();
        for (int cc = threadIdx.x; cc < 512 && base + cc < D; cc += 256) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WAVES_PER_BLOCK; ++w) { a += red[w][0][cc]; b += red[w][1][cc]; }
            dgamma_part[(int64_t)blockIdx.x * D + base + cc] = a;
            dbeta_part[(int64_t)blockIdx.x * D + base + cc] = b;
        }
    }
}

int ln_bwd_blocks(int64_t R) { return (int)std::min<int64_t>(cdiv(R, WAVES_PER_BLOCK), 1024); }

int ln_bwd(const float* dy, const float* sin, const float* mean, const float* rstd, const float* gamma, float* ds,
           float* dh, float* dgamma_part, float* dbeta_part, int64_t R, int D, DropoutCfg dc, hipStream_t s, unsigned short* dh_h) {
    if (R == 0) return NR_OK;
    NR_UNSUPPORTED(D > 64 * 4 * LN_MAXIT, "layernorm bwd: D=%d > %d", D, 64 * 4 * LN_MAXIT);
    ProfScope prof(FAM_LN, s, (double)R * D * 4 * (dh ? 4 : 3));
    const int grid = ln_bwd_blocks(R);
    const bool vec = (D % 4 == 0) && aligned16(dy, sin, ds, dh);
#define NR_LN_BWD(V, I) hipLaunchKernelGGL((ln_bwd_kernel<V, I>), dim3(grid), dim3(256), 0, s, dy, sin, mean, rstd, gamma, ds, dh, dgamma_part, dbeta_part, R, D, dc, dh_h)
    if (vec) {
        if (D <= 512) NR_LN_BWD(4, 2); else if (D <= 1024) NR_LN_BWD(4, 4); else NR_LN_BWD(4, 8);
    } else {
        NR_UNSUPPORTED(D > 64 * 32, "layernorm bwd (unaligned path): D=%d > 2048", D);
        if (D <= 512) NR_LN_BWD(1, 8); else NR_LN_BWD(1, 32);
    }
#undef NR_LN_BWD
    return check_launch("ln_bwd");
}

// 4 consecutive elements at element offset `off` of an f32 tensor, or of a bf16 one (IN_H: `p` is really 16-bit storage)
template <bool IN_H>
__device__ __forceinline__ float4 ld4(const float* p, int64_t off) {
    if (IN_H) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(p) + off);
        return make_float4(__builtin_bit_cast(float, u.x << 16), __builtin_bit_cast(float, u.x & 0xffff0000u),
                           __builtin_bit_cast(float, u.y << 16), __builtin_bit_cast(float, u.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(p + off);
}

// ------------------------------------------------------------------ column sums
// out[n] = sum_r in[r, n]; two deterministic stages through `part` [nchunk, N].
// Block = 16 column-quads (64 columns, 256 contiguous bytes per row) x 16 row lanes; grid =
// (column tiles, row chunks).  Each lane keeps 4 independent float4 accumulators in flight.
template <int VEC, bool IN_H = false>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, int64_t ld, float* __restrict__ out,
                                                     int64_t R, int N, int64_t rows_per_block,
                                                     const float* __restrict__ in_b, float* __restrict__ out_b) {
    __shared__ float red[16][64 + 1];
    if (blockIdx.z == 1) { in = in_b; out = out_b; }          // a second, same-shaped problem in the same launch
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int c0 = blockIdx.x * 64 + cx * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = min(R, r0 + rows_per_block);
    float4 a[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c0 < N) {
        int64_t r = r0 + ry;
        if (VEC == 4) {
            for (; r + 48 < r1; r += 64) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float4 v = ld4<IN_H>(in, (r + 16 * u) * ld + c0);
                    a[u].x += v.x; a[u].y += v.y; a[u].z += v.z; a[u].w += v.w;
                }
            }
            for (; r < r1; r += 16) {
                const float4 v = ld4<IN_H>(in, r * ld + c0);
                a[0].x += v.x; a[0].y += v.y; a[0].z += v.z; a[0].w += v.w;
            }
        } else {
            for (; r < r1; r += 16) {
                const float* q = in + r * ld + c0;
                a[0].x += q[0];
                if (c0 + 1 < N) a[0].y += q[1];
                if (c0 + 2 < N) a[0].z += q[2];
                if (c0 + 3 < N) a[0].w += q[3];
            }
        }
    }
    red[ry][cx * 4 + 0] = (a[0].x + a[1].x) + (a[2].x + a[3].x);
    red[ry][cx * 4 + 1] = (a[0].y + a[1].y) + (a[2].y + a[3].y);
    red[ry][cx * 4 + 2] = (a[0].z + a[1].z) + (a[2].z + a[3].z);
    red[ry][cx * 4 + 3] = (a[0].w + a[1].w) + (a[2].w + a[3].w);
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = blockIdx.x * 64 + threadIdx.x;
        if (c < N) {
            float t = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += red[k][threadIdx.x];
            out[(int64_t)blockIdx.y * N + c] = t;
        }
    }
}

int colsum_chunks(int64_t R) { return (int)std::max<int64_t>(1, std::min<int64_t>(cdiv(R, 128), 512)); }

int colsum(const float* in, int64_t ld, int64_t R, int N, float* out, float* part, hipStream_t s) {
    if (N == 0) return NR_OK;
    ProfScope prof(FAM_MISC, s, (double)R * N * 4);
    const int ctiles = cdiv(N, 64);
    const bool vec = (N % 4 == 0) && (ld % 4 == 0) && aligned16(in);
    const int nchunk = part ? colsum_chunks(R) : 1;
#define NR_COLSUM(V, GRID, IN, LD, OUT, ROWS, RPB) hipLaunchKernelGGL(colsum_kernel<V>, GRID, dim3(256), 0, s, IN, LD, OUT, ROWS, N, RPB, (const float*)nullptr, (float*)nullptr)
    if (nchunk == 1) {
        if (vec) NR_COLSUM(4, dim3(ctiles, 1), in, ld, out, R, std::max<int64_t>(R, 1));
        else     NR_COLSUM(1, dim3(ctiles, 1), in, ld, out, R, std::max<int64_t>(R, 1));
        return check_launch("colsum");
    }
    const int64_t rpb = cdiv(R, nchunk);
    if (vec) NR_COLSUM(4, dim3(ctiles, nchunk), in, ld, part, R, rpb);
    else     NR_COLSUM(1, dim3(ctiles, nchunk), in, ld, part, R, rpb);
    if (N % 4 == 0) NR_COLSUM(4, dim3(ctiles, 1), part, (int64_t)N, out, (int64_t)nchunk, (int64_t)nchunk);
    else            NR_COLSUM(1, dim3(ctiles, 1), part, (int64_t)N, out, (int64_t)nchunk, (int64_t)nchunk);
#undef NR_COLSUM
    return check_launch("colsum");
}

// ------------------------------------------------------------------ packed title tokens
// The synthetic / MIND title tensors pad every title to T tokens; about half of the rows of a training step are
// padding.  A padded position of a title with at least one real token never reaches anything that is kept (as a key
// its softmax weight is exp(x - 1e5 - max) = 0 exactly; as a query its row is dropped by the pooling softmax the same
// way; its gradients carry those exact zeros), so the news tower can run on the real rows only.  Titles with NO real
// token (news 0: the padding of candidate lists and histories) keep all T rows: with every key masked the shifted
// scores still rank.  Plan: len'[n] = len[n] > 0 ? len[n] : T, offs = exclusive scan; info[0] = total rows,
// info[1] = 1 iff every mask row is prefix-shaped (mask[n][j] == (j < len[n])), the only form the packing is exact for.
// title_keep (optional, [N] 0/1): titles the caller proves dead (history slots the user encoder masks out: their news
// vectors never reach the loss and receive exactly zero gradient) get NO rows at all.
// Stage 1, one wave per title: len'[n] into offs[n] (scanned in place by stage 2) and the prefix-shape check.
__global__ __launch_bounds__(256) void pack_lens_kernel(const int64_t* __restrict__ mask, const int64_t* __restrict__ title_keep,
                                                        int64_t N, int T, int* __restrict__ offs, int* __restrict__ info) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (n >= N) return;
    const int64_t* m = mask + n * T;
    int len = 0;
    for (int j0 = 0; j0 < T; j0 += 64) {
        const int j = j0 + lane;
        len += __popcll(__ballot(j < T && m[j] != 0));
    }
    bool bad = false;
    for (int j0 = 0; j0 < T; j0 += 64) {
        const int j = j0 + lane;
        if (j < T && ((m[j] != 0) != (j < len))) bad = true;
    }
    if (__ballot(bad) != 0ull && lane == 0) info[1] = 1;            // violation marker (benign race: every writer stores 1)
    if (lane == 0) offs[n] = (title_keep && title_keep[n] == 0) ? 0 : (len == 0 ? T : len);
}

// Stage 2, one workgroup: exclusive scan of offs[0..N) in place, offs[N] = info[0] = total
__global__ __launch_bounds__(1024) void pack_scan_kernel(int64_t N, int* __restrict__ offs, int* __restrict__ info) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < N; base += 1024) {
        const int64_t n = base + tid;
        const int len = n < N ? offs[n] : 0;
        int v = len;                                       // inclusive scan over the 1024 threads: wave scan + 16 wave totals
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(v, o, 64);
            if (lane >= o) v += t;
        }
        if (lane == 63) wsum[w] = v;
        __syncthreads();
        int pre = carry;
        for (int k = 0; k < w; ++k) pre += wsum[k];
        if (n < N) offs[n] = pre + v - len;               // exclusive
        __syncthreads();
        if (tid == 1023) carry = pre + v;
        __syncthreads();
    }
    if (tid == 0) { offs[N] = carry; info[0] = carry; info[1] = info[1] ? 0 : 1; }   // marker -> "all rows prefix-shaped"
}

// one wave per title: packed ids and the packed-row -> original-row map
__global__ __launch_bounds__(256) void pack_ids_kernel(const int64_t* __restrict__ ids, const int* __restrict__ offs, int64_t N, int T,
                                                       int64_t* __restrict__ ids_packed, int* __restrict__ row_map) {
    const int lane = threadIdx.x & 63;
    const int64_t n = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (n >= N) return;
    const int o = offs[n], len = offs[n + 1] - o;
    for (int j = lane; j < len; j += 64) {
        ids_packed[o + j] = ids[n * T + j];
        row_map[o + j] = (int)(n * T + j);
    }
}

// ---- title tensors of one train step in ONE launch (models/BaseModel.py forward; reference BaseModel.py:178-195 copies the
// candidate and history tensors to the device separately and encodes them in two calls).  Block-major layout: the B*C
// candidate titles first, then the B*L history titles, so the news vectors of either group are one contiguous slice.
//   ids / msk [B*C + B*L, T]; keep [B*C + B*L] (1 for candidates; history slot (b,l): keep_mode 1 -> his_mask[b,l] != 0,
//   keep_mode 2 -> l < sum_l' his_mask[b,l'], keep_mode 0 -> 1)
__global__ __launch_bounds__(256) void assemble_titles_kernel(const int64_t* __restrict__ cdd_ids, const int64_t* __restrict__ his_ids,
                                                              const int64_t* __restrict__ cdd_msk, const int64_t* __restrict__ his_msk,
                                                              const int64_t* __restrict__ his_mask, int keep_mode, int B, int C, int L, int T,
                                                              int64_t* __restrict__ ids, int64_t* __restrict__ msk, int64_t* __restrict__ keep) {
    const int64_t nc = (int64_t)B * C * T, total = nc + (int64_t)B * L * T;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const bool his = i >= nc;
        const int64_t j = his ? i - nc : i;
        ids[i] = his ? his_ids[j] : cdd_ids[j];
        msk[i] = his ? his_msk[j] : cdd_msk[j];
        if (j % T == 0) {
            const int64_t title = i / T;
            int64_t k = 1;
            if (his && keep_mode != 0) {
                const int64_t slot = j / T;                   // b * L + l
                if (keep_mode == 1) k = his_mask[slot] != 0;
                else {
                    const int64_t b = slot / L, l = slot - b * L;
                    int64_t len = 0;
                    for (int t = 0; t < L; ++t) len += his_mask[b * L + t];
                    k = l < len;
                }
            }
            keep[title] = k;
        }
    }
}
int assemble_titles(const int64_t* cdd_ids, const int64_t* his_ids, const int64_t* cdd_msk, const int64_t* his_msk, const int64_t* his_mask,
                    int keep_mode, int B, int C, int L, int T, int64_t* ids, int64_t* msk, int64_t* keep, hipStream_t s) {
    const int64_t total = (int64_t)B * (C + L) * T;
    if (total == 0) return NR_OK;
    const int grid = (int)std::min<int64_t>(cdiv(total, 256), 2048);
    hipLaunchKernelGGL(assemble_titles_kernel, dim3(grid), dim3(256), 0, s, cdd_ids, his_ids, cdd_msk, his_msk, his_mask, keep_mode, B, C, L, T, ids, msk, keep);
    return check_launch("assemble_titles");
}

int pack_plan(const int64_t* mask, const int64_t* title_keep, int64_t N, int T, int* offs, int* info, hipStream_t s) {
    // info[1]: stage 1 raises it on a mask row that is not prefix-shaped, stage 2 turns it into the "all rows are" flag
    if (hipMemsetAsync(info, 0, 2 * sizeof(int), s) != hipSuccess) { set_error("pack_plan: memset failed"); return NR_ERR_LAUNCH; }
    if (N > 0) hipLaunchKernelGGL(pack_lens_kernel, dim3(cdiv(N, WAVES_PER_BLOCK)), dim3(256), 0, s, mask, title_keep, N, T, offs, info);
    hipLaunchKernelGGL(pack_scan_kernel, dim3(1), dim3(1024), 0, s, N, offs, info);
    return check_launch("pack_plan");
}
int pack_ids(const int64_t* ids, const int* offs, int64_t N, int T, int64_t* ids_packed, int* row_map, hipStream_t s) {
    if (N == 0) return NR_OK;
    hipLaunchKernelGGL(pack_ids_kernel, dim3(cdiv(N, WAVES_PER_BLOCK)), dim3(256), 0, s, ids, offs, N, T, ids_packed, row_map);
    return check_launch("pack_ids");
}

// column sums of a bf16 [R,N] matrix (N % 4 == 0, 8-byte aligned rows) -> f32 out[N]; part as for colsum
int colsum_h(const unsigned short* in, int64_t ld, int64_t R, int N, float* out, float* part, hipStream_t s) {
    if (N == 0) return NR_OK;
    NR_CHECK_ARG(N % 4 == 0 && ld % 4 == 0 && (reinterpret_cast<uintptr_t>(in) & 7) == 0 && part, "colsum_h: N, ld must be multiples of 4, rows 8-byte aligned");
    ProfScope prof(FAM_MISC, s, (double)R * N * 2);
    const int ctiles = cdiv(N, 64);
    const int nchunk = colsum_chunks(R);
    const int64_t rpb = cdiv(R, nchunk);
    hipLaunchKernelGGL((colsum_kernel<4, true>), dim3(ctiles, nchunk), dim3(256), 0, s, reinterpret_cast<const float*>(in), ld, part, R, N, rpb,
                       (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL((colsum_kernel<4, false>), dim3(ctiles, 1), dim3(256), 0, s, (const float*)part, (int64_t)N, out, (int64_t)nchunk, N, (int64_t)nchunk,
                       (const float*)nullptr, (float*)nullptr);
    return check_launch("colsum_h");
}

// two same-shaped column sums in one launch pair (dgamma and dbeta partials of the LayerNorm backward):
// out_a[n] = sum_r a[r,n], out_b[n] = sum_r b[r,n]; part: 2 * colsum_chunks(R) * N floats
int colsum2(const float* a, const float* b, int64_t R, int N, float* out_a, float* out_b, float* part, hipStream_t s) {
    if (N == 0) return NR_OK;
    ProfScope prof(FAM_MISC, s, 2.0 * R * N * 4);
    const int ctiles = cdiv(N, 64);
    const bool vec = (N % 4 == 0) && aligned16(a, b);
    const int nchunk = colsum_chunks(R);
    float* part_b = part + (size_t)nchunk * N;
#define NR_COLSUM2(V, GRID, INA, INB, OUTA, OUTB, ROWS, RPB) hipLaunchKernelGGL(colsum_kernel<V>, GRID, dim3(256), 0, s, INA, (int64_t)N, OUTA, ROWS, N, RPB, INB, OUTB)
    if (nchunk == 1) {
        if (vec) NR_COLSUM2(4, dim3(ctiles, 1, 2), a, b, out_a, out_b, R, std::max<int64_t>(R, 1));
        else     NR_COLSUM2(1, dim3(ctiles, 1, 2), a, b, out_a, out_b, R, std::max<int64_t>(R, 1));
        return check_launch("colsum2");
    }
    const int64_t rpb = cdiv(R, nchunk);
    if (vec) NR_COLSUM2(4, dim3(ctiles, nchunk, 2), a, b, part, part_b, R, rpb);
    else     NR_COLSUM2(1, dim3(ctiles, nchunk, 2), a, b, part, part_b, R, rpb);
    if (N % 4 == 0) NR_COLSUM2(4, dim3(ctiles, 1, 2), (const float*)part, (const float*)part_b, out_a, out_b, (int64_t)nchunk, (int64_t)nchunk);
    else            NR_COLSUM2(1, dim3(ctiles, 1, 2), (const float*)part, (const float*)part_b, out_a, out_b, (int64_t)nchunk, (int64_t)nchunk);
#undef NR_COLSUM2
    return check_launch("colsum2");
}

// ------------------------------------------------------------------ bf16 shadow copies (the bf16 build, gemm_h.hip)
// y[i] = bf16(x[i]); n % 8 == 0 and 16-byte aligned pointers take the vector path
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ y, int64_t n, int vec) {
    if (vec) {
        const int64_t n8 = n >> 3;
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) {
            const float4 a = *reinterpret_cast<const float4*>(x + 8 * i), b = *reinterpret_cast<const float4*>(x + 8 * i + 4);
            uint4 o;
            o.x = pack2_bf16(a.x, a.y); o.y = pack2_bf16(a.z, a.w); o.z = pack2_bf16(b.x, b.y); o.w = pack2_bf16(b.z, b.w);
            *reinterpret_cast<uint4*>(y + 8 * i) = o;
        }
    } else {
        for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
            y[i] = (unsigned short)(pack2_bf16(x[i], 0.f) & 0xffffu);
    }
}
int cast_bf16(const float* x, unsigned short* y, int64_t n, hipStream_t s) {
    if (n == 0) return NR_OK;
    const int vec = (n % 8 == 0) && aligned16(x, y) ? 1 : 0;
    ProfScope prof(FAM_MISC, s, (double)n * 6);
    hipLaunchKernelGGL(cast_bf16_kernel, dim3((int)std::min<int64_t>(cdiv(vec ? n / 8 : n, 256), 16384)), dim3(256), 0, s, x, y, n, vec);
    return check_launch("cast_bf16");
}

// dst[c][r] = bf16(src[r][c]) for r < R, 0 for R <= r < Rp: the transposed, reduction-padded copies that turn
// dX = dY.W and dW = dY^T.X into the NT form of gemm_h.  64 x 64 tiles through LDS, both sides in full 128-B segments.
template <bool SRC_H>
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const void* __restrict__ src_, int64_t ld, unsigned short* __restrict__ dst,
                                                             int64_t R, int64_t Cn, int64_t Rp, unsigned short* __restrict__ straight) {
    __shared__ float t[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64, c0 = (int64_t)blockIdx.y * 64;
    const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rr = ty + 16 * i;
        const int64_t r = r0 + rr, c = c0 + 4 * tx;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < R) {
            if (c + 3 < Cn) {
                if (SRC_H) {
                    const uint2 u = *reinterpret_cast<const uint2*>(static_cast<const unsigned short*>(src_) + r * ld + c);
                    v[0] = __builtin_bit_cast(float, u.x << 16); v[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
                    v[2] = __builtin_bit_cast(float, u.y << 16); v[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
                } else {
                    const float4 f = *reinterpret_cast<const float4*>(static_cast<const float*>(src_) + r * ld + c);
                    v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
                    // (the plain bf16 copy [R, Cn] in the same pass: the forward's weight cast and the backward's weight transpose in one read)
                    if (straight) *reinterpret_cast<uint2*>(straight + r * Cn + c) = make_uint2(pack2_bf16(v[0], v[1]), pack2_bf16(v[2], v[3]));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c + e < Cn) v[e] = SRC_H ? __builtin_bit_cast(float, (uint32_t)static_cast<const unsigned short*>(src_)[r * ld + c + e] << 16)
                                                 : static_cast<const float*>(src_)[r * ld + c + e];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) t[rr][4 * tx + e] = v[e];
    }
    __syncthreads();
    const int c = threadIdx.x >> 2, rq = threadIdx.x & 3;
    if (c0 + c < Cn) {
        uint32_t o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = pack2_bf16(t[16 * rq + 2 * k][c], t[16 * rq + 2 * k + 1][c]);
        uint4* d = reinterpret_cast<uint4*>(dst + (c0 + c) * Rp + r0 + 16 * rq);
        d[0] = make_uint4(o[0], o[1], o[2], o[3]);
        d[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}
// src [R, Cn] (row pitch ld; f32, or bf16 when src_is_bf16; ld % 4 == 0) -> dst [Cn, Rp], Rp % 64 == 0, Rp >= R
int transpose_bf16(const void* src, int src_is_bf16, int64_t ld, int64_t R, int64_t Cn, unsigned short* dst, int64_t Rp, hipStream_t s, unsigned short* straight) {
    if (Cn == 0 || Rp == 0) return NR_OK;
    NR_CHECK_ARG(Rp % 64 == 0 && Rp >= R && ld % 4 == 0, "transpose_bf16: Rp=%lld must be a multiple of 64 >= R=%lld, ld %% 4 == 0", (long long)Rp, (long long)R);
    NR_CHECK_ARG(!straight || (!src_is_bf16 && Cn % 4 == 0), "transpose_bf16: the plain copy needs an f32 source with Cn %% 4 == 0");
    ProfScope prof(FAM_MISC, s, (double)R * Cn * (src_is_bf16 ? 4 : 6));
    const dim3 grid((unsigned)(Rp / 64), (unsigned)cdiv(Cn, 64));
    if (src_is_bf16) hipLaunchKernelGGL(transpose_bf16_kernel<true>, grid, dim3(256), 0, s, src, ld, dst, R, Cn, Rp, (unsigned short*)nullptr);
    else             hipLaunchKernelGGL(transpose_bf16_kernel<false>, grid, dim3(256), 0, s, src, ld, dst, R, Cn, Rp, straight);
    return check_launch("transpose_bf16");
}

// ------------------------------------------------------------------ small elementwise helpers
// conv weight re-layouts: w[D,E,3] <-> wf[D, 3E] (wf[d][kk*E+e] = w[d][e][kk]); and the
// flipped/transposed one for dX: wb[e][kk*D+d] = w[d][e][2-kk]
__global__ void conv_pack_kernel(const float* __restrict__ w, float* __restrict__ wf, float* __restrict__ wb, int D, int E) {
    const int64_t total = (int64_t)D * E * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 3);
        const int e = (int)((i / 3) % E);
        const int d = (int)(i / (3 * (int64_t)E));
        const float v = w[i];
        if (wf) wf[(int64_t)d * 3 * E + kk * E + e] = v;
        if (wb) wb[(int64_t)e * 3 * D + (2 - kk) * D + d] = v;
    }
}
__global__ void conv_unpack_grad_kernel(const float* __restrict__ gwf, float* __restrict__ gw, int D, int E) {
    const int64_t total = (int64_t)D * E * 3;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % 3);
        const int e = (int)((i / 3) % E);
        const int d = (int)(i / (3 * (int64_t)E));
        gw[i] = gwf[(int64_t)d * 3 * E + kk * E + e];
    }
}
int conv_pack(const float* w, float* wf, float* wb, int D, int E, hipStream_t s) {
    const int64_t total = (int64_t)D * E * 3;
    hipLaunchKernelGGL(conv_pack_kernel, dim3((int)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, s, w, wf, wb, D, E);
    return check_launch("conv_pack");
}
int conv_unpack_grad(const float* gwf, float* gw, int D, int E, hipStream_t s) {
    const int64_t total = (int64_t)D * E * 3;
    hipLaunchKernelGGL(conv_unpack_grad_kernel, dim3((int)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, s, gwf, gw, D, E);
    return check_launch("conv_unpack_grad");
}

// out[i] = a[i] + b[i]
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}
int add_f32(const float* a, const float* b, float* o, int64_t n, hipStream_t s) {
    if (n == 0) return NR_OK;
    hipLaunchKernelGGL(add_kernel, dim3((int)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0, s, a, b, o, n);
    return check_launch("add");
}

// out[i] = y[i] > 0 ? dy[i] : 0    (ReLU backward from the saved output)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = y[i] > 0.f ? dy[i] : 0.f;
}
int relu_bwd_f32(const float* dy, const float* y, float* o, int64_t n, hipStream_t s) {
    if (n == 0) return NR_OK;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0, s, dy, y, o, n);
    return check_launch("relu_bwd");
}

// out[i] = dy[i] * gelu'(z[i])    (erf-GELU backward from the saved pre-activation)
__global__ void gelu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, float* __restrict__ o, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        o[i] = dy[i] * gelu_grad_f(z[i]);
}
int gelu_bwd_f32(const float* dy, const float* z, float* o, int64_t n, hipStream_t s) {
    if (n == 0) return NR_OK;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3((int)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0, s, dy, z, o, n);
    return check_launch("gelu_bwd");
}

// out[i] = keep(i) ? x[i] / (1-p) : 0   (a dropout site that is not followed by a LayerNorm: BERT's embedding
// dropout; the same call with the same (seed, site) is its backward)
// With dc.row_map (packed rows of width D, D % 4 == 0) the keep-bits are those of the ORIGINAL element row_map[r]*D + c.
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ o, int64_t n, DropoutCfg dc, int D) {
    const uint32_t thr = drop_threshold(dc.p);
    const float scale = 1.0f / (1.0f - dc.p);
    const int64_t nq = (n + 3) >> 2;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq; q += (int64_t)gridDim.x * blockDim.x) {
        uint64_t ctr = (uint64_t)q;
        if (dc.row_map) {
            const int64_t r = (q * 4) / D, c = q * 4 - r * D;
            ctr = (uint64_t)((int64_t)dc.row_map[r] * D + c) >> 2;
        }
        const Philox4 rn = philox4x32_10(dc.seed, ctr, dc.site);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t i = q * 4 + e;
            if (i < n) o[i] = philox_pick(rn, e) >= thr ? x[i] * scale : 0.f;
        }
    }
}
int dropout_f32(const float* x, float* o, int64_t n, DropoutCfg dc, hipStream_t s, int D) {
    if (n == 0) return NR_OK;
    NR_CHECK_ARG(!dc.row_map || (D > 0 && D % 4 == 0), "dropout (packed rows): the row width must be a multiple of 4");
    ProfScope prof(FAM_MISC, s, (double)n * 8);
    hipLaunchKernelGGL(dropout_kernel, dim3((int)std::min<int64_t>(cdiv(cdiv(n, 4), 256), 8192)), dim3(256), 0, s, x, o, n, dc, D);
    return check_launch("dropout");
}

// dst[row_map[r], :] = src[r, :] over a zeroed dst [Rfull, D]: packed rows back to the padded layout (the position-row
// gradient of the BERT embedding stage sums over titles per position)
__global__ __launch_bounds__(256) void unpack_rows_kernel(const float* __restrict__ src, const int* __restrict__ row_map, int64_t R, int D,
                                                          float* __restrict__ dst) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (r >= R) return;
    const float* sp = src + r * D;
    float* dp = dst + (int64_t)row_map[r] * D;
    for (int c = lane * 4; c < D; c += 256) *reinterpret_cast<float4*>(dp + c) = *reinterpret_cast<const float4*>(sp + c);
}
int unpack_rows(const float* src, const int* row_map, int64_t R, int D, int64_t Rfull, float* dst, hipStream_t s) {
    NR_CHECK_ARG(D % 4 == 0 && aligned16(src, dst), "unpack_rows: D %% 4 == 0 and 16-byte aligned tensors");
    if (hipMemsetAsync(dst, 0, (size_t)Rfull * D * sizeof(float), s) != hipSuccess) { set_error("unpack_rows: memset failed"); return NR_ERR_LAUNCH; }
    if (R == 0) return NR_OK;
    ProfScope prof(FAM_MISC, s, (double)(R + Rfull) * D * 4);
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(cdiv(R, WAVES_PER_BLOCK)), dim3(256), 0, s, src, row_map, R, D, dst);
    return check_launch("unpack_rows");
}

// out[r, :] = x[r, :] + b[:]
__global__ void add_bias_rows_kernel(const float* __restrict__ x, const float* __restrict__ b, float* __restrict__ o, int64_t R, int D) {
    const int64_t n = R * D;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) o[i] = x[i] + b[i % D];
}
int add_bias_rows_f32(const float* x, const float* b, float* o, int64_t R, int D, hipStream_t s) {
    if (R == 0) return NR_OK;
    hipLaunchKernelGGL(add_bias_rows_kernel, dim3((int)std::min<int64_t>(cdiv(R * D, 256), 8192)), dim3(256), 0, s, x, b, o, R, D);
    return check_launch("add_bias_rows");
}

// mean of a short vector (loss over the batch); one block, fixed order
__global__ __launch_bounds__(256) void mean_kernel(const float* __restrict__ v, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float a = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) a += v[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}
int mean_f32(const float* v, int n, float* out, hipStream_t s) {
    hipLaunchKernelGGL(mean_kernel, dim3(1), dim3(256), 0, s, v, n, out);
    return check_launch("mean");
}

// ------------------------------------------------------------------ Adam (torch.optim.Adam defaults, dense)
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, float omb1, float b2, float omb2,
                                                   float eps, float step_size, float bc2_sqrt) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + omb1 * (gi - m[i]);                   // exp_avg.lerp_(grad, 1-beta1)
        const float vi = v[i] * b2 + omb2 * gi * gi;                  // mul_(beta2).addcmul_(g, g, 1-beta2)
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - step_size * (mi / denom);                      // addcdiv_(exp_avg, denom, -step_size)
        m[i] = mi; v[i] = vi;
    }
}
int adam_f32(float* p, const float* g, float* m, float* v, int64_t n, double lr, double b1, double b2, double eps, int64_t step,
             hipStream_t s) {
    if (n == 0) return NR_OK;
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    // torch forms 1-beta in python doubles and only then rounds to fp32 (1.0f - 0.999f is 4.7e-5 off)
    const float omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2);
    ProfScope prof(FAM_MISC, s, (double)n * 4 * 7);
    hipLaunchKernelGGL(adam_kernel, dim3((int)std::min<int64_t>(cdiv(n, 256), 8192)), dim3(256), 0, s, p, g, m, v, n, omb1, (float)b2,
                       omb2, (float)eps, step_size, bc2_sqrt);
    return check_launch("adam");
}

// All parameters of a model in ONE launch (39 per-tensor launches cost 4x the bandwidth time of the update itself).
// The tensor table travels by value in the kernel arguments; block b updates one 4096-element chunk of the tensor
// whose block range contains b.
struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    int64_t n[ADAM_MAX_TENSORS];
    int blk0[ADAM_MAX_TENSORS + 1];
    int count;
};
constexpr int ADAM_CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_multi_kernel(const AdamTable t, float omb1, float b2, float omb2, float eps,
                                                         float step_size, float bc2_sqrt) {
    int lo = 0, hi = t.count - 1;                  // last tensor with blk0 <= blockIdx.x
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    float* __restrict__ p = t.p[lo];
    const float* __restrict__ g = t.g[lo];
    float* __restrict__ m = t.m[lo];
    float* __restrict__ v = t.v[lo];
    const int64_t n = t.n[lo];
    const int64_t base = (int64_t)(blockIdx.x - t.blk0[lo]) * ADAM_CHUNK;
    const int64_t end = min(n, base + ADAM_CHUNK);
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
    auto upd = [&](float& pi, float gi, float& mi_, float& vi_) {
        const float mi = mi_ + omb1 * (gi - mi_);
        const float vi = vi_ * b2 + omb2 * gi * gi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi = pi - step_size * (mi / denom);
        mi_ = mi; vi_ = vi;
    };
    if (vec && base + ADAM_CHUNK <= n) {
#pragma unroll
        for (int u = 0; u < ADAM_CHUNK / 1024; ++u) {
            const int64_t i = base + (u * 256 + threadIdx.x) * 4;
            float4 pp = *reinterpret_cast<float4*>(p + i), mm = *reinterpret_cast<float4*>(m + i), vv = *reinterpret_cast<float4*>(v + i);
            const float4 gg = *reinterpret_cast<const float4*>(g + i);
            upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
            *reinterpret_cast<float4*>(p + i) = pp; *reinterpret_cast<float4*>(m + i) = mm; *reinterpret_cast<float4*>(v + i) = vv;
        }
    } else {
        for (int64_t i = base + threadIdx.x; i < end; i += 256) upd(p[i], g[i], m[i], v[i]);
    }
}

int adam_multi_f32(int count, float* const* p, const float* const* g, float* const* m, float* const* v, const int64_t* n,
                   double lr, double b1, double b2, double eps, int64_t step, hipStream_t s) {
    const double bc1 = 1.0 - pow(b1, (double)step);
    const double bc2 = 1.0 - pow(b2, (double)step);
    const float step_size = (float)(lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float omb1 = (float)(1.0 - b1), omb2 = (float)(1.0 - b2);
    int i = 0;
    while (i < count) {
        AdamTable t;
        t.count = 0;
        int blocks = 0;
        double elems = 0;
        for (; i < count && t.count < ADAM_MAX_TENSORS; ++i) {
            if (n[i] == 0) continue;
            const int c = t.count++;
            t.p[c] = p[i]; t.g[c] = g[i]; t.m[c] = m[i]; t.v[c] = v[i]; t.n[c] = n[i];
            t.blk0[c] = blocks;
            blocks += cdiv(n[i], ADAM_CHUNK);
            elems += (double)n[i];
        }
        if (t.count == 0) continue;
        t.blk0[t.count] = blocks;
        ProfScope prof(FAM_MISC, s, elems * 4 * 7);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, s, t, omb1, (float)b2, omb2, (float)eps, step_size, bc2_sqrt);
        if (int rc = check_launch("adam_multi")) return rc;
    }
    return NR_OK;
}

}  // namespace nr
