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

// ---- the whole forward recurrence in ONE persistent launch -----------------------------------
// The per-step pair (GEMM + gate kernel) is latency-bound: 100 launches of ~7.5 us for L = 50.  Here workgroup w owns
// GRU_U = 10 hidden units for all L steps.  Its slice of W_hh (30 rows: r, z, n of the 10 units = 30 of the 32 columns
// of one MFMA tile) stays in registers for the whole kernel, split over 12 waves by k; per step a wave multiplies the
// [B, k-slice] part of h_{t-1} (read from HBM/L2, where the other workgroups published it) into a 32x32 partial, the
// partials meet in LDS, 320 threads apply the gates and publish h_t, and all workgroups meet at a grid barrier
// (agent-scope release -> atomic counter -> acquire; cooperative launch, so co-residency is checked by the runtime,
// and the spin is bounded so that a lost workgroup ends the kernel instead of hanging the GPU).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int GRU_U = 10;
constexpr int GRU_WAVES = 12;
constexpr int GRU_NS = 48;           // MFMAs per wave and batch tile at most: k-slice <= 96, H <= 1152
constexpr int GRU_PLD = 33;

// The state crosses workgroups (and XCDs, each with its own L2) through agent-scope relaxed atomics: stores write
// through to the coherence point, loads bypass the non-coherent caches.  A release/acquire fence pair instead costs an
// L2 write-back + invalidate per wave and step (measured: 24 us per step, slower than the two launches it replaces).
__device__ __forceinline__ void st_agent(float* p, float v) {
    __hip_atomic_store(reinterpret_cast<unsigned*>(p), __builtin_bit_cast(unsigned, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_agent(const float* p) {
    return __builtin_bit_cast(float, __hip_atomic_load(reinterpret_cast<const unsigned*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void ld_agent2(const float* p, float& a, float& b) {          // p 8-byte aligned
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    a = __builtin_bit_cast(float, (unsigned)(v & 0xffffffffu));
    b = __builtin_bit_cast(float, (unsigned)(v >> 32));
}

// Grid barrier with a bounded spin.  A workgroup that gives up (a lost or never-scheduled peer) raises counter[1]
// (read back by nr_gru_status) AND remembers it in *timed_out (LDS): before the kernel ends, that workgroup overwrites
// the outputs it owns with NaN, so the failure reaches the loss / the gradients instead of passing as stale numbers.
__device__ __forceinline__ void gru_grid_sync(unsigned* counter, unsigned target, unsigned spin_limit, int* timed_out) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's state stores have reached the coherence point
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (spins++ >= spin_limit) {                   // default 2^22 polls ~ seconds: give up rather than hang
                __hip_atomic_store(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *timed_out = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __syncthreads();
}

static unsigned g_gru_spin_limit = 1u << 22;
void gru_set_spin_limit(unsigned polls) { g_gru_spin_limit = polls; }

__global__ __launch_bounds__(64 * GRU_WAVES) void gru_seq_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ w_hh,
                                                                    const float* __restrict__ b_hh, const int* __restrict__ lens,
                                                                    float* hall, float* __restrict__ gates, int B, int L, int H,
                                                                    int KS, unsigned* counter, unsigned spin_limit) {
    __shared__ float part[GRU_WAVES][32 * GRU_PLD];
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * GRU_U;
    const int kq0 = q * KS, nu = KS / 8;

    // this lane's column of the weight tile: gate = n / 10, unit = n % 10 (columns 30, 31 are padding)
    float w[GRU_NS];
    {
        const int gate = l31 / GRU_U, unit = l31 - gate * GRU_U, j = j0 + unit;
        const bool wvalid = l31 < 3 * GRU_U && j < H;
        const float* wrow = w_hh + (int64_t)(wvalid ? gate * H + j : 0) * H;
#pragma unroll
        for (int u = 0; u < GRU_NS / 4; ++u) {
            const int k = kq0 + 8 * u + 4 * hh;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (wvalid && u < nu && k < H) v = *reinterpret_cast<const f32x4*>(wrow + k);
            w[4 * u] = v[0]; w[4 * u + 1] = v[1]; w[4 * u + 2] = v[2]; w[4 * u + 3] = v[3];
        }
    }
    const int btiles = (B + 31) / 32;
    for (int t = 0; t < L; ++t) {
        const float* hp = hall + (size_t)t * B * H;
        float* hn = hall + (size_t)(t + 1) * B * H;
        for (int bt = 0; bt < btiles; ++bt) {
            const int b = bt * 32 + l31;
            const float* hrow = hp + (int64_t)(b < B ? b : 0) * H;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < GRU_NS / 4; ++u) {
                if (u < nu) {                              // uniform
                    const int k = kq0 + 8 * u + 4 * hh;
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
                    if (b < B && k < H) {
                        float a0, a1, a2, a3;
                        ld_agent2(hrow + k, a0, a1);
                        ld_agent2(hrow + k + 2, a2, a3);
                        a[0] = a0; a[1] = a1; a[2] = a2; a[3] = a3;
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], w[4 * u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], w[4 * u + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], w[4 * u + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], w[4 * u + 3], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) part[q][((r & 3) + 8 * (r >> 2) + 4 * hh) * GRU_PLD + l31] = acc[r];
            __syncthreads();
            if (tid < 32 * GRU_U) {
                const int bb = tid / GRU_U, unit = tid - bb * GRU_U;
                const int b2 = bt * 32 + bb, j = j0 + unit;
                if (b2 < B && j < H) {
                    float g3[3];
#pragma unroll
                    for (int g = 0; g < 3; ++g) {
                        float v = 0.f;
#pragma unroll
                        for (int qq = 0; qq < GRU_WAVES; ++qq) v += part[qq][bb * GRU_PLD + g * GRU_U + unit];
                        g3[g] = v + b_hh[g * H + j];
                    }
                    const float hprev = ld_agent(hp + (int64_t)b2 * H + j);
                    float hnext = hprev;
                    if (t < lens[b2]) {
                        const float* gib = gi + ((int64_t)b2 * L + t) * 3 * H;
                        const float r = sigmoid_f(gib[j] + g3[0]);
                        const float z = sigmoid_f(gib[H + j] + g3[1]);
                        const float n = tanhf(gib[2 * H + j] + r * g3[2]);
                        hnext = (1.0f - z) * n + z * hprev;
                        if (gates) {
                            float* sv = gates + ((size_t)t * B + b2) * 4 * H;
                            sv[j] = r; sv[H + j] = z; sv[2 * H + j] = n; sv[3 * H + j] = g3[2];
                        }
                    }
                    st_agent(hn + (int64_t)b2 * H + j, hnext);
                }
            }
            __syncthreads();
        }
        if (t + 1 < L) gru_grid_sync(counter, (unsigned)(t + 1) * gridDim.x, spin_limit, &timed_out);
    }
    if (timed_out) {                                       // block-uniform (read after the barrier's __syncthreads)
        float* hl = hall + (size_t)L * B * H;
        for (int e = tid; e < B * GRU_U; e += 64 * GRU_WAVES) {
            const int b2 = e / GRU_U, j = j0 + e % GRU_U;
            if (j < H) st_agent(hl + (int64_t)b2 * H + j, __builtin_nanf(""));
        }
    }
}

bool gru_seq_fwd_supported(int B, int H) { return H % 4 == 0 && H <= 12 * 2 * GRU_NS && B >= 1 && B <= 1024; }

// counter: 2 unsigned in device memory (arrivals, timeout flag), zeroed here
int gru_seq_fwd(const float* gi, const float* w_hh, const float* b_hh, const int* lens, float* hall, float* gates, int B, int L, int H,
                unsigned* counter, hipStream_t s) {
    if (hipMemsetAsync(counter, 0, 2 * sizeof(unsigned), s) != hipSuccess) { set_error("gru_seq_fwd: memset failed"); return NR_ERR_LAUNCH; }
    int KS = ((H + GRU_WAVES - 1) / GRU_WAVES + 7) / 8 * 8;
    unsigned spin_limit = g_gru_spin_limit;
    const int grid = cdiv(H, GRU_U);
    void* args[] = {(void*)&gi, (void*)&w_hh, (void*)&b_hh, (void*)&lens, (void*)&hall, (void*)&gates, (void*)&B, (void*)&L, (void*)&H,
                    (void*)&KS, (void*)&counter, (void*)&spin_limit};
    ProfScope prof(FAM_GEMM, s, 2.0 * B * H * 3.0 * H * L);
    if (hipLaunchCooperativeKernel((const void*)gru_seq_fwd_kernel, dim3(grid), dim3(64 * GRU_WAVES), args, 0, s) != hipSuccess) {
        (void)hipGetLastError();
        return NR_ERR_UNSUPPORTED;                         // the caller falls back to the per-step path
    }
    return check_launch("gru_seq_fwd");
}

// ---- the backward recurrence in one persistent launch: same scheme, transposed roles.  Workgroup w owns 32 hidden
// units (one MFMA tile of dh columns); its [3H, 32] slice of W_hh stays in registers, split over 12 waves by k.
// Per step: gate backward for the owned units (dh_t is local) -> publish dGH_t[:, owned r|z|n columns] and write
// dGI_t -> grid barrier -> every wave multiplies its k-slice of the full dGH_t into a 32x32 partial -> LDS ->
// dh_{t-1} = dh_t * z + dGH_t W_hh for the owned units.  Each step publishes into its own slab of dgh [L, B, 3H]
// (kept anyway for dW_hh), so one barrier per step is enough.
constexpr int GRU_NSB = 48;          // MFMAs per wave and batch tile at most: k-slice <= 96, 3H <= 1152

__global__ __launch_bounds__(64 * GRU_WAVES) void gru_seq_bwd_kernel(const float* __restrict__ dh_out, const float* __restrict__ gates,
                                                                    const float* __restrict__ hall, const int* __restrict__ lens,
                                                                    const float* __restrict__ w_hh, float* __restrict__ dgi,
                                                                    float* dgh, int B, int L, int H, int KS, unsigned* counter,
                                                                    unsigned spin_limit) {
    extern __shared__ float gsm[];
    __shared__ int timed_out;
    if (threadIdx.x == 0) timed_out = 0;
    float* part = gsm;                                   // [12][32 * 33]
    float* dhs = gsm + GRU_WAVES * 32 * GRU_PLD;         // [ceil(B/32)*32][33]: dh_t, then dh_t * z, then dh_{t-1}
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int j0 = blockIdx.x * 32;
    const int K3 = 3 * H;
    const int kq0 = q * KS, nu = KS / 8;
    const int btiles = (B + 31) / 32;

    float w[GRU_NSB];
    {
        const int j = j0 + l31;
#pragma unroll
        for (int u = 0; u < GRU_NSB / 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = kq0 + 8 * u + 4 * hh + i;
                w[4 * u + i] = (u < nu && k < K3 && j < H) ? w_hh[(int64_t)k * H + j] : 0.f;
            }
        }
    }
    for (int e = tid; e < btiles * 32 * 32; e += 64 * GRU_WAVES) {
        const int b = e >> 5, n = e & 31, j = j0 + n;
        dhs[b * GRU_PLD + n] = (b < B && j < H) ? dh_out[(int64_t)b * H + j] : 0.f;
    }
    __syncthreads();

    for (int t = L - 1; t >= 0; --t) {
        float* dgh_t = dgh + (size_t)t * B * K3;
        // gate backward for the owned units
        for (int e = tid; e < btiles * 32 * 32; e += 64 * GRU_WAVES) {
            const int b = e >> 5, n = e & 31, j = j0 + n;
            if (b >= B || j >= H) continue;
            const float g = dhs[b * GRU_PLD + n];
            float* gib = dgi + ((int64_t)b * L + t) * K3;
            float* ghb = dgh_t + (int64_t)b * K3;
            if (t >= lens[b]) {
                gib[j] = 0.f; gib[H + j] = 0.f; gib[2 * H + j] = 0.f;
                st_agent(ghb + j, 0.f); st_agent(ghb + H + j, 0.f); st_agent(ghb + 2 * H + j, 0.f);
                continue;                                 // dh passes through a frozen step unchanged
            }
            const float* sv = gates + ((size_t)t * B + b) * 4 * H;
            const float r = sv[j], z = sv[H + j], nn = sv[2 * H + j], ghn = sv[3 * H + j];
            const float hp = hall[((size_t)t * B + b) * H + j];
            const float dn_pre = g * (1.0f - z) * (1.0f - nn * nn);
            const float dz_pre = g * (hp - nn) * z * (1.0f - z);
            const float dr_pre = dn_pre * ghn * r * (1.0f - r);
            gib[j] = dr_pre; gib[H + j] = dz_pre; gib[2 * H + j] = dn_pre;
            st_agent(ghb + j, dr_pre); st_agent(ghb + H + j, dz_pre); st_agent(ghb + 2 * H + j, dn_pre * r);
            dhs[b * GRU_PLD + n] = g * z;
        }
        gru_grid_sync(counter, (unsigned)(L - t) * gridDim.x, spin_limit, &timed_out);
        // dh_{t-1}[:, owned] = dh_t * z + dGH_t . W_hh[:, owned]
        for (int bt = 0; bt < btiles; ++bt) {
            const int b = bt * 32 + l31;
            const float* grow = dgh_t + (int64_t)(b < B ? b : 0) * K3;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int u = 0; u < GRU_NSB / 4; ++u) {
                if (u < nu) {                              // uniform
                    const int k = kq0 + 8 * u + 4 * hh;
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    if (b < B && k < K3) {                 // 3H % 4 == 0: the four k stay in range together
                        ld_agent2(grow + k, a0, a1);
                        ld_agent2(grow + k + 2, a2, a3);
                    }
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w[4 * u], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w[4 * u + 1], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, w[4 * u + 2], acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, w[4 * u + 3], acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) part[q * 32 * GRU_PLD + ((r & 3) + 8 * (r >> 2) + 4 * hh) * GRU_PLD + l31] = acc[r];
            __syncthreads();
            for (int e = tid; e < 32 * 32; e += 64 * GRU_WAVES) {
                const int bb = e >> 5, n = e & 31;
                float v = dhs[(bt * 32 + bb) * GRU_PLD + n];
#pragma unroll
                for (int qq = 0; qq < GRU_WAVES; ++qq) v += part[qq * 32 * GRU_PLD + bb * GRU_PLD + n];
                dhs[(bt * 32 + bb) * GRU_PLD + n] = v;
            }
            __syncthreads();
        }
    }
    if (timed_out) {                                       // poison this workgroup's columns of dGI (-> dW_ih, dx become NaN)
        for (int e = tid; e < B * 32; e += 64 * GRU_WAVES) {
            const int b = e >> 5, j = j0 + (e & 31);
            if (j < H) dgi[((int64_t)b * L) * K3 + j] = __builtin_nanf("");
        }
    }
}

// H <= 384 only: with one workgroup per 32 hidden units the wider models have too few workgroups for the MFMA work of a
// step (H = 768: 24 workgroups, 26 us per step against 22 us for the gate kernel + skinny GEMM pair; H = 300: 16 us against 21)
bool gru_seq_bwd_supported(int B, int H) { return H % 4 == 0 && H <= 384 && B >= 1 && B <= 256; }

int gru_seq_bwd(const float* dh_out, const float* gates, const float* hall, const int* lens, const float* w_hh, float* dgi, float* dgh,
                int B, int L, int H, unsigned* counter, hipStream_t s) {
    if (hipMemsetAsync(counter, 0, 2 * sizeof(unsigned), s) != hipSuccess) { set_error("gru_seq_bwd: memset failed"); return NR_ERR_LAUNCH; }
    int KS = ((3 * H + GRU_WAVES - 1) / GRU_WAVES + 7) / 8 * 8;
    unsigned spin_limit = g_gru_spin_limit;
    const int grid = cdiv(H, 32);
    const size_t lds = (size_t)(GRU_WAVES * 32 * GRU_PLD + cdiv(B, 32) * 32 * GRU_PLD) * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gru_seq_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    void* args[] = {(void*)&dh_out, (void*)&gates, (void*)&hall, (void*)&lens, (void*)&w_hh, (void*)&dgi, (void*)&dgh, (void*)&B, (void*)&L,
                    (void*)&H, (void*)&KS, (void*)&counter, (void*)&spin_limit};
    ProfScope prof(FAM_GEMM, s, 2.0 * B * H * 3.0 * H * L);
    if (hipLaunchCooperativeKernel((const void*)gru_seq_bwd_kernel, dim3(grid), dim3(64 * GRU_WAVES), args, lds, s) != hipSuccess) {
        (void)hipGetLastError();
        return NR_ERR_UNSUPPORTED;                         // the caller falls back to the per-step path
    }
    return check_launch("gru_seq_bwd");
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
