// Shared device/host helpers for the newsrec HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define NR_OK 0
#define NR_ERR_ARG 1
#define NR_ERR_UNSUPPORTED 2
#define NR_ERR_LAUNCH 3
#define NR_ERR_WORKSPACE 4

namespace nr {

void set_error(const char* fmt, ...);

#define NR_CHECK_ARG(cond, ...)                 \
    do {                                        \
        if (!(cond)) {                          \
            nr::set_error(__VA_ARGS__);         \
            return NR_ERR_ARG;                  \
        }                                       \
    } while (0)

#define NR_UNSUPPORTED(cond, ...)               \
    do {                                        \
        if (cond) {                             \
            nr::set_error(__VA_ARGS__);         \
            return NR_ERR_UNSUPPORTED;          \
        }                                       \
    } while (0)

int check_launch(const char* what);

// optional per-family kernel timing (bench.py roofline leg); see prof.hip
enum Family { FAM_GEMM = 0, FAM_GATHER, FAM_ATTN, FAM_LN, FAM_POOL, FAM_SCORE, FAM_MISC, FAM_SCATTER, FAM_GEMM_SPLIT, FAM_HBM_SMALL, FAM_COUNT };
void prof_begin(int family, hipStream_t s, double work);
void prof_end(int family, hipStream_t s);

struct ProfScope {
    int fam; hipStream_t s;
    // launches of the HBM-bound families that move under 16 MB (the 32-sequence user-tower calls, the 32-row scoring) are
    // launch-bound, not bandwidth-bound: they are booked apart so that a family's GB/s is that of its real launches
    ProfScope(int f, hipStream_t st, double work) : fam(f), s(st) {
        if ((f == FAM_LN || f == FAM_POOL || f == FAM_GATHER || f == FAM_SCORE) && work < 16e6) fam = FAM_HBM_SMALL;
        prof_begin(fam, st, work);
    }
    ~ProfScope() { prof_end(fam, s); }
};

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// ---------------------------------------------------------------- device side
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// a / b for many a and one b.  With rb = fl(1 / b) (correctly rounded), q = a * rb and one residual correction the result
// is the correctly rounded quotient (Markstein), i.e. the same float as a / b, unless b's significand is all ones or the
// quotient leaves the normal range; 3 instructions instead of the ~12 of an IEEE division.
__device__ __forceinline__ float div_by(float a, float b, float rb) {
    const float q = a * rb;
    return fmaf(fmaf(-q, b, a), rb, q);
}

// erf for the GELU epilogues: x P(x^2) / Q(x^2) on [-4, 4] (the degree-(6,4) rational of Eigen's / XLA's float erf; its
// approximation error is 6.5e-8, evaluated in f32 with fma chains 3.7e-7 absolute at worst, 5.7e-8 rms), clamped outside,
// where erf is +-1 to f32.  17 issue slots (11 fma, v_rcp_f32, v_med3_f32, 3 mul) against ~40 for the library erff with
// its two branches: in the FFN1 epilogue of the split-operand GEMM erff was a third of the kernel (133 us against 100 us
// for the same product without GELU).  gelu() built on it is as close to the float64 GELU as torch's own f32 gelu
// (|error| / max(|z|, 1): 2.1e-7 here, 3.7e-7 torch CPU; checked on 4e6 points in [-6, 6]).
__device__ __forceinline__ float erf_fast(float x) {
    x = __builtin_amdgcn_fmed3f(x, -4.f, 4.f);
    const float t = x * x;
    float p = -2.72614225801306e-10f;
    p = fmaf(p, t, 2.77068142495902e-08f);
    p = fmaf(p, t, -2.10102402082508e-06f);
    p = fmaf(p, t, -5.69250639462346e-05f);
    p = fmaf(p, t, -7.34990630326855e-04f);
    p = fmaf(p, t, -2.95459980854025e-03f);
    p = fmaf(p, t, -1.60960333262415e-02f);
    float q = -1.45660718464996e-05f;
    q = fmaf(q, t, -2.13374055278905e-04f);
    q = fmaf(q, t, -1.68282697438203e-03f);
    q = fmaf(q, t, -7.37332916720468e-03f);
    q = fmaf(q, t, -1.42647390514189e-02f);
    return (p * x) * __builtin_amdgcn_rcpf(q);
}
// erf GELU and its derivative (torch.nn.functional.gelu default)
__device__ __forceinline__ float gelu_f(float z) {
    const float h = 0.5f * z;
    return fmaf(h, erf_fast(z * 0.70710678118654752440f), h);
}
__device__ __forceinline__ float gelu_grad_f(float z) {
    const float cdf = fmaf(0.5f, erf_fast(z * 0.70710678118654752440f), 0.5f);
    const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * z * z);     // exp(-z^2 / 2)
    return fmaf(z, pdf, cdf);
}

// Philox4x32-10, counter-based: one call yields 4 uniform u32.  The dropout
// stream of this library is DEFINED as philox(key=(seed_lo,seed_hi),
// counter=(idx_lo, idx_hi, site, 0)); keep = u32 >= p * 2^32.
struct Philox4 { uint32_t x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(uint64_t seed, uint64_t idx, uint32_t site) {
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    uint32_t c0 = (uint32_t)idx, c1 = (uint32_t)(idx >> 32), c2 = site, c3 = 0u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one full 32x32 -> 64 product per multiplier (v_mad_u64_u32) instead of a low and a high multiply:
        // 32-bit integer multiplies issue at quarter rate and are most of this function's cycles
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return Philox4{c0, c1, c2, c3};
}
__device__ __forceinline__ uint32_t philox_pick(const Philox4& r, int i) {
    return i == 0 ? r.x : (i == 1 ? r.y : (i == 2 ? r.z : r.w));
}
// keep-threshold for probability p of dropping
__host__ __device__ __forceinline__ uint32_t drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (uint32_t)t;
}

}  // namespace nr
