// extern "C" surface of libnewsrec_hip.so (declared in include/newsrec_hip.h) + error/profiler state.
#include "common.h"
#include "gemm.h"
#include "rowops.h"
#include "seqops.h"
#include "../../include/newsrec_hip.h"

#include <stdarg.h>
#include <cstdlib>
#include <atomic>
#include <mutex>
#include <vector>

namespace nr {

static thread_local char g_err[512] = "";
// weight-gradient products of a layer backward on a second stream (see tfm_layer_bwd); nr_set_side_stream
static std::atomic<int> g_side_on{(getenv("NR_SIDE_STREAM") && atoi(getenv("NR_SIDE_STREAM")) == 0) ? 0 : 1};

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return NR_ERR_LAUNCH;
    }
    return NR_OK;
}

// ---------------------------------------------------------------- profiler
struct ProfFam {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;   // reusable event pairs
    size_t used = 0;
    double work = 0;
};
static bool g_prof_on = false;
static ProfFam g_fam[FAM_COUNT];
static std::mutex g_prof_mu;

void prof_begin(int family, hipStream_t s, double work) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfFam& f = g_fam[family];
    if (f.used == f.pool.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        f.pool.emplace_back(a, b);
    }
    f.work += work;
    (void)hipEventRecord(f.pool[f.used].first, s);
}
void prof_end(int family, hipStream_t s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfFam& f = g_fam[family];
    if (f.used >= f.pool.size()) return;
    (void)hipEventRecord(f.pool[f.used].second, s);
    f.used++;
}

static inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }
static inline DropoutCfg DC(float p, uint64_t seed, uint32_t site) {
    DropoutCfg d; d.p = p; d.seed = seed; d.site = site; return d;
}
// attention: mask_mode 0 = additive -1e5 (reference TFMLayer), 1 = excluded keys + safe softmax (HF BertModel / SDPA)
static inline DropoutCfg DCA(float p, uint64_t seed, uint32_t site, int mask_mode) {
    DropoutCfg d = DC(p, seed, site); d.mask_fill = mask_mode ? -INFINITY : -1e5f; return d;
}
static inline DropoutCfg DCP(DropoutCfg d, const int* offs, const int* row_map) { d.offs = offs; d.row_map = row_map; return d; }
static inline float* bump(char*& cur, size_t floats) {
    float* p = reinterpret_cast<float*>(cur);
    cur += ((floats * sizeof(float) + 255) / 256) * 256;
    return p;
}
static inline size_t pad256(size_t floats) { return ((floats * sizeof(float) + 255) / 256) * 256; }

// y = act(x w^T + b)
// skinny problems (a GRU step: 32 sequences) leave most CUs idle unless the reduction is split: with a
// workspace of skinny_ws_bytes() the GEMM runs split-K + the deterministic slab reduce
static size_t skinny_ws_bytes(int64_t M, int64_t N, int64_t K) {
    return gemm_workspace_bytes((int)M, (int)N, choose_ksplit((int)M, (int)N, (int)K)) + 256;
}
static void maybe_split(GemmParams& p, float* ws, size_t ws_bytes) {
    if (!ws) return;
    const int ks = choose_ksplit(p.M, p.N, p.K);
    if (ks > 1 && gemm_workspace_bytes(p.M, p.N, ks) <= ws_bytes) p.ksplit = ks;
}
static int linear_fwd(const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t K, int64_t N, int epi,
                      float* y, float* z, hipStream_t s, float* ws = nullptr, size_t ws_bytes = 0) {
    GemmParams p;
    p.A = x; p.lda = ldx; p.B = w; p.ldb = K; p.C = y; p.ldc = N;
    p.M = (int)R; p.N = (int)N; p.K = (int)K; p.bias = b; p.epi = epi; p.out2 = z; p.ldout2 = N;
    maybe_split(p, ws, ws_bytes);
    return gemm_f32(p, true, true, ws, ws_bytes, s);
}
// dx = dy w  (+ epilogue)
static int linear_bwd_input(const float* dy, int64_t lddy, const float* w, int64_t R, int64_t N, int64_t K, int epi,
                            const float* aux, const float* aux2, float* dx, hipStream_t s, float* ws = nullptr, size_t ws_bytes = 0) {
    GemmParams p;
    p.A = dy; p.lda = lddy; p.B = w; p.ldb = K; p.C = dx; p.ldc = K;
    p.M = (int)R; p.N = (int)K; p.K = (int)N; p.epi = epi; p.aux = aux; p.ldaux = K; p.aux2 = aux2; p.ldaux2 = K;
    maybe_split(p, ws, ws_bytes);
    return gemm_f32(p, true, false, ws, ws_bytes, s);
}
static size_t bwd_weight_ws(int64_t R, int64_t N, int64_t K, bool with_bias) {
    const int ks = choose_ksplit((int)N, (int)K, (int)R);
    size_t b = pad256(gemm_workspace_bytes((int)N, (int)K, ks) / sizeof(float));
    if (with_bias) b += pad256(std::max((size_t)colsum_chunks(R), (size_t)ks) * N);
    return b;
}
// dw = dy^T x ; db = colsum(dy)
static int linear_bwd_weight(const float* dy, int64_t lddy, const float* x, int64_t ldx, int64_t R, int64_t N, int64_t K,
                             float* dw, float* db, char* ws, size_t ws_bytes, int conv_T, int conv_E, hipStream_t s) {
    const size_t need = bwd_weight_ws(R, N, K, db != nullptr);
    if (ws_bytes < need || (!ws && need)) {
        set_error("linear_bwd_weight: workspace %zu < %zu bytes", ws_bytes, need);
        return NR_ERR_WORKSPACE;
    }
    char* cur = ws;
    GemmParams p;
    p.A = dy; p.lda = lddy; p.B = x; p.ldb = ldx; p.C = dw; p.ldc = K;
    p.M = (int)N; p.N = (int)K; p.K = (int)R;
    p.ksplit = choose_ksplit((int)N, (int)K, (int)R);
    if (conv_T > 0) { p.conv_on = 2; p.conv_T = conv_T; p.conv_E = conv_E; }
    const size_t slab_floats = gemm_workspace_bytes((int)N, (int)K, p.ksplit) / sizeof(float);
    float* slabs = bump(cur, slab_floats);
    float* part = db ? bump(cur, std::max((size_t)colsum_chunks(R), (size_t)p.ksplit) * N) : nullptr;
    // the bias gradient db = colsum(dy) rides on the GEMM (dy tiles are in LDS anyway) unless the slow path runs
    const bool fused = db && gemm_fast_path(p, false, false);
    if (fused) { p.csum = db; p.csum_part = part; }
    int rc = gemm_f32(p, false, false, slabs, slab_floats * sizeof(float), s);
    if (rc != NR_OK) return rc;
    if (db && !fused) rc = colsum(dy, lddy, R, (int)N, db, part, s);
    return rc;
}

}  // namespace nr

using namespace nr;

extern "C" {

int nr_version(void) { return 1; }
int nr_set_gemm_compute(int mode) {
    NR_CHECK_ARG(mode == 0 || mode == 1 || mode == 2, "nr_set_gemm_compute: mode must be 0 (fp32 instruction), 1 (bf16 operands) or 2 (fp32 from three bf16 pieces)");
    nr::set_gemm_compute(mode);
    return NR_OK;
}

int nr_set_side_stream(int on) {
    return g_side_on.exchange(on ? 1 : 0);
}
int nr_get_gemm_compute(void) { return nr::get_gemm_compute(); }
const char* nr_last_error(void) { return g_err; }

int nr_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    for (auto& f : g_fam) { f.used = 0; f.work = 0; }
    return NR_OK;
}

int nr_prof_read(int family, double* ms, int64_t* launches, double* work) {
    NR_CHECK_ARG(family >= 0 && family < FAM_COUNT, "nr_prof_read: family %d out of range", family);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfFam& f = g_fam[family];
    double total = 0;
    for (size_t i = 0; i < f.used; ++i) {
        if (hipEventSynchronize(f.pool[i].second) != hipSuccess) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, f.pool[i].first, f.pool[i].second) == hipSuccess) total += t;
    }
    if (ms) *ms = total;
    if (launches) *launches = (int64_t)f.used;
    if (work) *work = f.work;
    f.used = 0; f.work = 0;
    return NR_OK;
}

int nr_index_status(int* out_flag, void* stream) { return index_status(out_flag, S(stream)); }

int nr_gather_rows_f32(const float* table, int64_t V, int64_t E, const int64_t* ids, int64_t n, float* out, int* oob_flag, void* stream) {
    NR_CHECK_ARG(n >= 0 && V > 0 && E > 0, "gather_rows: bad sizes n=%lld V=%lld E=%lld", (long long)n, (long long)V, (long long)E);
    NR_CHECK_ARG(n == 0 || (table && ids && out), "gather_rows: null pointer");
    return gather_rows(table, V, (int)E, ids, n, out, oob_flag, S(stream));
}

int nr_scatter_add_rows_f32(const float* grad_rows, const int64_t* ids, int64_t n, int64_t E, int64_t V, int64_t padding_idx, float* grad_table, void* stream) {
    NR_CHECK_ARG(n >= 0 && V > 0 && E > 0, "scatter_add_rows: bad sizes");
    NR_CHECK_ARG(n == 0 || (grad_rows && ids && grad_table), "scatter_add_rows: null pointer");
    return scatter_add_rows(grad_rows, ids, n, (int)E, V, padding_idx, grad_table, S(stream));
}

int nr_linear_fwd_f32(const float* x, const float* w, const float* b, int64_t R, int64_t K, int64_t N, int act, float* y, float* z, void* stream) {
    NR_CHECK_ARG(R >= 0 && K > 0 && N > 0, "linear_fwd: bad sizes");
    NR_CHECK_ARG(act >= 0 && act <= 3, "linear_fwd: act %d not in 0..3", act);
    NR_CHECK_ARG(R == 0 || (x && w && y), "linear_fwd: null pointer");
    return linear_fwd(x, K, w, b, R, K, N, act, y, z, S(stream));
}

int nr_linear_bwd_input_f32(const float* dy, const float* w, int64_t R, int64_t N, int64_t K, int mode, const float* aux, const float* aux2, float* dx, void* stream) {
    NR_CHECK_ARG(R >= 0 && K > 0 && N > 0, "linear_bwd_input: bad sizes");
    NR_CHECK_ARG(R == 0 || (dy && w && dx), "linear_bwd_input: null pointer");
    int epi;
    switch (mode) {
        case 0: epi = EPI_NONE; break;
        case 1: epi = EPI_ADD_AUX; NR_CHECK_ARG(aux, "linear_bwd_input: mode 1 needs aux"); break;
        case 2: epi = EPI_MUL_GELU_GRAD; NR_CHECK_ARG(aux, "linear_bwd_input: mode 2 needs aux"); break;
        case 3: epi = EPI_ADD_AUX_RELUMASK; NR_CHECK_ARG(aux && aux2, "linear_bwd_input: mode 3 needs aux, aux2"); break;
        case 4: epi = EPI_RELUMASK; NR_CHECK_ARG(aux2, "linear_bwd_input: mode 4 needs aux2"); break;
        default: set_error("linear_bwd_input: mode %d not in 0..4", mode); return NR_ERR_ARG;
    }
    return linear_bwd_input(dy, N, w, R, N, K, epi, aux, aux2, dx, S(stream));
}

size_t nr_linear_bwd_weight_ws_bytes(int64_t R, int64_t N, int64_t K) { return bwd_weight_ws(R, N, K, true); }

int nr_linear_bwd_weight_f32(const float* dy, const float* x, int64_t R, int64_t N, int64_t K, float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(R >= 0 && K > 0 && N > 0, "linear_bwd_weight: bad sizes");
    NR_CHECK_ARG(dy && x && dw, "linear_bwd_weight: null pointer");
    return linear_bwd_weight(dy, N, x, K, R, N, K, dw, db, (char*)ws, ws_bytes, 0, 0, S(stream));
}

int nr_conv1d_k3_pack(const float* w, int64_t D, int64_t E, float* wpack, void* stream) {
    NR_CHECK_ARG(w && wpack && D > 0 && E > 0, "conv_pack: bad argument");
    return conv_pack(w, wpack, wpack + D * 3 * E, (int)D, (int)E, S(stream));
}

int nr_conv1d_k3_relu_fwd_f32(const float* x, const float* wpack, const float* b, int64_t M, int64_t T, int64_t E, int64_t D, float* y, void* stream) {
    NR_CHECK_ARG(M >= 0 && T > 0 && E > 0 && D > 0, "conv1d_fwd: bad sizes");
    NR_CHECK_ARG(M == 0 || (x && wpack && y), "conv1d_fwd: null pointer");
    GemmParams p;
    p.A = x - E; p.lda = E; p.B = wpack; p.ldb = 3 * E; p.C = y; p.ldc = D;
    p.M = (int)(M * T); p.N = (int)D; p.K = (int)(3 * E); p.bias = b; p.epi = EPI_RELU;
    p.conv_on = 1; p.conv_T = (int)T; p.conv_E = (int)E;
    return gemm_f32(p, true, true, nullptr, 0, S(stream));
}

size_t nr_conv1d_k3_bwd_ws_bytes(int64_t M, int64_t T, int64_t E, int64_t D) {
    return pad256((size_t)D * 3 * E) + bwd_weight_ws(M * T, D, 3 * E, true);
}

// The four weight-gradient products of the layer backward feed nothing inside the layer: with enough rows to matter they
// run on a second stream of the same device (forked by events after the tensor each one reads is complete, joined before
// the call returns), next to the input-gradient products, the LayerNorm and the attention backward of the main stream —
// the partial last rounds of one kernel and the latency-bound phases of another then hold work of both.
// NR_SIDE_STREAM=0 in the environment or nr_set_side_stream(0) keeps everything on the caller's stream.
struct SideStream {
    hipStream_t s = nullptr; hipEvent_t fork[4] = {}, join = nullptr; bool ok = false;
    std::mutex mu;                          // record + wait of one event as a pair: host threads may share the GPU
    hipError_t edge(hipEvent_t ev, hipStream_t from, hipStream_t to) {
        std::lock_guard<std::mutex> lk(mu);
        const hipError_t e = hipEventRecord(ev, from);
        return e != hipSuccess ? e : hipStreamWaitEvent(to, ev, 0);
    }
};
static SideStream* side_stream() {
    static SideStream g_side[64];
    static std::mutex mu;
    if (!g_side_on.load()) return nullptr;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(mu);
    SideStream& ss = g_side[dev];
    if (!ss.s) {
        if (hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) != hipSuccess) { ss.s = nullptr; return nullptr; }
        ss.ok = hipEventCreateWithFlags(&ss.join, hipEventDisableTiming) == hipSuccess;
        for (int i = 0; i < 4; ++i) ss.ok = ss.ok && hipEventCreateWithFlags(&ss.fork[i], hipEventDisableTiming) == hipSuccess;
    }
    return ss.ok ? &ss : nullptr;
}
// One layer backward's use of the second stream.  fork(i): the side stream waits for everything queued on the main
// stream so far; if that edge cannot be made, the products that would have followed it go to the MAIN stream instead
// (stream() hands out the main stream from then on: slower, never unordered).  finish() — also run by the destructor on
// early returns — makes the main stream wait for the side stream; if THAT edge fails the side stream is drained on the
// host and the call reports NR_ERR_LAUNCH, so no caller ever reads a gradient that is still being written.
struct SideUse {
    SideStream* ss; hipStream_t main; bool joined = false, used = false;
    SideUse(SideStream* ss_, hipStream_t main_) : ss(ss_), main(main_) {}
    hipStream_t stream() const { return ss ? ss->s : main; }
    void fork(int i) {
        if (!ss) return;
        if (ss->edge(ss->fork[i], main, ss->s) == hipSuccess) { used = true; return; }
        (void)hipGetLastError();
        if (used) (void)finish();              // earlier products are on the side stream: order the main stream behind them
        ss = nullptr;                           // from here on everything runs on the caller's stream
    }
    int finish() {
        if (joined || !ss || !used) { joined = true; return NR_OK; }
        joined = true;
        if (ss->edge(ss->join, ss->s, main) == hipSuccess) return NR_OK;
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ss->s);
        set_error("side stream: joining the weight-gradient stream failed (drained it on the host instead)");
        return NR_ERR_LAUNCH;
    }
    ~SideUse() { (void)finish(); }
};
static const int64_t SIDE_MIN_ROWS = getenv("NR_SIDE_MIN_ROWS") ? atoll(getenv("NR_SIDE_MIN_ROWS")) : 1024;

int nr_conv1d_k3_bwd_f32(const float* dy, const float* x, const float* wpack, int64_t M, int64_t T, int64_t E, int64_t D, float* dx, float* dw, float* db, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(M >= 0 && T > 0 && E > 0 && D > 0, "conv1d_bwd: bad sizes");
    NR_CHECK_ARG(dy && x && wpack && dw, "conv1d_bwd: null pointer");
    if (ws_bytes < nr_conv1d_k3_bwd_ws_bytes(M, T, E, D) || !ws) { set_error("conv1d_bwd: workspace too small"); return NR_ERR_WORKSPACE; }
    hipStream_t s = S(stream);
    const int64_t R = M * T;
    // the weight gradient (and its unpacking) on the second stream, beside the input gradient
    SideStream* ss = (dx && R >= SIDE_MIN_ROWS) ? side_stream() : nullptr;
    SideUse su(ss, s);
    su.fork(0);
    const hipStream_t sw = su.stream();
    if (dx) {
        GemmParams p;
        p.A = dy - D; p.lda = D; p.B = wpack + D * 3 * E; p.ldb = 3 * D; p.C = dx; p.ldc = E;
        p.M = (int)R; p.N = (int)E; p.K = (int)(3 * D);
        p.conv_on = 1; p.conv_T = (int)T; p.conv_E = (int)D;
        if (int rc = gemm_f32(p, true, true, nullptr, 0, s)) return rc;
    }
    char* cur = (char*)ws;
    float* gwf = bump(cur, (size_t)D * 3 * E);
    const size_t rest = ws_bytes - (size_t)(cur - (char*)ws);
    if (int rc = linear_bwd_weight(dy, D, x - E, E, R, D, 3 * E, gwf, db, cur, rest, (int)T, (int)E, sw)) return rc;
    if (int rc = conv_unpack_grad(gwf, dw, (int)D, (int)E, sw)) return rc;
    return su.finish();
}

int nr_add_layernorm_fwd_f32(const float* h, const float* x, int64_t x_period, const float* gamma, const float* beta, int64_t R, int64_t D, float eps, float p_drop, uint64_t seed, uint32_t site, float* y, float* s_out, float* mean, float* rstd, void* stream) {
    NR_CHECK_ARG(R >= 0 && D > 0 && x_period >= 0, "layernorm_fwd: bad sizes");
    NR_CHECK_ARG(R == 0 || (h && gamma && beta && y), "layernorm_fwd: null pointer");
    NR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "layernorm_fwd: p_drop %f not in [0,1)", p_drop);
    return ln_fwd(h, x, gamma, beta, y, s_out, mean, rstd, R, (int)D, eps, DC(p_drop, seed, site), S(stream), x_period);
}

size_t nr_layernorm_bwd_ws_bytes(int64_t R, int64_t D) {
    return 2 * pad256((size_t)ln_bwd_blocks(R) * D) + pad256((size_t)2 * colsum_chunks(ln_bwd_blocks(R)) * D);
}

// the row pass (ds, dh and per-workgroup partials of dgamma / dbeta in `ws`) ...
static int layernorm_bwd_rows(const float* dy, const float* s_in, const float* mean, const float* rstd, const float* gamma, int64_t R,
                              int D, DropoutCfg dc, float* ds, float* dh, char* ws, hipStream_t s, unsigned short* dh_h = nullptr) {
    char* cur = ws;
    const int nb = ln_bwd_blocks(R);
    float* pg = bump(cur, (size_t)nb * D);
    float* pb = bump(cur, (size_t)nb * D);
    return ln_bwd(dy, s_in, mean, rstd, gamma, ds, dh, pg, pb, R, D, dc, s, dh_h);
}
// ... and the two column sums that finish dgamma / dbeta (they feed nothing else: the layer backward runs them on its
// second stream, which needs a `ws` of its own per LayerNorm)
static int layernorm_bwd_params(int64_t R, int D, float* dgamma, float* dbeta, char* ws, hipStream_t s) {
    char* cur = ws;
    const int nb = ln_bwd_blocks(R);
    float* pg = bump(cur, (size_t)nb * D);
    float* pb = bump(cur, (size_t)nb * D);
    float* part = bump(cur, (size_t)2 * colsum_chunks(nb) * D);
    return colsum2(pg, pb, nb, D, dgamma, dbeta, part, s);       // dgamma and dbeta in one launch pair
}
static int layernorm_bwd(const float* dy, const float* s_in, const float* mean, const float* rstd, const float* gamma, int64_t R,
                         int D, DropoutCfg dc, float* ds, float* dh, float* dgamma, float* dbeta, char* ws, hipStream_t s,
                         unsigned short* dh_h = nullptr) {
    if (int rc = layernorm_bwd_rows(dy, s_in, mean, rstd, gamma, R, D, dc, ds, dh, ws, s, dh_h)) return rc;
    return layernorm_bwd_params(R, D, dgamma, dbeta, ws, s);
}

int nr_add_layernorm_bwd_f32(const float* dy, const float* s_in, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* ds, float* dh, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(R > 0 && D > 0, "layernorm_bwd: bad sizes");
    NR_CHECK_ARG(dy && s_in && mean && rstd && gamma && ds && dgamma && dbeta, "layernorm_bwd: null pointer");
    if (!ws || ws_bytes < nr_layernorm_bwd_ws_bytes(R, D)) { set_error("layernorm_bwd: workspace too small"); return NR_ERR_WORKSPACE; }
    return layernorm_bwd(dy, s_in, mean, rstd, gamma, R, (int)D, DC(p_drop, seed, site), ds, dh, dgamma, dbeta, (char*)ws, S(stream));
}

int nr_add_layernorm_packed_fwd_f32(const float* h, const float* x, int64_t x_period, const int32_t* row_map, const float* gamma, const float* beta, int64_t R, int64_t D, float eps, float p_drop, uint64_t seed, uint32_t site, float* y, float* s_out, float* mean, float* rstd, void* stream) {
    NR_CHECK_ARG(R >= 0 && D > 0 && x_period >= 0, "layernorm_packed_fwd: bad sizes");
    NR_CHECK_ARG(R == 0 || (h && gamma && beta && y && row_map), "layernorm_packed_fwd: null pointer");
    NR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "layernorm_packed_fwd: p_drop %f not in [0,1)", p_drop);
    return ln_fwd(h, x, gamma, beta, y, s_out, mean, rstd, R, (int)D, eps, DCP(DC(p_drop, seed, site), nullptr, row_map), S(stream), x_period);
}

int nr_add_layernorm_packed_bwd_f32(const float* dy, const float* s_in, const float* mean, const float* rstd, const float* gamma, const int32_t* row_map, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* ds, float* dh, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(R > 0 && D > 0, "layernorm_packed_bwd: bad sizes");
    NR_CHECK_ARG(dy && s_in && mean && rstd && gamma && ds && dgamma && dbeta && row_map, "layernorm_packed_bwd: null pointer");
    if (!ws || ws_bytes < nr_layernorm_bwd_ws_bytes(R, D)) { set_error("layernorm_packed_bwd: workspace too small"); return NR_ERR_WORKSPACE; }
    return layernorm_bwd(dy, s_in, mean, rstd, gamma, R, (int)D, DCP(DC(p_drop, seed, site), nullptr, row_map), ds, dh, dgamma, dbeta, (char*)ws, S(stream));
}

int nr_dropout_packed_f32(const float* x, const int32_t* row_map, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* out, void* stream) {
    NR_CHECK_ARG(R >= 0 && D > 0 && (R == 0 || (x && out && row_map)), "dropout_packed: bad arguments");
    NR_CHECK_ARG(p_drop > 0.f && p_drop < 1.f, "dropout_packed: p_drop %f not in (0,1)", p_drop);
    return dropout_f32(x, out, R * D, DCP(DC(p_drop, seed, site), nullptr, row_map), S(stream), (int)D);
}

int nr_unpack_rows_f32(const float* src, const int32_t* row_map, int64_t R, int64_t D, int64_t Rfull, float* dst, void* stream) {
    NR_CHECK_ARG(R >= 0 && D > 0 && Rfull >= 0 && dst && (R == 0 || (src && row_map)), "unpack_rows: bad arguments");
    return unpack_rows(src, row_map, R, (int)D, Rfull, dst, S(stream));
}

// ---- the one entry point that allocates nothing (callers that cannot hipMalloc) ---------------------------------
// The per-block dgamma/dbeta partials and the column-sum partials of the LayerNorm backward live in a file-scope
// device array instead of a caller workspace.  Every GPU holds its own copy at its own address: the address is looked
// up once per device (hipGetSymbolAddress with that device current) and only that pointer is ever handed to kernels.
#define NR_NOALLOC_SCRATCH_BYTES ((size_t)(8u << 20) + (64u << 10))     /* 8 MiB + 64 KiB: D <= 1024 at any R */
static __device__ __align__(256) unsigned char scratch[NR_NOALLOC_SCRATCH_BYTES];
static std::mutex g_scratch_mu;
static void* g_scratch_ptr[64] = {nullptr};                               // per device ordinal

static int noalloc_scratch(char** out) {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { set_error("noalloc: no current HIP device (ordinal %d)", dev); return NR_ERR_LAUNCH; }
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    if (!g_scratch_ptr[dev]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p) { set_error("noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e)); return NR_ERR_LAUNCH; }
        g_scratch_ptr[dev] = p;
    }
    *out = static_cast<char*>(g_scratch_ptr[dev]);
    return NR_OK;
}

size_t nr_noalloc_scratch_bytes(void) { return NR_NOALLOC_SCRATCH_BYTES; }

int nr_add_layernorm_bwd_noalloc_f32(const float* dy, const float* s_in, const float* mean, const float* rstd, const float* gamma, int64_t R, int64_t D, float p_drop, uint64_t seed, uint32_t site, float* ds, float* dh, float* dgamma, float* dbeta, void* stream) {
    NR_CHECK_ARG(R > 0 && D > 0, "layernorm_bwd_noalloc: bad sizes");
    NR_CHECK_ARG(dy && s_in && mean && rstd && gamma && ds && dgamma && dbeta, "layernorm_bwd_noalloc: null pointer");
    const size_t need = nr_layernorm_bwd_ws_bytes(R, D);
    if (need > NR_NOALLOC_SCRATCH_BYTES) {
        set_error("layernorm_bwd_noalloc: R=%lld D=%lld needs %zu bytes of partials, the built-in scratch holds %zu (use nr_add_layernorm_bwd_f32 with a workspace)",
                  (long long)R, (long long)D, need, (size_t)NR_NOALLOC_SCRATCH_BYTES);
        return NR_ERR_WORKSPACE;
    }
    char* ws = nullptr;
    if (int rc = noalloc_scratch(&ws)) return rc;
    // nothing in the scratch has to start at zero: every partial a later kernel reads is written earlier in this call
    return layernorm_bwd(dy, s_in, mean, rstd, gamma, R, (int)D, DC(p_drop, seed, site), ds, dh, dgamma, dbeta, ws, S(stream));
}

int nr_mhsa_fwd_f32(const float* qkv, const int64_t* mask, int64_t N, int64_t Sq, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, float* ctx, float* row_max, float* row_sum, void* stream) {
    NR_CHECK_ARG(N >= 0 && Sq > 0 && H > 0 && dh > 0, "mhsa_fwd: bad sizes");
    NR_CHECK_ARG(N == 0 || (qkv && mask && ctx), "mhsa_fwd: null pointer");
    NR_CHECK_ARG((row_max == nullptr) == (row_sum == nullptr), "mhsa_fwd: row_max/row_sum must both be given or both NULL");
    return mhsa_fwd(qkv, mask, ctx, row_max, row_sum, N, (int)Sq, (int)H, (int)dh, DCA(p_drop, seed, site, mask_mode), S(stream));
}

int nr_mhsa_bwd_f32(const float* qkv, const int64_t* mask, const float* ctx, const float* dctx, const float* row_max, const float* row_sum, int64_t N, int64_t Sq, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, float* dqkv, void* stream) {
    NR_CHECK_ARG(N >= 0 && Sq > 0 && H > 0 && dh > 0, "mhsa_bwd: bad sizes");
    NR_CHECK_ARG(N == 0 || (qkv && mask && ctx && dctx && row_max && row_sum && dqkv), "mhsa_bwd: null pointer");
    return mhsa_bwd(qkv, mask, ctx, dctx, row_max, row_sum, dqkv, N, (int)Sq, (int)H, (int)dh, DCA(p_drop, seed, site, mask_mode), S(stream));
}

// the same core on bf16 tensors (the bf16 build of the BERT-sized layers keeps qkv / ctx and their gradients as bf16 in HBM)
int nr_mhsa_fwd_bf16(const void* qkv, const int64_t* mask, int64_t N, int64_t Sq, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, void* ctx, float* row_max, float* row_sum, void* stream) {
    NR_CHECK_ARG(N >= 0 && Sq > 0 && H > 0 && dh > 0, "mhsa_fwd_bf16: bad sizes");
    NR_CHECK_ARG(N == 0 || (qkv && mask && ctx), "mhsa_fwd_bf16: null pointer");
    NR_CHECK_ARG((row_max == nullptr) == (row_sum == nullptr), "mhsa_fwd_bf16: row_max/row_sum must both be given or both NULL");
    return mhsa_fwd_mfma_h((const unsigned short*)qkv, mask, (unsigned short*)ctx, row_max, row_sum, N, (int)Sq, (int)H, (int)dh, DCA(p_drop, seed, site, mask_mode), S(stream));
}

int nr_mhsa_bwd_bf16(const void* qkv, const int64_t* mask, const void* dctx, const float* row_max, const float* row_sum, int64_t N, int64_t Sq, int64_t H, int64_t dh, float p_drop, uint64_t seed, uint32_t site, int mask_mode, void* dqkv, void* stream) {
    NR_CHECK_ARG(N >= 0 && Sq > 0 && H > 0 && dh > 0, "mhsa_bwd_bf16: bad sizes");
    NR_CHECK_ARG(N == 0 || (qkv && mask && dctx && row_max && row_sum && dqkv), "mhsa_bwd_bf16: null pointer");
    return mhsa_bwd_mfma_h((const unsigned short*)qkv, mask, (const unsigned short*)dctx, row_max, row_sum, (unsigned short*)dqkv, N, (int)Sq, (int)H, (int)dh, DCA(p_drop, seed, site, mask_mode), S(stream));
}

int nr_query_pool_fwd_f32(const float* key, const float* value, const float* table, const int64_t* idx, int64_t table_rows, const float* query, const int64_t* mask, int64_t nseq, int64_t Sq, int64_t D, float* out, float* prob, void* stream) {
    NR_CHECK_ARG(nseq >= 0 && Sq > 0 && D > 0, "query_pool_fwd: bad sizes");
    NR_CHECK_ARG(nseq == 0 || (query && out), "query_pool_fwd: null pointer");
    NR_CHECK_ARG(idx ? (table && table_rows > 0) : (key && value), "query_pool_fwd: need key/value or table+idx");
    return pool_fwd(key, value, table, idx, table_rows, query, mask, out, prob, nseq, (int)Sq, (int)D, S(stream));
}

size_t nr_query_pool_bwd_ws_bytes(int64_t nseq, int64_t D) {
    return pad256((size_t)nseq * D) + pad256((size_t)colsum_chunks(nseq) * D);
}

int nr_query_pool_bwd_f32(const float* key, const float* value, const float* query, const float* prob, const float* dout, int64_t nseq, int64_t Sq, int64_t D, int tanh_key, int key_is_value, float* dkeypre, float* dvalue, float* dquery, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(nseq > 0 && Sq > 0 && D > 0, "query_pool_bwd: bad sizes");
    NR_CHECK_ARG(key && value && query && prob && dout && dvalue && dquery && (key_is_value || dkeypre), "query_pool_bwd: null pointer");
    if (!ws || ws_bytes < nr_query_pool_bwd_ws_bytes(nseq, D)) { set_error("query_pool_bwd: workspace too small"); return NR_ERR_WORKSPACE; }
    char* cur = (char*)ws;
    float* dq_part = bump(cur, (size_t)nseq * D);
    float* part = bump(cur, (size_t)colsum_chunks(nseq) * D);
    hipStream_t s = S(stream);
    if (int rc = pool_bwd(key, value, query, prob, dout, dkeypre, dvalue, dq_part, nseq, (int)Sq, (int)D, tanh_key, key_is_value, s)) return rc;
    return colsum(dq_part, D, nseq, (int)D, dquery, part, s);
}

int nr_query_pool_packed_fwd_f32(const float* key, const float* value, const int32_t* offs, int64_t rows, const float* query, const int64_t* mask, int64_t nseq, int64_t Sq, int64_t D, float* out, float* prob, void* stream) {
    NR_CHECK_ARG(nseq >= 0 && Sq > 0 && D > 0 && rows >= 0 && rows <= nseq * Sq, "query_pool_packed_fwd: bad sizes");
    NR_CHECK_ARG(nseq == 0 || (key && value && offs && query && out), "query_pool_packed_fwd: null pointer");
    return pool_fwd(key, value, nullptr, nullptr, 0, query, mask, out, prob, nseq, (int)Sq, (int)D, S(stream), offs, rows);
}

int nr_query_pool_packed_bwd_f32(const float* key, const float* value, const int32_t* offs, int64_t rows, const float* query, const float* prob, const float* dout, int64_t nseq, int64_t Sq, int64_t D, int tanh_key, float* dkeypre, float* dvalue, float* dquery, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(nseq > 0 && Sq > 0 && D > 0 && rows >= 0 && rows <= nseq * Sq, "query_pool_packed_bwd: bad sizes");
    NR_CHECK_ARG(key && value && offs && query && prob && dout && dvalue && dquery && dkeypre, "query_pool_packed_bwd: null pointer");
    if (!ws || ws_bytes < nr_query_pool_bwd_ws_bytes(nseq, D)) { set_error("query_pool_packed_bwd: workspace too small"); return NR_ERR_WORKSPACE; }
    char* cur = (char*)ws;
    float* dq_part = bump(cur, (size_t)nseq * D);
    float* part = bump(cur, (size_t)colsum_chunks(nseq) * D);
    hipStream_t s = S(stream);
    if (int rc = pool_bwd(key, value, query, prob, dout, dkeypre, dvalue, dq_part, nseq, (int)Sq, (int)D, tanh_key, 0, s, offs, rows)) return rc;
    return colsum(dq_part, D, nseq, (int)D, dquery, part, s);
}

int nr_assemble_titles(const int64_t* cdd_ids, const int64_t* his_ids, const int64_t* cdd_mask, const int64_t* his_attn_mask, const int64_t* his_mask,
                       int keep_mode, int64_t B, int64_t C, int64_t L, int64_t T, int64_t* ids, int64_t* mask, int64_t* keep, void* stream) {
    NR_CHECK_ARG(B >= 0 && C >= 0 && L >= 0 && T > 0 && B * (C + L) * T < (int64_t)1 << 40, "assemble_titles: bad sizes");
    NR_CHECK_ARG(keep_mode >= 0 && keep_mode <= 2, "assemble_titles: keep_mode %d not in 0..2", keep_mode);
    NR_CHECK_ARG(B * (C + L) == 0 || (ids && mask && keep), "assemble_titles: null output");
    NR_CHECK_ARG(B * C == 0 || (cdd_ids && cdd_mask), "assemble_titles: null candidate tensors");
    NR_CHECK_ARG(B * L == 0 || (his_ids && his_attn_mask && (keep_mode == 0 || his_mask)), "assemble_titles: null history tensors");
    return assemble_titles(cdd_ids, his_ids, cdd_mask, his_attn_mask, his_mask, keep_mode, (int)B, (int)C, (int)L, (int)T, ids, mask, keep, S(stream));
}

int nr_query_pool_packed_bwd_split_f32(const float* key, const float* value, const int32_t* offs, int64_t rows, const float* query, const float* prob, const float* dout0, const float* dout1, int64_t split, int64_t nseq, int64_t Sq, int64_t D, int tanh_key, float* dkeypre, float* dvalue, float* dquery, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(nseq > 0 && Sq > 0 && D > 0 && rows >= 0 && rows <= nseq * Sq && split >= 0 && split <= nseq, "query_pool_packed_bwd_split: bad sizes");
    NR_CHECK_ARG(key && value && offs && query && prob && dvalue && dquery && dkeypre && (split == 0 || dout0) && (split == nseq || dout1), "query_pool_packed_bwd_split: null pointer");
    if (!ws || ws_bytes < nr_query_pool_bwd_ws_bytes(nseq, D)) { set_error("query_pool_packed_bwd_split: workspace too small"); return NR_ERR_WORKSPACE; }
    char* cur = (char*)ws;
    float* dq_part = bump(cur, (size_t)nseq * D);
    float* part = bump(cur, (size_t)colsum_chunks(nseq) * D);
    hipStream_t s = S(stream);
    // (split == 0: everything reads dout1; the kernel's test is n >= split)
    if (int rc = pool_bwd(key, value, query, prob, dout0 ? dout0 : dout1, dkeypre, dvalue, dq_part, nseq, (int)Sq, (int)D, tanh_key, 0, s, offs, rows, dout1, split)) return rc;
    return colsum(dq_part, D, nseq, (int)D, dquery, part, s);
}

int nr_pack_plan(const int64_t* mask, const int64_t* title_keep, int64_t N, int64_t T, int32_t* offs, int32_t* info, void* stream) {
    NR_CHECK_ARG(N >= 0 && T > 0 && T <= 1024 && N * T < (int64_t)1 << 31, "pack_plan: bad sizes N=%lld T=%lld", (long long)N, (long long)T);
    NR_CHECK_ARG(offs && info && (N == 0 || mask), "pack_plan: null pointer");
    return pack_plan(mask, title_keep, N, (int)T, offs, info, S(stream));
}

int nr_pack_ids(const int64_t* ids, const int32_t* offs, int64_t N, int64_t T, int64_t* ids_packed, int32_t* row_map, void* stream) {
    NR_CHECK_ARG(N >= 0 && T > 0, "pack_ids: bad sizes");
    NR_CHECK_ARG(N == 0 || (ids && offs && ids_packed && row_map), "pack_ids: null pointer");
    return pack_ids(ids, offs, N, (int)T, ids_packed, row_map, S(stream));
}

int nr_seq_reduce_f32(const float* x, const float* table, const int64_t* idx, int64_t table_rows, int64_t nseq, int64_t Sq, int64_t D, int mean, float* out, void* stream) {
    NR_CHECK_ARG(nseq >= 0 && Sq > 0 && D > 0, "seq_reduce: bad sizes");
    NR_CHECK_ARG(nseq == 0 || out, "seq_reduce: null pointer");
    NR_CHECK_ARG(idx ? (table && table_rows > 0) : (x != nullptr || nseq == 0), "seq_reduce: need x or table+idx");
    return seq_reduce(x, table, idx, table_rows, out, nseq, (int)Sq, (int)D, mean, S(stream));
}

int nr_seq_reduce_bwd_f32(const float* dout, int64_t nseq, int64_t Sq, int64_t D, int mean, float* dx, void* stream) {
    NR_CHECK_ARG(nseq >= 0 && Sq > 0 && D > 0, "seq_reduce_bwd: bad sizes");
    NR_CHECK_ARG(nseq == 0 || (dout && dx), "seq_reduce_bwd: null pointer");
    return seq_broadcast(dout, dx, nseq, (int)Sq, (int)D, mean ? 1.0f / (float)Sq : 1.0f, S(stream));
}

int nr_score_fwd_f32(const float* cdd, const float* table, const int64_t* idx, int64_t table_rows, const float* user, const int64_t* label, int64_t B, int64_t C, int64_t D, float* logits, float* loss_rows, float* loss, void* stream) {
    NR_CHECK_ARG(B >= 0 && C > 0 && D > 0, "score_fwd: bad sizes");
    NR_CHECK_ARG(B == 0 || (user && logits), "score_fwd: null pointer");
    NR_CHECK_ARG(idx ? (table && table_rows > 0) : (cdd != nullptr || B == 0), "score_fwd: need cdd or table+idx");
    NR_CHECK_ARG(!loss || (loss_rows && label && B > 0), "score_fwd: loss needs loss_rows, label and B > 0");
    hipStream_t s = S(stream);
    if (int rc = score_fwd(cdd, table, idx, table_rows, user, label, logits, loss ? loss_rows : nullptr, B, (int)C, (int)D, s)) return rc;
    if (loss) return mean_f32(loss_rows, (int)B, loss, s);
    return NR_OK;
}

int nr_score_ce_bwd_f32(const float* cdd, const float* user, const int64_t* label, const float* logits, const float* grad_out, int64_t B, int64_t C, int64_t D, float* d_cdd, float* d_user, void* stream) {
    NR_CHECK_ARG(B > 0 && C > 0 && D > 0, "score_ce_bwd: bad sizes");
    NR_CHECK_ARG(cdd && user && label && logits && grad_out && d_cdd && d_user, "score_ce_bwd: null pointer");
    return score_bwd(cdd, user, label, logits, grad_out, nullptr, d_cdd, d_user, B, (int)C, (int)D, S(stream));
}

int nr_score_bwd_f32(const float* cdd, const float* user, const float* dlogits, int64_t B, int64_t C, int64_t D, float* d_cdd, float* d_user, void* stream) {
    NR_CHECK_ARG(B > 0 && C > 0 && D > 0, "score_bwd: bad sizes");
    NR_CHECK_ARG(cdd && user && dlogits && d_cdd && d_user, "score_bwd: null pointer");
    return score_bwd(cdd, user, nullptr, nullptr, nullptr, dlogits, d_cdd, d_user, B, (int)C, (int)D, S(stream));
}

int nr_relu_bwd_f32(const float* dy, const float* y, int64_t n, float* out, void* stream) {
    NR_CHECK_ARG(n >= 0 && (n == 0 || (dy && y && out)), "relu_bwd: bad argument");
    return relu_bwd_f32(dy, y, out, n, S(stream));
}

int nr_gelu_bwd_f32(const float* dy, const float* z, int64_t n, float* out, void* stream) {
    NR_CHECK_ARG(n >= 0 && (n == 0 || (dy && z && out)), "gelu_bwd: bad argument");
    return gelu_bwd_f32(dy, z, out, n, S(stream));
}

// ------------------------------------------------------------------------------------ bf16 build: building blocks
int nr_cast_bf16(const float* x, int64_t n, uint16_t* y, void* stream) {
    NR_CHECK_ARG(n >= 0 && (n == 0 || (x && y)), "cast_bf16: bad argument");
    return cast_bf16(x, y, n, S(stream));
}

int nr_transpose_bf16(const void* src, int src_is_bf16, int64_t R, int64_t Cn, int64_t ld, uint16_t* dst, int64_t Rp, void* stream) {
    NR_CHECK_ARG(R >= 0 && Cn >= 0 && (R == 0 || Cn == 0 || (src && dst)), "transpose_bf16: bad argument");
    return transpose_bf16(src, src_is_bf16, ld, R, Cn, dst, Rp, S(stream));
}

size_t nr_gemm_nt_bf16_ws_bytes(int64_t M, int64_t N, int64_t K, int split_k) {
    return split_k ? gemm_h_workspace_bytes((int)M, (int)N, choose_ksplit_h((int)M, (int)N, (int)K)) + 256 : 0;
}

int nr_gemm_nt_bf16(const uint16_t* A, const uint16_t* B, const float* bias, int64_t M, int64_t N, int64_t K, int act, float* C, uint16_t* Ch, uint16_t* zh, int split_k, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_nt_bf16: bad sizes");
    NR_CHECK_ARG(act >= 0 && act <= 3, "gemm_nt_bf16: act %d not in 0..3", act);
    GemmHParams p;
    p.A = A; p.lda = K; p.B = B; p.ldb = K; p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.C = C; p.ldc = N; p.Ch = Ch; p.ldch = N; p.bias = bias; p.epi = act; p.out2h = zh; p.ldout2h = N;
    if (split_k) p.ksplit = choose_ksplit_h(p.M, p.N, p.K);
    return gemm_h(p, (float*)ws, ws_bytes, S(stream));
}

int nr_gemm_bf16(const uint16_t* A, int a_rm, int64_t lda, const uint16_t* B, int b_rm, int64_t ldb, const float* bias, int64_t M, int64_t N, int64_t K, int act, float* C, uint16_t* Ch, uint16_t* zh, int split_k, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(M >= 0 && N > 0 && K > 0, "gemm_bf16: bad sizes");
    NR_CHECK_ARG(act >= 0 && act <= 3, "gemm_bf16: act %d not in 0..3", act);
    NR_CHECK_ARG(lda >= (a_rm ? M : K) && ldb >= (b_rm ? N : K), "gemm_bf16: pitch smaller than the row it strides");
    GemmHParams p;
    p.A = A; p.lda = lda; p.a_rm = a_rm ? 1 : 0; p.B = B; p.ldb = ldb; p.b_rm = b_rm ? 1 : 0; p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.C = C; p.ldc = N; p.Ch = Ch; p.ldch = N; p.bias = bias; p.epi = act; p.out2h = zh; p.ldout2h = N;
    if (split_k) p.ksplit = choose_ksplit_h(p.M, p.N, p.K);
    return gemm_h(p, (float*)ws, ws_bytes, S(stream));
}

int nr_adam_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1, double beta2, double eps, int64_t step, void* stream) {
    NR_CHECK_ARG(n >= 0 && step >= 1, "adam_step: bad sizes");
    NR_CHECK_ARG(n == 0 || (param && grad && exp_avg && exp_avg_sq), "adam_step: null pointer");
    return adam_f32(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, step, S(stream));
}

int nr_dropout_f32(const float* x, int64_t n, float p_drop, uint64_t seed, uint32_t site, float* out, void* stream) {
    NR_CHECK_ARG(n >= 0 && (n == 0 || (x && out)), "dropout: bad arguments");
    NR_CHECK_ARG(p_drop > 0.f && p_drop < 1.f, "dropout: p_drop %f not in (0,1)", p_drop);
    return dropout_f32(x, out, n, DC(p_drop, seed, site), S(stream));
}

int nr_add_bias_rows_f32(const float* x, const float* b, int64_t R, int64_t D, float* out, void* stream) {
    NR_CHECK_ARG(R >= 0 && D > 0 && (R == 0 || (x && b && out)), "add_bias_rows: bad arguments");
    return add_bias_rows_f32(x, b, out, R, (int)D, S(stream));
}

int nr_colsum_f32(const float* x, int64_t R, int64_t N, float* out, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(R >= 0 && N > 0 && (R == 0 || x) && out, "colsum: bad arguments");
    NR_CHECK_ARG(ws && ws_bytes >= nr_colsum_ws_bytes(R, N), "colsum: workspace too small");
    if (R == 0) return hipMemsetAsync(out, 0, N * sizeof(float), S(stream)) == hipSuccess ? NR_OK : NR_ERR_LAUNCH;
    return colsum(x, N, R, (int)N, out, (float*)ws, S(stream));
}
size_t nr_colsum_ws_bytes(int64_t R, int64_t N) { return pad256((size_t)colsum_chunks(R) * N) + 256; }

int nr_adam_multi_f32(int count, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const int64_t* numel, double lr, double beta1, double beta2, double eps, int64_t step, void* stream) {
    NR_CHECK_ARG(count >= 0 && step >= 1, "adam_multi: bad sizes");
    NR_CHECK_ARG(count == 0 || (params && grads && exp_avg && exp_avg_sq && numel), "adam_multi: null table");
    for (int i = 0; i < count; ++i) {
        NR_CHECK_ARG(numel[i] >= 0, "adam_multi: tensor %d has negative size", i);
        NR_CHECK_ARG(numel[i] == 0 || (params[i] && grads[i] && exp_avg[i] && exp_avg_sq[i]), "adam_multi: tensor %d has a null pointer", i);
    }
    return adam_multi_f32(count, params, grads, exp_avg, exp_avg_sq, numel, lr, beta1, beta2, eps, step, S(stream));
}

// ------------------------------------------------------------------------------------ TFM layer
static bool contiguous3(const float* a, const float* b, const float* c, size_t n) { return b == a + n && c == b + n; }

static int copy_d2d(void* dst, const void* src, size_t bytes, hipStream_t s) {
    if (hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) {
        set_error("hipMemcpyAsync failed");
        return NR_ERR_LAUNCH;
    }
    return NR_OK;
}

// one implementation for the dense [N,S,D] layout (offs == NULL, R = N*S) and the packed one (R = offs[N] rows)
static int tfm_layer_fwd(const float* x, const int64_t* mask, int64_t N, int64_t Sq, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w,
                         float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved* sv, float* y,
                         const int* offs, const int* row_map, hipStream_t s) {
    NR_CHECK_ARG(N >= 0 && Sq > 0 && D > 0 && H > 0 && R >= 0, "tfm_layer_fwd: bad sizes");
    NR_CHECK_ARG(w && sv && x && mask && y, "tfm_layer_fwd: null pointer");
    NR_CHECK_ARG(D % H == 0, "tfm_layer_fwd: hidden dim %lld not divisible by %lld heads (the reference's dense layers need all_head_size == hidden_dim)", (long long)D, (long long)H);
    NR_CHECK_ARG(sv->qkv && sv->ctx && sv->s1 && sv->y1 && sv->a && sv->s2, "tfm_layer_fwd: missing activation buffer");
    NR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "tfm_layer_fwd: p_drop %f not in [0,1)", p_drop);
    if (N == 0 || R == 0) return NR_OK;
    const size_t DD = (size_t)D * D;
    const float* wqkv = w->wq;
    const float* bqkv = w->bq;
    if (!contiguous3(w->wq, w->wk, w->wv, DD) || !contiguous3(w->bq, w->bk, w->bv, (size_t)D)) {
        NR_CHECK_ARG(sv->wqkv && sv->bqkv, "tfm_layer_fwd: q/k/v weights are not contiguous and no pack buffer given");
        const float* ws3[3] = {w->wq, w->wk, w->wv};
        const float* bs3[3] = {w->bq, w->bk, w->bv};
        for (int i = 0; i < 3; ++i) {
            if (int rc = copy_d2d(sv->wqkv + i * DD, ws3[i], DD * sizeof(float), s)) return rc;
            if (int rc = copy_d2d(sv->bqkv + i * D, bs3[i], D * sizeof(float), s)) return rc;
        }
        wqkv = sv->wqkv; bqkv = sv->bqkv;
    }
    float* mean1 = sv->stats ? sv->stats : nullptr;
    float* rstd1 = sv->stats ? sv->stats + R : nullptr;
    float* mean2 = sv->stats ? sv->stats + 2 * R : nullptr;
    float* rstd2 = sv->stats ? sv->stats + 3 * R : nullptr;
    if (int rc = linear_fwd(x, D, wqkv, bqkv, R, D, 3 * D, EPI_NONE, sv->qkv, nullptr, s)) return rc;
    if (int rc = mhsa_fwd(sv->qkv, mask, sv->ctx, sv->row_max, sv->row_sum, N, (int)Sq, (int)H, (int)(D / H),
                          DCP(DCA(p_drop, seed, site_base, mask_mode), offs, nullptr), s)) return rc;
    if (int rc = linear_fwd(sv->ctx, D, w->wo, w->bo, R, D, D, EPI_NONE, sv->s1, nullptr, s)) return rc;
    if (int rc = ln_fwd(sv->s1, x, w->ln1_g, w->ln1_b, sv->y1, sv->s1, mean1, rstd1, R, (int)D, 1e-12f,
                        DCP(DC(p_drop, seed, site_base + 1), nullptr, row_map), s)) return rc;
    if (int rc = linear_fwd(sv->y1, D, w->w1, w->b1, R, D, 4 * D, EPI_GELU, sv->a, sv->z, s)) return rc;
    if (int rc = linear_fwd(sv->a, 4 * D, w->w2, w->b2, R, 4 * D, D, EPI_NONE, sv->s2, nullptr, s)) return rc;
    return ln_fwd(sv->s2, sv->y1, w->ln2_g, w->ln2_b, y, sv->s2, mean2, rstd2, R, (int)D, 1e-12f,
                  DCP(DC(p_drop, seed, site_base + 2), nullptr, row_map), s);
}

int nr_tfm_layer_fwd_f32(const float* x, const int64_t* mask, int64_t N, int64_t Sq, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved* sv, float* y, void* stream) {
    return tfm_layer_fwd(x, mask, N, Sq, N * Sq, D, H, w, p_drop, seed, site_base, mask_mode, sv, y, nullptr, nullptr, S(stream));
}

int nr_tfm_layer_packed_fwd_f32(const float* x, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t Sq, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved* sv, float* y, void* stream) {
    NR_CHECK_ARG(offs && row_map, "tfm_layer_packed_fwd: null offsets / row map");
    NR_CHECK_ARG(R <= N * Sq, "tfm_layer_packed_fwd: %lld packed rows exceed N*S = %lld", (long long)R, (long long)(N * Sq));
    // 32 < S <= 64 (a user's history): forward only — nr_tfm_layer_packed_bwd_f32 refuses it, so sv needs no backward buffers
    NR_UNSUPPORTED(!mhsa_mfma_supported((int)Sq, (int)(H > 0 ? D / H : 0)) && !mhsa_mfma2_supported((int)Sq, (int)(H > 0 ? D / H : 0)),
                   "tfm_layer_packed_fwd: packed sequences need S <= 64 and head dim <= 64");
    return tfm_layer_fwd(x, mask, N, Sq, R, D, H, w, p_drop, seed, site_base, mask_mode, sv, y, offs, row_map, S(stream));
}

size_t nr_tfm_layer_bwd_ws_bytes(int64_t N, int64_t Sq, int64_t D) {
    const int64_t R = N * Sq;
    size_t b = 0;
    b += 8 * pad256((size_t)R * D);                 // ds2 dh2 dy1 ds1 dh1 dctx + 2 spare
    b += pad256((size_t)R * 4 * D);                 // dz
    b += pad256((size_t)R * 3 * D);                 // dqkv
    b += pad256((size_t)3 * D * D) + pad256((size_t)3 * D);
    b += 2 * nr_layernorm_bwd_ws_bytes(R, D);       // one per LayerNorm (their parameter sums run on the second stream)
    size_t wmax = bwd_weight_ws(R, D, 4 * D, true);
    wmax = std::max(wmax, bwd_weight_ws(R, 4 * D, D, true));
    wmax = std::max(wmax, bwd_weight_ws(R, 3 * D, D, true));
    wmax = std::max(wmax, bwd_weight_ws(R, D, D, true));
    return b + wmax;
}

static int tfm_layer_bwd(const float* dy, const float* x, const int64_t* mask, int64_t N, int64_t Sq, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved* sv, nr_tfm_grads* g, float* dx, void* ws, size_t ws_bytes, const int* offs, const int* row_map, hipStream_t s) {
    NR_CHECK_ARG(N > 0 && Sq > 0 && D > 0 && H > 0 && D % H == 0 && R > 0, "tfm_layer_bwd: bad sizes");
    NR_CHECK_ARG(dy && x && mask && w && sv && g && dx, "tfm_layer_bwd: null pointer");
    NR_CHECK_ARG(sv->qkv && sv->ctx && sv->s1 && sv->y1 && sv->z && sv->a && sv->s2 && sv->stats && sv->row_max && sv->row_sum, "tfm_layer_bwd: missing saved activation");
    if (!ws || ws_bytes < nr_tfm_layer_bwd_ws_bytes(R, 1, D)) { set_error("tfm_layer_bwd: workspace too small"); return NR_ERR_WORKSPACE; }
    const size_t DD = (size_t)D * D;
    char* cur = (char*)ws;
    float* ds2 = bump(cur, (size_t)R * D);
    float* dh2 = p_drop > 0.f ? bump(cur, (size_t)R * D) : ds2;
    float* dy1 = bump(cur, (size_t)R * D);
    float* ds1 = bump(cur, (size_t)R * D);
    float* dh1 = p_drop > 0.f ? bump(cur, (size_t)R * D) : ds1;
    float* dctx = bump(cur, (size_t)R * D);
    float* dz = bump(cur, (size_t)R * 4 * D);
    float* dqkv = bump(cur, (size_t)R * 3 * D);
    float* dwqkv = bump(cur, 3 * DD);
    float* dbqkv = bump(cur, (size_t)3 * D);
    char* lnws = cur; cur += nr_layernorm_bwd_ws_bytes(R, D);
    char* lnws1 = cur; cur += nr_layernorm_bwd_ws_bytes(R, D);      // LN1's own partials: LN2's are still being summed on the second stream
    char* wws = cur;
    const size_t wws_bytes = ws_bytes - (size_t)(cur - (char*)ws);
    const float* mean1 = sv->stats, *rstd1 = sv->stats + R, *mean2 = sv->stats + 2 * R, *rstd2 = sv->stats + 3 * R;
    const float* wqkv = contiguous3(w->wq, w->wk, w->wv, DD) ? w->wq : sv->wqkv;
    NR_CHECK_ARG(wqkv, "tfm_layer_bwd: q/k/v weights are not contiguous and no packed copy saved");
    SideStream* ss = R >= SIDE_MIN_ROWS ? side_stream() : nullptr;
    SideUse su(ss, s);
    hipStream_t sw = su.stream();                            // stream of the weight-gradient products
    auto fork = [&](int i) { su.fork(i); sw = su.stream(); };

    // LN2 <- dy
    if (int rc = layernorm_bwd_rows(dy, sv->s2, mean2, rstd2, w->ln2_g, R, (int)D, DCP(DC(p_drop, seed, site_base + 2), nullptr, row_map), ds2,
                                    p_drop > 0.f ? dh2 : nullptr, lnws, s)) return rc;
    // FFN2: dW2 = dh2^T a ; dz = (dh2 W2) * gelu'(z)
    fork(0);
    if (int rc = layernorm_bwd_params(R, (int)D, g->dln2_g, g->dln2_b, lnws, sw)) return rc;
    if (int rc = linear_bwd_weight(dh2, D, sv->a, 4 * D, R, D, 4 * D, g->dw2, g->db2, wws, wws_bytes, 0, 0, sw)) return rc;
    if (int rc = linear_bwd_input(dh2, D, w->w2, R, D, 4 * D, EPI_MUL_GELU_GRAD, sv->z, nullptr, dz, s)) return rc;
    // FFN1: dW1 = dz^T y1 ; dy1 = dz W1 + ds2 (residual)
    fork(1);
    if (int rc = linear_bwd_weight(dz, 4 * D, sv->y1, D, R, 4 * D, D, g->dw1, g->db1, wws, wws_bytes, 0, 0, sw)) return rc;
    if (int rc = linear_bwd_input(dz, 4 * D, w->w1, R, 4 * D, D, EPI_ADD_AUX, ds2, nullptr, dy1, s)) return rc;
    // LN1
    if (int rc = layernorm_bwd_rows(dy1, sv->s1, mean1, rstd1, w->ln1_g, R, (int)D, DCP(DC(p_drop, seed, site_base + 1), nullptr, row_map), ds1,
                                    p_drop > 0.f ? dh1 : nullptr, lnws1, s)) return rc;
    // attention output projection
    fork(2);
    if (int rc = layernorm_bwd_params(R, (int)D, g->dln1_g, g->dln1_b, lnws1, sw)) return rc;
    if (int rc = linear_bwd_weight(dh1, D, sv->ctx, D, R, D, D, g->dwo, g->dbo, wws, wws_bytes, 0, 0, sw)) return rc;
    if (int rc = linear_bwd_input(dh1, D, w->wo, R, D, D, EPI_NONE, nullptr, nullptr, dctx, s)) return rc;
    // attention core
    if (int rc = mhsa_bwd(sv->qkv, mask, sv->ctx, dctx, sv->row_max, sv->row_sum, dqkv, N, (int)Sq, (int)H, (int)(D / H),
                          DCP(DCA(p_drop, seed, site_base, mask_mode), offs, nullptr), s)) return rc;
    // QKV projection: packed dW [3D, D], db [3D]; dx = dqkv Wqkv + ds1 (residual)
    const bool gcontig = contiguous3(g->dwq, g->dwk, g->dwv, DD) && contiguous3(g->dbq, g->dbk, g->dbv, (size_t)D);
    float* dwp = gcontig ? g->dwq : dwqkv;
    float* dbp = gcontig ? g->dbq : dbqkv;
    fork(3);
    if (int rc = linear_bwd_weight(dqkv, 3 * D, x, D, R, 3 * D, D, dwp, dbp, wws, wws_bytes, 0, 0, sw)) return rc;
    if (!gcontig) {
        float* dws[3] = {g->dwq, g->dwk, g->dwv};
        float* dbs[3] = {g->dbq, g->dbk, g->dbv};
        for (int i = 0; i < 3; ++i) {
            if (int rc = copy_d2d(dws[i], dwqkv + i * DD, DD * sizeof(float), sw)) return rc;
            if (int rc = copy_d2d(dbs[i], dbqkv + i * D, D * sizeof(float), sw)) return rc;
        }
    }
    if (int rc = linear_bwd_input(dqkv, 3 * D, wqkv, R, 3 * D, D, EPI_ADD_AUX, ds1, nullptr, dx, s)) return rc;
    return su.finish();
}

int nr_tfm_layer_bwd_f32(const float* dy, const float* x, const int64_t* mask, int64_t N, int64_t Sq, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved* sv, nr_tfm_grads* g, float* dx, void* ws, size_t ws_bytes, void* stream) {
    return tfm_layer_bwd(dy, x, mask, N, Sq, N * Sq, D, H, w, p_drop, seed, site_base, mask_mode, sv, g, dx, ws, ws_bytes, nullptr, nullptr, S(stream));
}

int nr_tfm_layer_packed_bwd_f32(const float* dy, const float* x, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t Sq, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved* sv, nr_tfm_grads* g, float* dx, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(offs && row_map, "tfm_layer_packed_bwd: null offsets / row map");
    NR_CHECK_ARG(R <= N * Sq, "tfm_layer_packed_bwd: %lld packed rows exceed N*S = %lld", (long long)R, (long long)(N * Sq));
    NR_UNSUPPORTED(!mhsa_mfma_supported((int)Sq, (int)(H > 0 ? D / H : 0)), "tfm_layer_packed_bwd: packed sequences need S <= 32 and head dim <= 64");
    return tfm_layer_bwd(dy, x, mask, N, Sq, R, D, H, w, p_drop, seed, site_base, mask_mode, sv, g, dx, ws, ws_bytes, offs, row_map, S(stream));
}

// ------------------------------------------------------------------------------------ TFM layer, bf16 build
// The same layer with every GEMM operand and the attention tensors stored as bf16 in HBM (gemm_h.hip); the residual
// stream, the LayerNorm inputs / statistics, the softmax arithmetic and all parameter gradients stay f32.
static inline size_t pad256b(size_t bytes) { return (bytes + 255) / 256 * 256; }
static inline unsigned short* bumph(char*& cur, size_t elems) {
    unsigned short* p = reinterpret_cast<unsigned short*>(cur);
    cur += pad256b(elems * sizeof(unsigned short));
    return p;
}
static bool tfm_bf16_supported(int64_t Sq, int64_t D, int64_t H) {
    return D % 64 == 0 && H > 0 && D % H == 0 && mhsa_mfma_supported((int)Sq, (int)(D / H));
}
static int gemm_nt(const unsigned short* A, int64_t lda, const unsigned short* B, int64_t ldb, int64_t M, int64_t N, int64_t K, const float* bias, int epi,
                   float* C, unsigned short* Ch, const float* aux, const unsigned short* auxh, unsigned short* out2h, hipStream_t s,
                   unsigned short* ChT = nullptr, int64_t ldcht = 0) {
    GemmHParams p;
    p.ChT = ChT; p.ldcht = ldcht;
    p.A = A; p.lda = lda; p.B = B; p.ldb = ldb; p.M = (int)M; p.N = (int)N; p.K = (int)K;
    p.C = C; p.ldc = N; p.Ch = Ch; p.ldch = N; p.bias = bias; p.epi = epi;
    p.aux = aux; p.ldaux = N; p.auxh = auxh; p.ldauxh = N; p.out2h = out2h; p.ldout2h = N;
    return gemm_h(p, nullptr, 0, s);
}
int nr_tfm_layer_fwd_bf16(const float* x, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t Sq, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, nr_tfm_saved_bf16* sv, float* y, void* stream) {
    NR_CHECK_ARG(N >= 0 && Sq > 0 && D > 0 && H > 0 && R >= 0 && R <= N * Sq, "tfm_layer_fwd_bf16: bad sizes");
    NR_CHECK_ARG((offs == nullptr) == (row_map == nullptr) && (offs || R == N * Sq), "tfm_layer_fwd_bf16: packed rows need offs and row_map; dense rows R = N*S");
    NR_CHECK_ARG(w && sv && x && mask && y, "tfm_layer_fwd_bf16: null pointer");
    NR_UNSUPPORTED(!tfm_bf16_supported(Sq, D, H), "tfm_layer_fwd_bf16: needs D %% 64 == 0, S <= 32, head dim <= 64 (got S=%lld D=%lld H=%lld)", (long long)Sq, (long long)D, (long long)H);
    NR_CHECK_ARG(sv->x_h && sv->qkv_h && sv->ctx_h && sv->y1_h && sv->a_h && sv->s1 && sv->y1 && sv->s2 && sv->w_h, "tfm_layer_fwd_bf16: missing buffer");
    NR_CHECK_ARG(p_drop >= 0.f && p_drop < 1.f, "tfm_layer_fwd_bf16: p_drop %f not in [0,1)", p_drop);
    const size_t DD = (size_t)D * D;
    NR_CHECK_ARG(contiguous3(w->wq, w->wk, w->wv, DD) && contiguous3(w->bq, w->bk, w->bv, (size_t)D), "tfm_layer_fwd_bf16: q|k|v weights and biases must be contiguous (packed [3D, D])");
    if (N == 0 || R == 0) return NR_OK;
    hipStream_t s = S(stream);
    unsigned short* wqkv_h = sv->w_h, *wo_h = wqkv_h + 3 * DD, *w1_h = wo_h + DD, *w2_h = w1_h + 4 * DD;
    if (sv->wT_h) {
        // one pass per weight: the bf16 copy for this forward and the transposed bf16 copy the backward's dX products take
        unsigned short* t = sv->wT_h;
        if (int rc = transpose_bf16(w->wq, 0, D, 3 * D, D, t, 3 * D, s, wqkv_h)) return rc;                   // [3D,D] -> [D,3D]
        if (int rc = transpose_bf16(w->wo, 0, D, D, D, t + 3 * DD, D, s, wo_h)) return rc;
        if (int rc = transpose_bf16(w->w1, 0, D, 4 * D, D, t + 4 * DD, 4 * D, s, w1_h)) return rc;            // [4D,D] -> [D,4D]
        if (int rc = transpose_bf16(w->w2, 0, 4 * D, D, 4 * D, t + 8 * DD, D, s, w2_h)) return rc;            // [D,4D] -> [4D,D]
    } else {
        if (int rc = cast_bf16(w->wq, wqkv_h, 3 * DD, s)) return rc;
        if (int rc = cast_bf16(w->wo, wo_h, DD, s)) return rc;
        if (int rc = cast_bf16(w->w1, w1_h, 4 * DD, s)) return rc;
        if (int rc = cast_bf16(w->w2, w2_h, 4 * DD, s)) return rc;
    }
    if (int rc = cast_bf16(x, sv->x_h, R * D, s)) return rc;
    float* mean1 = sv->stats ? sv->stats : nullptr;
    float* rstd1 = sv->stats ? sv->stats + R : nullptr;
    float* mean2 = sv->stats ? sv->stats + 2 * R : nullptr;
    float* rstd2 = sv->stats ? sv->stats + 3 * R : nullptr;
    if (int rc = gemm_nt(sv->x_h, D, wqkv_h, D, R, 3 * D, D, w->bq, EPI_NONE, nullptr, sv->qkv_h, nullptr, nullptr, nullptr, s)) return rc;
    if (int rc = mhsa_fwd_mfma_h(sv->qkv_h, mask, sv->ctx_h, sv->row_max, sv->row_sum, N, (int)Sq, (int)H, (int)(D / H), DCP(DCA(p_drop, seed, site_base, mask_mode), offs, nullptr), s)) return rc;
    if (int rc = gemm_nt(sv->ctx_h, D, wo_h, D, R, D, D, w->bo, EPI_NONE, sv->s1, nullptr, nullptr, nullptr, nullptr, s)) return rc;
    if (int rc = ln_fwd(sv->s1, x, w->ln1_g, w->ln1_b, sv->y1, sv->s1, mean1, rstd1, R, (int)D, 1e-12f, DCP(DC(p_drop, seed, site_base + 1), nullptr, row_map), s, 0, sv->y1_h)) return rc;
    // (no transposed GELU output any more: the backward's dW2 product reads a_h as it is)
    if (int rc = gemm_nt(sv->y1_h, D, w1_h, D, R, 4 * D, D, w->b1, EPI_GELU, nullptr, sv->a_h, nullptr, nullptr, sv->z_h, s)) return rc;
    if (int rc = gemm_nt(sv->a_h, 4 * D, w2_h, 4 * D, R, D, 4 * D, w->b2, EPI_NONE, sv->s2, nullptr, nullptr, nullptr, nullptr, s)) return rc;
    return ln_fwd(sv->s2, sv->y1, w->ln2_g, w->ln2_b, y, sv->s2, mean2, rstd2, R, (int)D, 1e-12f, DCP(DC(p_drop, seed, site_base + 2), nullptr, row_map), s);
}

// dW [N,K] = dY^T X straight from the row-major bf16 tensors dy [R,N], x [R,K]: both operands reduction-major (transposed
// LDS reads in gemm_h_kernel), any R, no transposed copies (split over the reduction, slabs in ws)
static int gemm_dw_rm(const unsigned short* dy, const unsigned short* x, int64_t N, int64_t K, int64_t R, float* dw, float* ws, size_t ws_bytes, hipStream_t s) {
    GemmHParams p;
    p.A = dy; p.lda = N; p.a_rm = 1; p.B = x; p.ldb = K; p.b_rm = 1; p.M = (int)N; p.N = (int)K; p.K = (int)R;
    p.C = dw; p.ldc = K;
    p.ksplit = choose_ksplit_h(p.M, p.N, p.K);
    return gemm_h(p, ws, ws_bytes, s);
}

static size_t dw_ws_h(int64_t N, int64_t K, int64_t Rp) { return pad256b(gemm_h_workspace_bytes((int)N, (int)K, choose_ksplit_h((int)N, (int)K, (int)Rp))); }

size_t nr_tfm_layer_bwd_bf16_ws_bytes(int64_t R, int64_t D) {
    const int64_t Rp = (R + 63) / 64 * 64;
    size_t b = 0;
    b += 5 * pad256((size_t)R * D);                                        // ds2 dh2 dy1 ds1 dh1 (f32)
    b += 3 * pad256b((size_t)R * D * 2);                                   // dh2_h, dh1_h, dctx_h
    b += pad256b((size_t)R * 4 * D * 2);                                   // dz_h
    b += pad256b((size_t)R * 3 * D * 2);                                   // dqkv_h
    b += pad256b((size_t)4 * D * D * 2);                                   // transposed bf16 weight
    b += nr_layernorm_bwd_ws_bytes(R, D);
    b += pad256((size_t)colsum_chunks(R) * 4 * D);                         // column-sum partials
    size_t wmax = std::max(std::max(dw_ws_h(D, 4 * D, Rp), dw_ws_h(4 * D, D, Rp)), std::max(dw_ws_h(3 * D, D, Rp), dw_ws_h(D, D, Rp)));
    return b + wmax + 256;
}

int nr_tfm_layer_bwd_bf16(const float* dy, const int64_t* mask, const int32_t* offs, const int32_t* row_map, int64_t N, int64_t Sq, int64_t R, int64_t D, int64_t H, const nr_tfm_weights* w, float p_drop, uint64_t seed, uint32_t site_base, int mask_mode, const nr_tfm_saved_bf16* sv, nr_tfm_grads* g, float* dx, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(N > 0 && Sq > 0 && D > 0 && H > 0 && R > 0 && R <= N * Sq, "tfm_layer_bwd_bf16: bad sizes");
    NR_CHECK_ARG((offs == nullptr) == (row_map == nullptr) && (offs || R == N * Sq), "tfm_layer_bwd_bf16: packed rows need offs and row_map; dense rows R = N*S");
    NR_CHECK_ARG(dy && mask && w && sv && g && dx, "tfm_layer_bwd_bf16: null pointer");
    NR_UNSUPPORTED(!tfm_bf16_supported(Sq, D, H), "tfm_layer_bwd_bf16: needs D %% 64 == 0, S <= 32, head dim <= 64");
    NR_CHECK_ARG(sv->x_h && sv->qkv_h && sv->ctx_h && sv->y1_h && sv->z_h && sv->a_h && sv->s1 && sv->s2 && sv->stats && sv->row_max && sv->row_sum, "tfm_layer_bwd_bf16: missing saved activation");
    const size_t DD = (size_t)D * D;
    NR_CHECK_ARG(contiguous3(w->wq, w->wk, w->wv, DD), "tfm_layer_bwd_bf16: q|k|v weights must be contiguous");
    NR_CHECK_ARG(contiguous3(g->dwq, g->dwk, g->dwv, DD) && contiguous3(g->dbq, g->dbk, g->dbv, (size_t)D), "tfm_layer_bwd_bf16: q|k|v gradients must be contiguous");
    if (!ws || ws_bytes < nr_tfm_layer_bwd_bf16_ws_bytes(R, D)) { set_error("tfm_layer_bwd_bf16: workspace too small"); return NR_ERR_WORKSPACE; }
    hipStream_t s = S(stream);
    char* cur = (char*)ws;
    float* ds2 = bump(cur, (size_t)R * D);
    float* dh2 = p_drop > 0.f ? bump(cur, (size_t)R * D) : ds2;
    float* dy1 = bump(cur, (size_t)R * D);
    float* ds1 = bump(cur, (size_t)R * D);
    float* dh1 = p_drop > 0.f ? bump(cur, (size_t)R * D) : ds1;
    unsigned short* dh2_h = bumph(cur, (size_t)R * D);
    unsigned short* dh1_h = bumph(cur, (size_t)R * D);          // (not dh2_h again: the second stream may still be reading that one)
    unsigned short* dctx_h = bumph(cur, (size_t)R * D);
    unsigned short* dz_h = bumph(cur, (size_t)R * 4 * D);
    unsigned short* dqkv_h = bumph(cur, (size_t)R * 3 * D);
    unsigned short* wT = bumph(cur, (size_t)4 * DD);
    char* lnws = cur; cur += nr_layernorm_bwd_ws_bytes(R, D);
    float* cpart = bump(cur, (size_t)colsum_chunks(R) * 4 * D);
    float* slabs = reinterpret_cast<float*>(cur);
    const size_t slab_bytes = ws_bytes - (size_t)(cur - (char*)ws);
    const float* mean1 = sv->stats, *rstd1 = sv->stats + R, *mean2 = sv->stats + 2 * R, *rstd2 = sv->stats + 3 * R;
    // Everything that only serves a weight / bias gradient (the dW product and its slab sum, the column sums) runs on the
    // second stream, in order, so cpart / slabs are reused there without hazards; the main stream keeps LayerNorm, the weight
    // transposes, the dX products and the attention backward.  Round 3: the dW products read dY and X where they lie
    // (row-major [R, .] = reduction-major operands, gemm_dw_rm): the eight transposed copies per layer and the transposed
    // second outputs of two GEMM epilogues (aT, dzT: 2 x R x 4D x 2 bytes each) are gone.
    SideStream* ss = R >= SIDE_MIN_ROWS ? side_stream() : nullptr;
    SideUse su(ss, s);
    hipStream_t sw = su.stream();
    auto fork = [&](int i) { su.fork(i); sw = su.stream(); };

    // LN2 <- dy
    if (int rc = layernorm_bwd(dy, sv->s2, mean2, rstd2, w->ln2_g, R, (int)D, DCP(DC(p_drop, seed, site_base + 2), nullptr, row_map), ds2, p_drop > 0.f ? dh2 : nullptr, g->dln2_g, g->dln2_b, lnws, s, dh2_h)) return rc;
    // FFN2: dW2 [D,4D] = dh2^T a ; db2 ; dz = (dh2 W2) * gelu'(z)
    fork(0);
    if (int rc = gemm_dw_rm(dh2_h, sv->a_h, D, 4 * D, R, g->dw2, slabs, slab_bytes, sw)) return rc;
    if (int rc = colsum(dh2, D, R, (int)D, g->db2, cpart, sw)) return rc;
    const unsigned short* w2T = sv->wT_h ? sv->wT_h + 8 * DD : wT;
    if (!sv->wT_h) if (int rc = transpose_bf16(w->w2, 0, 4 * D, D, 4 * D, wT, D, s)) return rc;        // [D,4D] -> [4D,D]
    if (int rc = gemm_nt(dh2_h, D, w2T, D, R, 4 * D, D, nullptr, EPI_MUL_GELU_GRAD, nullptr, dz_h, nullptr, sv->z_h, nullptr, s)) return rc;
    // FFN1: dW1 [4D,D] = dz^T y1 ; db1 ; dy1 = dz W1 + ds2
    fork(1);
    if (int rc = gemm_dw_rm(dz_h, sv->y1_h, 4 * D, D, R, g->dw1, slabs, slab_bytes, sw)) return rc;
    if (int rc = colsum_h(dz_h, 4 * D, R, (int)(4 * D), g->db1, cpart, sw)) return rc;
    const unsigned short* w1T = sv->wT_h ? sv->wT_h + 4 * DD : wT;
    if (!sv->wT_h) if (int rc = transpose_bf16(w->w1, 0, D, 4 * D, D, wT, 4 * D, s)) return rc;        // [4D,D] -> [D,4D]
    if (int rc = gemm_nt(dz_h, 4 * D, w1T, 4 * D, R, D, 4 * D, nullptr, EPI_ADD_AUX, dy1, nullptr, ds2, nullptr, nullptr, s)) return rc;
    // LN1
    if (int rc = layernorm_bwd(dy1, sv->s1, mean1, rstd1, w->ln1_g, R, (int)D, DCP(DC(p_drop, seed, site_base + 1), nullptr, row_map), ds1, p_drop > 0.f ? dh1 : nullptr, g->dln1_g, g->dln1_b, lnws, s, dh1_h)) return rc;
    // attention output projection: dWo = dh1^T ctx ; dbo ; dctx = dh1 Wo
    fork(2);
    if (int rc = gemm_dw_rm(dh1_h, sv->ctx_h, D, D, R, g->dwo, slabs, slab_bytes, sw)) return rc;
    if (int rc = colsum(dh1, D, R, (int)D, g->dbo, cpart, sw)) return rc;
    const unsigned short* woT = sv->wT_h ? sv->wT_h + 3 * DD : wT;
    if (!sv->wT_h) if (int rc = transpose_bf16(w->wo, 0, D, D, D, wT, D, s)) return rc;
    if (int rc = gemm_nt(dh1_h, D, woT, D, R, D, D, nullptr, EPI_NONE, nullptr, dctx_h, nullptr, nullptr, nullptr, s)) return rc;
    // attention core
    if (int rc = mhsa_bwd_mfma_h(sv->qkv_h, mask, dctx_h, sv->row_max, sv->row_sum, dqkv_h, N, (int)Sq, (int)H, (int)(D / H), DCP(DCA(p_drop, seed, site_base, mask_mode), offs, nullptr), s)) return rc;
    // QKV projection: dWqkv [3D,D] = dqkv^T x ; dbqkv ; dx = dqkv Wqkv + ds1
    fork(3);
    if (int rc = gemm_dw_rm(dqkv_h, sv->x_h, 3 * D, D, R, g->dwq, slabs, slab_bytes, sw)) return rc;
    if (int rc = colsum_h(dqkv_h, 3 * D, R, (int)(3 * D), g->dbq, cpart, sw)) return rc;
    const unsigned short* wqT = sv->wT_h ? sv->wT_h : wT;
    if (!sv->wT_h) if (int rc = transpose_bf16(w->wq, 0, D, 3 * D, D, wT, 3 * D, s)) return rc;        // [3D,D] -> [D,3D]
    if (int rc = gemm_nt(dqkv_h, 3 * D, wqT, 3 * D, R, D, 3 * D, nullptr, EPI_ADD_AUX, dx, nullptr, ds1, nullptr, nullptr, s)) return rc;
    return su.finish();
}

// ------------------------------------------------------------------------------------ GRU user encoder
static bool g_gru_per_step = getenv("NR_GRU_PER_STEP") != nullptr;   // tests/bench: force the per-step path
int nr_gru_fwd_f32(const float* x, const int64_t* mask, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                   int64_t B, int64_t L, int64_t Din, int64_t H, float* gi, float* gh, float* hall, float* gates, int32_t* lens,
                   void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(B >= 0 && L > 0 && Din > 0 && H > 0, "gru_fwd: bad sizes");
    if (B == 0) return NR_OK;
    NR_CHECK_ARG(x && mask && w_ih && w_hh && b_ih && b_hh && gi && gh && hall && lens, "gru_fwd: null pointer");
    hipStream_t s = S(stream);
    if (int rc = gru_lens(mask, lens, (int)B, (int)L, s)) return rc;
    if (hipMemsetAsync(hall, 0, (size_t)B * H * sizeof(float), s) != hipSuccess) { set_error("gru_fwd: memset failed"); return NR_ERR_LAUNCH; }
    // input half of every step in one GEMM: GI[b*L+t, :] = x[b,t,:] W_ih^T + b_ih
    if (int rc = linear_fwd(x, Din, w_ih, b_ih, B * L, Din, 3 * H, EPI_NONE, gi, nullptr, s)) return rc;
    // the recurrence: one persistent launch when the shape allows it (the counter lives at the head of the workspace) ...
    if (ws && ws_bytes >= 256 && hipMemsetAsync(ws, 0, 2 * sizeof(unsigned), s) != hipSuccess) { set_error("gru_fwd: memset failed"); return NR_ERR_LAUNCH; }
    if (ws && ws_bytes >= 256 && gru_seq_fwd_supported((int)B, (int)H) && !g_gru_per_step) {
        const int rc = gru_seq_fwd(gi, w_hh, b_hh, lens, hall, gates, (int)B, (int)L, (int)H, (unsigned*)ws, s);
        if (rc != NR_ERR_UNSUPPORTED) return rc;
    }
    // ... else one skinny GEMM + one gate kernel per step
    if (ws && ws_bytes >= 256) { ws = (char*)ws + 256; ws_bytes -= 256; }
    for (int t = 0; t < (int)L; ++t) {
        const float* hp = hall + (size_t)t * B * H;
        if (int rc = linear_fwd(hp, H, w_hh, b_hh, B, H, 3 * H, EPI_NONE, gh, nullptr, s, (float*)ws, ws_bytes)) return rc;
        if (int rc = gru_gate_fwd(gi, gh, hp, lens, hall + (size_t)(t + 1) * B * H, gates ? gates + (size_t)t * B * 4 * H : nullptr,
                                  (int)B, (int)L, (int)H, t, s)) return rc;
    }
    return NR_OK;
}

size_t nr_gru_fwd_ws_bytes(int64_t B, int64_t H) { return skinny_ws_bytes(B, 3 * H, H) + 256; }

size_t nr_gru_bwd_ws_bytes(int64_t B, int64_t L, int64_t Din, int64_t H) {
    return 2 * pad256((size_t)B * L * 3 * H) + 3 * pad256((size_t)B * H) +
           std::max(std::max(bwd_weight_ws(B * L, 3 * H, H, true), bwd_weight_ws(B * L, 3 * H, Din, true)), skinny_ws_bytes(B, H, 3 * H)) + 2048;
}

int nr_gru_bwd_f32(const float* dh_out, const float* x, const float* w_ih, const float* w_hh, int64_t B, int64_t L, int64_t Din,
                   int64_t H, const float* hall, const float* gates, const int32_t* lens, float* dx, float* dw_ih, float* dw_hh,
                   float* db_ih, float* db_hh, void* ws, size_t ws_bytes, void* stream) {
    NR_CHECK_ARG(B > 0 && L > 0 && Din > 0 && H > 0, "gru_bwd: bad sizes");
    NR_CHECK_ARG(dh_out && x && w_ih && w_hh && hall && gates && lens && dw_ih && dw_hh && db_ih && db_hh, "gru_bwd: null pointer");
    NR_CHECK_ARG(ws && ws_bytes >= nr_gru_bwd_ws_bytes(B, L, Din, H), "gru_bwd: workspace too small");
    hipStream_t s = S(stream);
    char* cur = (char*)ws;
    unsigned* counter = reinterpret_cast<unsigned*>(bump(cur, 64));   // head of the workspace: nr_gru_status reads it
    if (hipMemsetAsync(counter, 0, 2 * sizeof(unsigned), s) != hipSuccess) { set_error("gru_bwd: memset failed"); return NR_ERR_LAUNCH; }
    float* dgi = bump(cur, (size_t)B * L * 3 * H);           // rows b*L + t  (the order of x)
    float* dgh = bump(cur, (size_t)B * L * 3 * H);           // rows t*B + b  (the order of hall)
    float* direct = bump(cur, (size_t)B * H);                 // dh_t * z
    float* pp[2] = {bump(cur, (size_t)B * H), bump(cur, (size_t)B * H)};
    const size_t rest = ws_bytes - (size_t)(cur - (char*)ws);
    const float* dh = dh_out;
    int k = 0;
    // the recurrence: one persistent launch when the shape allows it, else a gate kernel + a skinny GEMM per step
    bool stepwise = true;
    if (gru_seq_bwd_supported((int)B, (int)H) && !g_gru_per_step) {
        const int rc = gru_seq_bwd(dh_out, gates, hall, lens, w_hh, dgi, dgh, (int)B, (int)L, (int)H, counter, s);
        if (rc != NR_ERR_UNSUPPORTED) {
            if (rc != NR_OK) return rc;
            stepwise = false;
        }
    }
    for (int t = (int)L - 1; t >= 0 && stepwise; --t) {
        float* dgh_t = dgh + (size_t)t * B * 3 * H;
        if (int rc = gru_gate_bwd(dh, gates + (size_t)t * B * 4 * H, hall + (size_t)t * B * H, lens, dgi, dgh_t, direct, (int)B,
                                  (int)L, (int)H, t, s)) return rc;
        // dh_{t-1} = dh_t * z  +  dGH_t W_hh
        if (int rc = linear_bwd_input(dgh_t, 3 * H, w_hh, B, 3 * H, H, EPI_ADD_AUX, direct, nullptr, pp[k], s, (float*)cur, rest)) return rc;
        dh = pp[k];
        k ^= 1;
    }
    if (int rc = linear_bwd_weight(dgh, 3 * H, hall, H, B * L, 3 * H, H, dw_hh, db_hh, cur, rest, 0, 0, s)) return rc;
    if (int rc = linear_bwd_weight(dgi, 3 * H, x, Din, B * L, 3 * H, Din, dw_ih, db_ih, cur, rest, 0, 0, s)) return rc;
    if (dx) return linear_bwd_input(dgi, 3 * H, w_ih, B * L, 3 * H, Din, EPI_NONE, nullptr, nullptr, dx, s);
    return NR_OK;
}

int nr_gru_status(const void* ws, void* stream) {
    NR_CHECK_ARG(ws, "gru_status: null workspace");
    unsigned host[2] = {0, 0};
    if (hipStreamSynchronize(S(stream)) != hipSuccess || hipMemcpy(host, ws, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) {
        set_error("gru_status: %s", hipGetErrorString(hipGetLastError()));
        return NR_ERR_LAUNCH;
    }
    if (host[1] != 0) {
        set_error("gru: the grid barrier of the persistent recurrence timed out (a workgroup never arrived); the outputs of the "
                  "workgroups that gave up were overwritten with NaN");
        return NR_ERR_LAUNCH;
    }
    return NR_OK;
}

int nr_gru_debug_set_spin_limit(uint32_t polls) { gru_set_spin_limit(polls ? polls : (1u << 22)); return NR_OK; }

// ------------------------------------------------------------------------------------ evaluation metrics
int nr_impression_metrics_f64(const float* preds, const int64_t* labels, const uint8_t* masks, int64_t Ce, const int64_t* row_start,
                              const int32_t* row_count, int64_t G, int nk, const int* ks_host, double* out, int32_t* flags,
                              void* stream) {
    NR_CHECK_ARG(G >= 0 && Ce > 0 && nk >= 0 && nk <= 4, "impression_metrics: bad sizes (at most 4 nDCG cut-offs)");
    NR_CHECK_ARG(G == 0 || (preds && labels && masks && row_start && row_count && out && flags), "impression_metrics: null pointer");
    NR_CHECK_ARG(nk == 0 || ks_host, "impression_metrics: null cut-off list");
    for (int c = 0; c < nk; ++c) NR_CHECK_ARG(ks_host[c] > 0, "impression_metrics: cut-off %d must be positive", ks_host[c]);
    return impression_metrics(preds, labels, masks, (int)Ce, row_start, row_count, G, nk, ks_host, out, flags, S(stream));
}

}  // extern "C"
