// extern "C" surface of libftmi355 (declared in include/ftmi355.h): argument checking, translation of the
// plain-pointer C structs into the internal launch arguments, error reporting.
#include <stdlib.h>

#include <functional>
#include <mutex>
#include <thread>
#include <string.h>
#include <utility>
#include <vector>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

// Error messages live in a small ring, each entry tagged with the id of the thread whose call failed: ftmi_last_error() returns the newest
// entry of the CALLING thread (forward runs on the Python thread, backward on the autograd engine's thread -- neither can overwrite the
// message the other is about to read), falling back to the newest entry overall.  No thread-local storage inside the library.
namespace {
constexpr int kErrRing = 32;
struct ErrEntry {
    unsigned long long seq = 0;
    unsigned long long tid = 0;
    int code = 0;
    char msg[480] = "";
};
std::mutex g_err_mu;
ErrEntry g_err[kErrRing];
unsigned long long g_err_seq = 0;
unsigned long long self_tid() { return (unsigned long long)std::hash<std::thread::id>{}(std::this_thread::get_id()); }
}  // namespace

int set_error(int code, const char* msg) {
    std::lock_guard<std::mutex> lk(g_err_mu);
    ErrEntry& e = g_err[g_err_seq % kErrRing];
    e.seq = ++g_err_seq;
    e.tid = self_tid();
    e.code = code;
    strncpy(e.msg, msg, sizeof(e.msg) - 1);
    e.msg[sizeof(e.msg) - 1] = 0;
    return code;
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return 0;
    char buf[400];
    snprintf(buf, sizeof(buf), "%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return set_error(FTMI_ERR_LAUNCH, buf);
}

// ---------------- profiler ----------------
namespace {
struct ProfRec {
    hipEvent_t a, b;
    int k;
    double flops;
};
std::mutex g_prof_mu;
bool g_prof_on = false;
int g_prof_stride = 1;
long g_prof_count[PROF_NCLASS] = {};        // all launches seen while enabled
double g_prof_flops[PROF_NCLASS] = {};    // their algorithmic FLOPs
std::vector<ProfRec> g_prof_recs;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_prof_pool;
hipEvent_t g_prof_open[PROF_NCLASS];
double g_prof_open_flops[PROF_NCLASS];
hipEvent_t g_prof_open_b[PROF_NCLASS];
}  // namespace

int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? (int)strtol(e, nullptr, 0) : dflt;  // base 0: decimal, 0x.. or 0.. (bit-mask switches)
}

bool prof_enabled() { return g_prof_on; }

bool prof_begin(int k, double flops, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    const long n = g_prof_count[k]++;
    g_prof_flops[k] += flops;
    if (n % g_prof_stride != 0) return false;  // only every stride-th launch of a class is bracketed by events
    hipEvent_t a, b;
    if (!g_prof_pool.empty()) {
        a = g_prof_pool.back().first;
        b = g_prof_pool.back().second;
        g_prof_pool.pop_back();
    } else {
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
    }
    hipEventRecord(a, st);
    g_prof_open[k] = a;
    g_prof_open_b[k] = b;
    g_prof_open_flops[k] = flops;
    return true;
}

void prof_end(int k, hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    hipEventRecord(g_prof_open_b[k], st);
    g_prof_recs.push_back(ProfRec{g_prof_open[k], g_prof_open_b[k], k, g_prof_open_flops[k]});
}

size_t ltx_workspace_bytes(const ftmi_ltx_config& c);
int ltx_workspace_offset(const ftmi_ltx_config& c, const char* name, int layer, size_t* off);
int ltx_forward(const ftmi_ltx_config& c, const ftmi_ltx_weights& w, const bf16_t* x_t, const bf16_t* text, const float* key_bias,
                const float* sigma, bf16_t* pred, void* ws, size_t ws_bytes, hipStream_t st);
int ltx_backward_range(const ftmi_ltx_config& c, const ftmi_ltx_weights& w, const bf16_t* text, const float* key_bias, const bf16_t* dpred,
                       float* grad_a, float* grad_b, void* ws, size_t ws_bytes, int l_hi, int l_lo, int accumulate, hipStream_t st);

static int fill_attn(const ftmi_attn_desc* d, AttnArgs& a) {
    if (!d) return set_error(FTMI_ERR_INVALID, "attention: null descriptor");
    if (d->d != 64 && d->d != 128) return set_error(FTMI_ERR_UNSUPPORTED, "attention: head_dim must be 64 (or 128, forward only)");
    a.d = d->d;
    a.B = d->B; a.H = d->H; a.Sq = d->Sq; a.Sk = d->Sk; a.scale = d->scale;
    a.q_sb = d->q_strides[0]; a.q_sh = d->q_strides[1]; a.q_ss = d->q_strides[2];
    a.k_sb = d->k_strides[0]; a.k_sh = d->k_strides[1]; a.k_ss = d->k_strides[2];
    a.v_sb = d->v_strides[0]; a.v_sh = d->v_strides[1]; a.v_ss = d->v_strides[2];
    a.o_sb = d->o_strides[0]; a.o_sh = d->o_strides[1]; a.o_ss = d->o_strides[2];
    a.do_sb = d->do_strides[0]; a.do_sh = d->do_strides[1]; a.do_ss = d->do_strides[2];
    a.dq_sb = d->dq_strides[0]; a.dq_sh = d->dq_strides[1]; a.dq_ss = d->dq_strides[2];
    a.dk_sb = d->dk_strides[0]; a.dk_sh = d->dk_strides[1]; a.dk_ss = d->dk_strides[2];
    a.dv_sb = d->dv_strides[0]; a.dv_sh = d->dv_strides[1]; a.dv_ss = d->dv_strides[2];
    a.kb_sb = d->bias_strides[0]; a.kb_sh = d->bias_strides[1];
    return 0;
}

}  // namespace ftmi

using namespace ftmi;

extern "C" {

int ftmi_version(void) { return 100; }

int ftmi_prof_enable(int stride) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = stride > 0;
    g_prof_stride = stride > 0 ? stride : 1;
    if (g_prof_on) {
        // Create the events NOW, outside any timed region: hipEventCreate on demand inside the step occasionally stalled the
        // launch thread for ~25 ms (seen as one 90 ms step in ten bench runs; never with the profiler off).
        const size_t want = 4096;
        g_prof_recs.reserve(want);
        while (g_prof_pool.size() < want) {
            hipEvent_t a, b;
            if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) break;
            g_prof_pool.emplace_back(a, b);
        }
    }
    return 0;
}

int ftmi_prof_summary(int kclass, double* total_ms, long* launches, double* total_flops, long* all_launches, double* all_flops,
                      int reset) {
    if (kclass < 0 || kclass >= PROF_NCLASS) return set_error(FTMI_ERR_INVALID, "ftmi_prof_summary: bad kernel class");
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double ms = 0, fl = 0;
    long n = 0;
    std::vector<ProfRec> keep;
    for (const ProfRec& r : g_prof_recs) {
        if (r.k != kclass) {
            keep.push_back(r);
            continue;
        }
        float t = 0.f;
        if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            ms += t;
            fl += r.flops;
            ++n;
        }
        if (reset) g_prof_pool.push_back({r.a, r.b});
        else keep.push_back(r);
    }
    if (reset) g_prof_recs.swap(keep);
    if (total_ms) *total_ms = ms;
    if (launches) *launches = n;
    if (total_flops) *total_flops = fl;
    if (all_launches) *all_launches = g_prof_count[kclass];
    if (all_flops) *all_flops = g_prof_flops[kclass];
    if (reset) {
        g_prof_count[kclass] = 0;
        g_prof_flops[kclass] = 0;
    }
    return 0;
}

int ftmi_last_error(char* buf, size_t len) {
    if (!buf || len == 0) return FTMI_ERR_INVALID;
    std::lock_guard<std::mutex> lk(g_err_mu);
    const unsigned long long me = self_tid();
    const ErrEntry* mine = nullptr;
    const ErrEntry* newest = nullptr;
    for (const ErrEntry& e : g_err) {
        if (!e.seq) continue;
        if (!newest || e.seq > newest->seq) newest = &e;
        if (e.tid == me && (!mine || e.seq > mine->seq)) mine = &e;
    }
    const ErrEntry* pick = mine ? mine : newest;
    strncpy(buf, pick ? pick->msg : "", len - 1);
    buf[len - 1] = 0;
    return 0;
}

int ftmi_attn_fwd(const ftmi_attn_desc* desc, const void* q, const void* k, const void* v, void* out, float* lse, const float* key_bias,
                  ftmi_stream stream) {
    AttnArgs a;
    int rc = fill_attn(desc, a);
    if (rc) return rc;
    if (!q || !k || !v || !out) return set_error(FTMI_ERR_INVALID, "ftmi_attn_fwd: null tensor");
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)out; a.lse2 = lse; a.kbias = key_bias;
    return attn_fwd(a, (hipStream_t)stream);
}

int ftmi_attn_bwd(const ftmi_attn_desc* desc, const void* q, const void* k, const void* v, const void* out, const float* lse,
                  const void* dout, void* dq, void* dk, void* dv, float* delta_ws, const float* key_bias, ftmi_stream stream) {
    AttnArgs a;
    int rc = fill_attn(desc, a);
    if (rc) return rc;
    if (!q || !k || !v || !out || !lse || !dout || !dq || !dk || !dv || !delta_ws) return set_error(FTMI_ERR_INVALID, "ftmi_attn_bwd: null tensor");
    a.q = (const bf16_t*)q; a.k = (const bf16_t*)k; a.v = (const bf16_t*)v; a.o = (bf16_t*)out; a.lse2 = (float*)lse; a.kbias = key_bias;
    a.dout = (const bf16_t*)dout; a.dq = (bf16_t*)dq; a.dk = (bf16_t*)dk; a.dv = (bf16_t*)dv; a.delta = delta_ws;
    return attn_bwd(a, (hipStream_t)stream);
}

int ftmi_gemm_nt_plan(int M, int N, int K, int K2, int epilogue) { return gemm_nt_plan(M, N, K, K2, epilogue); }

int ftmi_reload_switches(void) { return EnvSwitch::reload_all(); }

int ftmi_fused_status(void) { return gemm_fused_status(); }

int ftmi_gemm_nt(int M, int N, int K, const void* x, long ldx, const void* w, long ldw, const void* bias, float alpha, void* out, long ldo,
                 int epilogue, void* out2, const void* resid, const void* gate, int rows_per_batch, const void* aux, long ld_side, int variant,
                 ftmi_stream stream) {
    if (!x || !w || !out) return set_error(FTMI_ERR_INVALID, "ftmi_gemm_nt: null tensor");
    if (epilogue < 0 || epilogue > 3) return set_error(FTMI_ERR_INVALID, "ftmi_gemm_nt: bad epilogue");
    if (epilogue == EPI_RESID && !resid) return set_error(FTMI_ERR_INVALID, "ftmi_gemm_nt: residual epilogue without residual");
    if (epilogue == EPI_DGELU && !aux) return set_error(FTMI_ERR_INVALID, "ftmi_gemm_nt: gelu' epilogue without pre-activation");
    GemmNtArgs a;
    a.X = (const bf16_t*)x; a.ldx = ldx; a.W = (const bf16_t*)w; a.ldw = ldw; a.M = M; a.N = N; a.K = K;
    a.bias = (const bf16_t*)bias; a.alpha = alpha; a.out = (bf16_t*)out; a.ldo = ldo; a.epi = epilogue;
    if (ld_side < 0 || (ld_side % 8) || (ld_side > 0 && ld_side < N)) return set_error(FTMI_ERR_INVALID, "ftmi_gemm_nt: bad side stride");
    const long lds = ld_side > 0 ? ld_side : ldo;  // row stride of out2 / resid / aux
    a.out2 = (bf16_t*)out2; a.ldo2 = lds; a.resid = (const bf16_t*)resid; a.ldr = lds;
    a.gate = (const bf16_t*)gate; a.gate_bstride = N; a.rows_per_batch = rows_per_batch > 0 ? rows_per_batch : M;
    a.aux = (const bf16_t*)aux; a.ldaux = lds; a.variant = variant;
    return gemm_nt(a, (hipStream_t)stream);
}

#ifdef FTMI_EXPERIMENTAL
int ftmi_gemm_sk_plan(int ntiles, int n_workgroups, int nk, int owner_cost, int min_piece, int partial_cost, int add_cost, int* work) {
    if (!work || ntiles < 0 || n_workgroups < 1 || nk < 1 || owner_cost < 0 || min_piece < 1 || 2 * min_piece > nk + 1 || partial_cost < 0 || add_cost < 0)
        return set_error(FTMI_ERR_INVALID, "ftmi_gemm_sk_plan: bad arguments");
    if (!sk_build_work(ntiles, n_workgroups, nk, owner_cost, min_piece, partial_cost, add_cost, work)) return set_error(FTMI_ERR_UNSUPPORTED, "ftmi_gemm_sk_plan: no split for this geometry");
    return 0;
}

int ftmi_gemm_sk_status(void) { return gemm_nt_sk_status(); }

int ftmi_gemm_sk_trace(unsigned long long* out, int capacity) { return gemm_nt_sk_trace(out, capacity); }
#endif

int ftmi_gemm_tn(int M, int P, int Q, const void* u, long ldu, const void* v, long ldv, float* c, long ldc, float scale, ftmi_stream stream) {
    if (!u || !v || !c) return set_error(FTMI_ERR_INVALID, "ftmi_gemm_tn: null tensor");
    GemmTnArgs a;
    a.U = (const bf16_t*)u; a.ldu = ldu; a.V = (const bf16_t*)v; a.ldv = ldv; a.C = c; a.ldc = ldc; a.M = M; a.P = P; a.Q = Q; a.scale = scale;
    return gemm_tn(a, (hipStream_t)stream);
}

int ftmi_fp8_upcast(const void* src, void* dst, int rows, int cols, int transpose, ftmi_stream stream) {
    if (!src || !dst) return set_error(FTMI_ERR_INVALID, "ftmi_fp8_upcast: null tensor");
    return fp8_upcast((const uint8_t*)src, (bf16_t*)dst, rows, cols, transpose, (hipStream_t)stream);
}

int ftmi_transpose_bf16(const void* in, void* out, int rows, int cols, ftmi_stream stream) {
    if (!in || !out) return set_error(FTMI_ERR_INVALID, "ftmi_transpose_bf16: null tensor");
    return transpose_bf16((const bf16_t*)in, (bf16_t*)out, rows, cols, (hipStream_t)stream);
}

static int lora_down_call(const bf16_t* X, long ldx, int M, const bf16_t* w_sp, int r, int K, float alpha, bf16_t* out, hipStream_t st) {
    GemmNtArgs d;
    d.X = X; d.ldx = ldx; d.W = w_sp; d.ldw = K; d.M = M; d.N = 2 * r; d.K = K; d.alpha = alpha; d.split_r = r;
    d.out = out; d.ldo = 3L * r; d.variant = 8;
    return gemm_nt(d, st);
}

int ftmi_linear_lora_fwd(int M, int K, int N, int r, float lora_scale, const void* x, const void* w, const void* bias, const void* a_sp,
                         const void* b_ext, void* y, void* xa_out, int variant, ftmi_stream stream) {
    if (!x || !w || !y) return set_error(FTMI_ERR_INVALID, "ftmi_linear_lora_fwd: null tensor");
    if (r > 0 && (!a_sp || !b_ext || !xa_out)) return set_error(FTMI_ERR_INVALID, "ftmi_linear_lora_fwd: LoRA tensors missing");
    if (r < 0 || (r % 64)) return set_error(FTMI_ERR_UNSUPPORTED, "ftmi_linear_lora_fwd: rank must be 0 or a multiple of 64");
    hipStream_t st = (hipStream_t)stream;
    GemmNtArgs a;
    a.X = (const bf16_t*)x; a.ldx = K; a.W = (const bf16_t*)w; a.ldw = K; a.M = M; a.N = N; a.K = K;
    a.bias = (const bf16_t*)bias; a.out = (bf16_t*)y; a.ldo = N; a.variant = variant;
    if (r > 0) {
        int rc = lora_down_call((const bf16_t*)x, K, M, (const bf16_t*)a_sp, r, K, lora_scale, (bf16_t*)xa_out, st);
        if (rc) return rc;
        a.X2 = (const bf16_t*)xa_out; a.ldx2 = 3 * r; a.W2 = (const bf16_t*)b_ext; a.ldw2 = 3 * r; a.K2 = 3 * r;
    }
    return gemm_nt(a, st);
}

int ftmi_linear_lora_bwd(int M, int K, int N, int r, float lora_scale, const void* x, const void* dy, const void* xa, const void* w_t,
                         const void* bt_sp, const void* at_ext, void* dxa_ws, void* dx, float* grad_a, float* grad_b, int variant,
                         ftmi_stream stream) {
    if (!dy || (dx && !w_t)) return set_error(FTMI_ERR_INVALID, "ftmi_linear_lora_bwd: null tensor");
    if (r < 0 || (r % 64)) return set_error(FTMI_ERR_UNSUPPORTED, "ftmi_linear_lora_bwd: rank must be 0 or a multiple of 64");
    if (r > 0 && (!x || !xa || !bt_sp || !at_ext || !dxa_ws || !grad_a || !grad_b))
        return set_error(FTMI_ERR_INVALID, "ftmi_linear_lora_bwd: LoRA tensors missing");
    hipStream_t st = (hipStream_t)stream;
    if (r > 0) {  // dxa = s * dy B  (B^T is the K-contiguous operand), fp32-equivalent
        int rc = lora_down_call((const bf16_t*)dy, N, M, (const bf16_t*)bt_sp, r, N, lora_scale, (bf16_t*)dxa_ws, st);
        if (rc) return rc;
    }
    if (dx) {
        GemmNtArgs a;  // dx = dy W (+ dxa A as a K-extension)
        a.X = (const bf16_t*)dy; a.ldx = N; a.W = (const bf16_t*)w_t; a.ldw = N; a.M = M; a.N = K; a.K = N;
        a.out = (bf16_t*)dx; a.ldo = K; a.variant = variant;
        if (r > 0) { a.X2 = (const bf16_t*)dxa_ws; a.ldx2 = 3 * r; a.W2 = (const bf16_t*)at_ext; a.ldw2 = 3 * r; a.K2 = 3 * r; }
        int rc = gemm_nt(a, st);
        if (rc) return rc;
    }
    if (r > 0) {
        GemmTnArgs t;  // dB += dy^T xa   (xa = hi + lo planes)
        t.U = (const bf16_t*)dy; t.ldu = N; t.V = (const bf16_t*)xa; t.ldv = 3 * r; t.v_fold = r; t.C = grad_b; t.ldc = r; t.M = M; t.P = N; t.Q = r;
        int rc = gemm_tn(t, st);
        if (rc) return rc;
        GemmTnArgs u;  // dA += dxa^T x
        u.U = (const bf16_t*)dxa_ws; u.ldu = 3 * r; u.u_fold = r; u.V = (const bf16_t*)x; u.ldv = K; u.C = grad_a; u.ldc = K; u.M = M; u.P = r; u.Q = K;
        rc = gemm_tn(u, st);
        if (rc) return rc;
    }
    return 0;
}

int ftmi_norm_modulate_fwd(const void* x, const void* shift, const void* onep, long mod_bstride, void* y, int rows, int rows_per_batch, int D,
                           float eps, int layernorm, ftmi_stream stream) {
    if (!x || !shift || !onep || !y || rows <= 0 || rows_per_batch <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_norm_modulate_fwd: bad argument");
    return norm_modulate_fwd((const bf16_t*)x, (const bf16_t*)shift, (const bf16_t*)onep, mod_bstride, (bf16_t*)y, rows, rows_per_batch, D, eps, layernorm,
                             (hipStream_t)stream);
}

int ftmi_norm_modulate_bwd(const void* x, const void* dy, const void* onep, long mod_bstride, const void* dres, void* dx, int rows,
                           int rows_per_batch, int D, float eps, int layernorm, ftmi_stream stream) {
    if (!x || !dy || !onep || !dx || rows <= 0 || rows_per_batch <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_norm_modulate_bwd: bad argument");
    return norm_modulate_bwd((const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)onep, mod_bstride, (const bf16_t*)dres, (bf16_t*)dx, rows, rows_per_batch, D,
                             eps, layernorm, (hipStream_t)stream);
}

int ftmi_qknorm_rope_fwd(const void* x, long ldx, const void* w, const float* cos_t, const float* sin_t, void* y, long ldy, int rows,
                         int rows_per_batch, int D, float eps, ftmi_stream stream) {
    if (!x || !w || !y || rows <= 0 || rows_per_batch <= 0 || (!cos_t) != (!sin_t)) return set_error(FTMI_ERR_INVALID, "ftmi_qknorm_rope_fwd: bad argument");
    return qknorm_rope_fwd((const bf16_t*)x, ldx, (const bf16_t*)w, cos_t, sin_t, (bf16_t*)y, ldy, rows, rows_per_batch, D, eps, (hipStream_t)stream);
}

int ftmi_qknorm_rope_bwd(const void* x, long ldx, const void* w, const float* cos_t, const float* sin_t, const void* dy, long lddy, void* dx,
                         long lddx, int rows, int rows_per_batch, int D, float eps, ftmi_stream stream) {
    if (!x || !w || !dy || !dx || rows <= 0 || rows_per_batch <= 0 || (!cos_t) != (!sin_t)) return set_error(FTMI_ERR_INVALID, "ftmi_qknorm_rope_bwd: bad argument");
    return qknorm_rope_bwd((const bf16_t*)x, ldx, (const bf16_t*)w, cos_t, sin_t, (const bf16_t*)dy, lddy, (bf16_t*)dx, lddx, rows, rows_per_batch, D, eps,
                           (hipStream_t)stream);
}

size_t ftmi_ltx_workspace_bytes(const ftmi_ltx_config* cfg) { return cfg ? ltx_workspace_bytes(*cfg) : 0; }

int ftmi_ltx_workspace_offset(const ftmi_ltx_config* cfg, const char* name, int layer, size_t* offset) {
    if (!cfg || !name || !offset) return set_error(FTMI_ERR_INVALID, "ftmi_ltx_workspace_offset: null argument");
    return ltx_workspace_offset(*cfg, name, layer, offset);
}

int ftmi_ltx_forward(const ftmi_ltx_config* cfg, const ftmi_ltx_weights* w, const void* x_t, const void* text, const float* key_bias,
                     const float* sigma, void* pred, void* ws, size_t ws_bytes, ftmi_stream stream) {
    if (!cfg || !w || !x_t || !text || !sigma || !pred || !ws) return set_error(FTMI_ERR_INVALID, "ftmi_ltx_forward: null argument");
    if (cfg->r > 0 && (!w->lora_a_sp || !w->lora_b_ext)) return set_error(FTMI_ERR_INVALID, "ftmi_ltx_forward: LoRA working copies missing");
    return ltx_forward(*cfg, *w, (const bf16_t*)x_t, (const bf16_t*)text, key_bias, sigma, (bf16_t*)pred, ws, ws_bytes, (hipStream_t)stream);
}

int ftmi_ltx_backward_range(const ftmi_ltx_config* cfg, const ftmi_ltx_weights* w, const void* text, const float* key_bias, const void* dpred,
                            float* grad_a, float* grad_b, void* ws, size_t ws_bytes, int l_hi, int l_lo, int accumulate, ftmi_stream stream) {
    if (!cfg || !w || !dpred || !ws) return set_error(FTMI_ERR_INVALID, "ftmi_ltx_backward: null argument");
    if (cfg->r > 0 && (!grad_a || !grad_b || !w->lora_at_ext || !w->lora_bt_sp || !w->lora_at_qkv_ext))
        return set_error(FTMI_ERR_INVALID, "ftmi_ltx_backward: LoRA gradient buffers / working copies missing");
    return ltx_backward_range(*cfg, *w, (const bf16_t*)text, key_bias, (const bf16_t*)dpred, grad_a, grad_b, ws, ws_bytes, l_hi, l_lo, accumulate,
                              (hipStream_t)stream);
}

int ftmi_ltx_backward(const ftmi_ltx_config* cfg, const ftmi_ltx_weights* w, const void* text, const float* key_bias, const void* dpred,
                      float* grad_a, float* grad_b, void* ws, size_t ws_bytes, ftmi_stream stream) {
    if (!cfg) return set_error(FTMI_ERR_INVALID, "ftmi_ltx_backward: null argument");
    return ftmi_ltx_backward_range(cfg, w, text, key_bias, dpred, grad_a, grad_b, ws, ws_bytes, cfg->L, 0, /*accumulate=*/1, stream);
}

int ftmi_ltx_noise_pack(const void* latents, const void* noise, const float* mean, const float* std_, const float* sigma,
                        const float* sigma_first, int first_frame_tokens, void* x_t, void* target, int B, int C, int S, ftmi_stream stream) {
    if (!latents || !noise || !mean || !std_ || !sigma || !x_t || !target) return set_error(FTMI_ERR_INVALID, "ftmi_ltx_noise_pack: null argument");
    return noise_pack((const bf16_t*)latents, (const bf16_t*)noise, mean, std_, sigma, sigma_first, first_frame_tokens, (bf16_t*)x_t,
                      (bf16_t*)target, B, C, S, (hipStream_t)stream);
}

int ftmi_ddim_add_noise(const void* latents, const void* noise, const float* sqrt_alpha, const float* sqrt_one_minus_alpha, float scaling_factor,
                        void* x0, void* noisy, int B, long per_sample, ftmi_stream stream) {
    if (!latents || !noise || !sqrt_alpha || !sqrt_one_minus_alpha || !noisy || B <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_ddim_add_noise: bad argument");
    return ddim_mix((const bf16_t*)latents, (const bf16_t*)noise, sqrt_alpha, sqrt_one_minus_alpha, scaling_factor, (bf16_t*)x0, (bf16_t*)noisy, B, per_sample, 0,
                    (hipStream_t)stream);
}

int ftmi_ddim_get_velocity(const void* sample, const void* noise, const float* sqrt_alpha, const float* sqrt_one_minus_alpha, void* out, int B,
                           long per_sample, ftmi_stream stream) {
    if (!sample || !noise || !sqrt_alpha || !sqrt_one_minus_alpha || !out || B <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_ddim_get_velocity: bad argument");
    return ddim_mix((const bf16_t*)sample, (const bf16_t*)noise, sqrt_alpha, sqrt_one_minus_alpha, 1.0f, nullptr, (bf16_t*)out, B, per_sample, 1, (hipStream_t)stream);
}

int ftmi_posterior_sample(const void* moments, const void* eps, void* out, int B, long per_sample, ftmi_stream stream) {
    if (!moments || !eps || !out || B <= 0 || per_sample <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_posterior_sample: bad argument");
    return posterior_sample((const bf16_t*)moments, (const bf16_t*)eps, (bf16_t*)out, B, per_sample, (hipStream_t)stream);
}

int ftmi_cog_ln_mod_fwd(const void* x, const void* w, const void* b, const void* shift, const void* onep, void* y, int rows, int D,
                        int rows_per_batch, int text_len, float eps, ftmi_stream stream) {
    if (!x || !w || !b || !shift || !onep || !y) return set_error(FTMI_ERR_INVALID, "ftmi_cog_ln_mod_fwd: null argument");
    CogLnArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b; a.shift = (const bf16_t*)shift; a.onep = (const bf16_t*)onep;
    a.y = (bf16_t*)y; a.rows = rows; a.D = D; a.rows_per_batch = rows_per_batch; a.seg0 = text_len; a.eps = eps;
    return cog_ln_mod_fwd(a, (hipStream_t)stream);
}

int ftmi_cog_ln_mod_bwd(const void* x, const void* w, const void* onep, const void* dy, const void* dres, void* dx, int rows, int D,
                        int rows_per_batch, int text_len, float eps, ftmi_stream stream) {
    if (!x || !w || !onep || !dy || !dx) return set_error(FTMI_ERR_INVALID, "ftmi_cog_ln_mod_bwd: null argument");
    CogLnArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.onep = (const bf16_t*)onep; a.dy = (const bf16_t*)dy; a.dres = (const bf16_t*)dres;
    a.dx = (bf16_t*)dx; a.rows = rows; a.D = D; a.rows_per_batch = rows_per_batch; a.seg0 = text_len; a.eps = eps;
    return cog_ln_mod_bwd(a, (hipStream_t)stream);
}

int ftmi_cog_head_ln_fwd(const void* x, long ld, const void* w, const void* b, void* y, int rows, int D, float eps, const float* rope_cos,
                         const float* rope_sin, int rows_per_batch, int text_len, ftmi_stream stream) {
    if (!x || !w || !b || !y || ld < D || (ld % 8) || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_cog_head_ln_fwd: bad argument");
    CogLnArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.b = (const bf16_t*)b; a.y = (bf16_t*)y; a.rows = rows; a.D = D; a.ld = ld; a.eps = eps;
    a.cos = rope_cos; a.sin = rope_sin; a.seg0 = rope_cos ? text_len : 0;
    a.rows_per_batch = rope_cos ? rows_per_batch : (rows > 0 ? rows : 1);
    return cog_head_ln_fwd(a, (hipStream_t)stream);
}

int ftmi_cog_head_ln_bwd(const void* x, long ld, const void* w, const void* dy, void* dx, int rows, int D, float eps, const float* rope_cos,
                         const float* rope_sin, int rows_per_batch, int text_len, ftmi_stream stream) {
    if (!x || !w || !dy || !dx || ld < D || (ld % 8) || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_cog_head_ln_bwd: bad argument");
    CogLnArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.dy = (const bf16_t*)dy; a.dx = (bf16_t*)dx; a.rows = rows; a.D = D; a.ld = ld; a.eps = eps;
    a.cos = rope_cos; a.sin = rope_sin; a.seg0 = rope_cos ? text_len : 0;
    a.rows_per_batch = rope_cos ? rows_per_batch : (rows > 0 ? rows : 1);
    return cog_head_ln_bwd(a, (hipStream_t)stream);
}

int ftmi_head_rms_rope_fwd(const void* x, long ld, const void* w, void* y, long ld_y, int rows, int D, int head_dim, float eps, const float* rope_cos,
                           const float* rope_sin, int rows_per_batch, int rope_from, ftmi_stream stream) {
    if (!x || !w || !y || ld < D || (ld % 8) || (ld_y % 8) || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_head_rms_rope_fwd: bad argument");
    CogLnArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.y = (bf16_t*)y; a.rows = rows; a.D = D; a.ld = ld; a.ld_out = ld_y; a.eps = eps; a.head_dim = head_dim; a.rms = 1;
    a.cos = rope_cos; a.sin = rope_sin; a.seg0 = rope_cos ? rope_from : 0;
    a.rows_per_batch = rope_cos ? rows_per_batch : (rows > 0 ? rows : 1);
    return cog_head_ln_fwd(a, (hipStream_t)stream);
}

int ftmi_head_rms_rope_bwd(const void* x, long ld, const void* w, const void* dy, long ld_dy, void* dx, long ld_dx, int rows, int D, int head_dim, float eps,
                           const float* rope_cos, const float* rope_sin, int rows_per_batch, int rope_from, ftmi_stream stream) {
    if (!x || !w || !dy || !dx || ld < D || (ld % 8) || (ld_dy % 8) || (ld_dx % 8) || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_head_rms_rope_bwd: bad argument");
    CogLnArgs a;
    a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.dy = (const bf16_t*)dy; a.dx = (bf16_t*)dx; a.rows = rows; a.D = D; a.ld = ld; a.ld_dy = ld_dy; a.ld_out = ld_dx;
    a.eps = eps; a.head_dim = head_dim; a.rms = 1;
    a.cos = rope_cos; a.sin = rope_sin; a.seg0 = rope_cos ? rope_from : 0;
    a.rows_per_batch = rope_cos ? rows_per_batch : (rows > 0 ? rows : 1);
    return cog_head_ln_bwd(a, (hipStream_t)stream);
}

int ftmi_cog_gate_residual(const void* res, const void* y, const void* gate, void* out, int rows, int D, int rows_per_batch, int text_len,
                           ftmi_stream stream) {
    if (!y || !gate || !out) return set_error(FTMI_ERR_INVALID, "ftmi_cog_gate_residual: null argument");
    CogLnArgs a;
    a.x = (const bf16_t*)y; a.onep = (const bf16_t*)gate; a.dres = (const bf16_t*)res; a.y = (bf16_t*)out; a.rows = rows; a.D = D;
    a.rows_per_batch = rows_per_batch; a.seg0 = text_len;
    return cog_gate_residual(a, (hipStream_t)stream);
}

size_t ftmi_hy_single_saved_bytes(const ftmi_hy_single_config* cfg) { return cfg ? hy_single_saved_bytes(*cfg) : 0; }
size_t ftmi_hy_single_scratch_bytes(const ftmi_hy_single_config* cfg) { return cfg ? hy_single_scratch_bytes(*cfg) : 0; }
int ftmi_hy_single_forward(const ftmi_hy_single_config* cfg, const ftmi_hy_single_weights* w, const void* x, const void* temb_silu, const float* key_bias,
                           const float* rope_cos, const float* rope_sin, void* out, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                           ftmi_stream stream) {
    if (!cfg || !w || !x || !temb_silu || !saved || !scratch || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_hy_single_forward: bad argument");
    return hy_single_forward(*cfg, *w, (const bf16_t*)x, (const bf16_t*)temb_silu, key_bias, rope_cos, rope_sin, (bf16_t*)out, saved, saved_bytes, scratch,
                             scratch_bytes, (hipStream_t)stream);
}
int ftmi_hy_single_backward(const ftmi_hy_single_config* cfg, const ftmi_hy_single_weights* w, const void* x, const void* dout, const float* key_bias,
                            const float* rope_cos, const float* rope_sin, const void* ones_rows, void* dx, float* grad_a, float* grad_b, void* saved,
                            size_t saved_bytes, void* scratch, size_t scratch_bytes, ftmi_stream stream) {
    if (!cfg || !w || !x || !dout || !ones_rows || !dx || !saved || !scratch || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_hy_single_backward: bad argument");
    return hy_single_backward(*cfg, *w, (const bf16_t*)x, (const bf16_t*)dout, key_bias, rope_cos, rope_sin, (const bf16_t*)ones_rows, (bf16_t*)dx, grad_a,
                              grad_b, saved, saved_bytes, scratch, scratch_bytes, (hipStream_t)stream);
}

size_t ftmi_hy_dual_saved_bytes(const ftmi_hy_dual_config* cfg) { return cfg ? hy_dual_saved_bytes(*cfg) : 0; }
size_t ftmi_hy_dual_scratch_bytes(const ftmi_hy_dual_config* cfg) { return cfg ? hy_dual_scratch_bytes(*cfg) : 0; }
int ftmi_hy_dual_forward(const ftmi_hy_dual_config* cfg, const ftmi_hy_dual_weights* w, const void* x_v, const void* x_t, const void* temb_silu,
                         const float* key_bias, const float* rope_cos, const float* rope_sin, void* out_v, void* out_t, void* saved, size_t saved_bytes,
                         void* scratch, size_t scratch_bytes, ftmi_stream stream) {
    if (!cfg || !w || !x_v || !x_t || !temb_silu || !saved || !scratch || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_hy_dual_forward: bad argument");
    return hy_dual_forward(*cfg, *w, (const bf16_t*)x_v, (const bf16_t*)x_t, (const bf16_t*)temb_silu, key_bias, rope_cos, rope_sin, (bf16_t*)out_v, (bf16_t*)out_t,
                           saved, saved_bytes, scratch, scratch_bytes, (hipStream_t)stream);
}
int ftmi_hy_dual_backward(const ftmi_hy_dual_config* cfg, const ftmi_hy_dual_weights* w, const void* x_v, const void* x_t, const void* dout_v, const void* dout_t,
                          const float* key_bias, const float* rope_cos, const float* rope_sin, const void* ones_row, void* dx_v, void* dx_t, float* grad_a,
                          float* grad_b, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, ftmi_stream stream) {
    if (!cfg || !w || !x_v || !x_t || !dout_v || !dout_t || !ones_row || !dx_v || !dx_t || !saved || !scratch || (rope_cos == nullptr) != (rope_sin == nullptr))
        return set_error(FTMI_ERR_INVALID, "ftmi_hy_dual_backward: bad argument");
    return hy_dual_backward(*cfg, *w, (const bf16_t*)x_v, (const bf16_t*)x_t, (const bf16_t*)dout_v, (const bf16_t*)dout_t, key_bias, rope_cos, rope_sin,
                            (const bf16_t*)ones_row, (bf16_t*)dx_v, (bf16_t*)dx_t, grad_a, grad_b, saved, saved_bytes, scratch, scratch_bytes, (hipStream_t)stream);
}

int ftmi_cog_patchify(const void* latents, void* tokens, int B, int F, int C, int H, int W, int patch, ftmi_stream stream) {
    if (!latents || !tokens) return set_error(FTMI_ERR_INVALID, "ftmi_cog_patchify: null argument");
    return cog_patch_permute((const bf16_t*)latents, (bf16_t*)tokens, B, F, C, H, W, patch, 1, (hipStream_t)stream);
}

int ftmi_cog_unpatchify(const void* tokens, void* latents, int B, int F, int C, int H, int W, int patch, ftmi_stream stream) {
    if (!latents || !tokens) return set_error(FTMI_ERR_INVALID, "ftmi_cog_unpatchify: null argument");
    return cog_patch_permute((const bf16_t*)tokens, (bf16_t*)latents, B, F, C, H, W, patch, 0, (hipStream_t)stream);
}

size_t ftmi_cog_workspace_bytes(const ftmi_cog_config* cfg) { return cfg ? cog_workspace_bytes(*cfg) : 0; }

int ftmi_cog_blocks_forward(const ftmi_cog_config* cfg, const ftmi_cog_weights* w, const void* tokens_in, const void* temb_silu, void* tokens_out,
                            void* workspace, size_t workspace_bytes, ftmi_stream stream) {
    if (!cfg || !w || !tokens_in || !temb_silu || !tokens_out || !workspace) return set_error(FTMI_ERR_INVALID, "ftmi_cog_blocks_forward: null argument");
    return cog_blocks_forward(*cfg, *w, (const bf16_t*)tokens_in, (const bf16_t*)temb_silu, (bf16_t*)tokens_out, workspace, workspace_bytes, (hipStream_t)stream);
}

int ftmi_cog_blocks_backward(const ftmi_cog_config* cfg, const ftmi_cog_weights* w, const void* tokens_in, const void* d_tokens_out, void* d_tokens_in,
                             float* grad_a, float* grad_b, void* workspace, size_t workspace_bytes, int l_hi, int l_lo, int accumulate, ftmi_stream stream) {
    if (!cfg || !w || !tokens_in || !d_tokens_out || !workspace || (cfg->r > 0 && (!grad_a || !grad_b)))
        return set_error(FTMI_ERR_INVALID, "ftmi_cog_blocks_backward: null argument");
    return cog_blocks_backward(*cfg, *w, (const bf16_t*)tokens_in, (const bf16_t*)d_tokens_out, (bf16_t*)d_tokens_in, grad_a, grad_b, workspace, workspace_bytes,
                               l_hi, l_lo, accumulate, (hipStream_t)stream);
}

int ftmi_mse_loss(const void* pred, const void* target, const float* weight, float* loss, void* dpred, int B, long per_sample, float grad_scale,
                  float* scratch, ftmi_stream stream) {
    if (!pred || !target || !loss || !scratch) return set_error(FTMI_ERR_INVALID, "ftmi_mse_loss: null argument");
    return mse_loss_fwd_bwd((const bf16_t*)pred, (const bf16_t*)target, weight, loss, (bf16_t*)dpred, B, per_sample, grad_scale, scratch, (hipStream_t)stream);
}

int ftmi_clip_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float max_norm, float lr, float beta1,
                         float beta2, float eps, float weight_decay, int step, float* scratch, float* grad_norm_out, ftmi_stream stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || !scratch) return set_error(FTMI_ERR_INVALID, "ftmi_clip_adamw_step: null argument");
    if (step < 1) return set_error(FTMI_ERR_INVALID, "ftmi_clip_adamw_step: step counts from 1");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, 2 * sizeof(float), st) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "ftmi_clip_adamw_step: memset failed");
    int rc = sumsq(grads, n, scratch, st);
    if (rc) return rc;
    return adamw_clip_step(params, grads, exp_avg, exp_avg_sq, n, scratch, max_norm, lr, beta1, beta2, eps, weight_decay, step, grad_norm_out, st);
}

int ftmi_clip_grad_norm(float* grads, long n, float max_norm, float* scratch, float* grad_norm_out, ftmi_stream stream) {
    if (!grads || !scratch || n <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_clip_grad_norm: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, 2 * sizeof(float), st) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "ftmi_clip_grad_norm: memset failed");
    int rc = sumsq(grads, n, scratch, st);
    if (rc) return rc;
    return clip_scale(grads, n, scratch, max_norm, grad_norm_out, st);
}

int ftmi_grad_sumsq(const float* grads, long n, float* scratch, ftmi_stream stream) {
    if (!grads || !scratch || n <= 0) return set_error(FTMI_ERR_INVALID, "ftmi_grad_sumsq: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, 2 * sizeof(float), st) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "ftmi_grad_sumsq: memset failed");
    return sumsq(grads, n, scratch, st);
}

int ftmi_clip_by_sumsq(float* grads, long n, const float* sumsq, float max_norm, float* grad_norm_out, ftmi_stream stream) {
    if (!grads || !sumsq || n < 0) return set_error(FTMI_ERR_INVALID, "ftmi_clip_by_sumsq: bad argument");
    if (n == 0) return 0;
    return clip_scale(grads, n, sumsq, max_norm, grad_norm_out, (hipStream_t)stream);
}

int ftmi_adamw_bf16_step(void* params, const float* grads, void* exp_avg, void* exp_avg_sq, long n, const float* sumsq_in, float max_norm, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int step, float* grad_norm_out, ftmi_stream stream) {
    if (!params || !grads || !exp_avg || !exp_avg_sq || n < 0 || step < 1) return set_error(FTMI_ERR_INVALID, "ftmi_adamw_bf16_step: bad argument");
    return adamw_bf16_step((bf16_t*)params, grads, (bf16_t*)exp_avg, (bf16_t*)exp_avg_sq, n, sumsq_in, max_norm, lr, beta1, beta2, eps, weight_decay, step,
                           grad_norm_out, (hipStream_t)stream);
}

namespace {
WanRowArgs wan_args(const ftmi_wan_row_args* p) {
    WanRowArgs a;
    a.x = (const bf16_t*)p->x; a.ld_x = p->ld_x; a.w = (const bf16_t*)p->w; a.b = (const bf16_t*)p->b; a.shift = p->shift; a.scale = p->scale;
    a.mod_bstride = p->mod_bstride; a.dy = (const bf16_t*)p->dy; a.ld_dy = p->ld_dy; a.dres = (const bf16_t*)p->dres; a.y = (bf16_t*)p->y; a.ld_y = p->ld_y;
    a.red1 = p->red1; a.red2 = p->red2; a.red_per_batch = p->red_per_batch; a.rope_cos = p->rope_cos; a.rope_sin = p->rope_sin; a.head_dim = p->head_dim;
    a.rows = p->rows; a.D = p->D; a.rows_per_batch = p->rows_per_batch; a.eps = p->eps;
    return a;
}
}  // namespace

#define FTMI_WAN_ENTRY(NAME)                                                                        \
    int ftmi_##NAME(const ftmi_wan_row_args* args, ftmi_stream stream) {                            \
        if (!args) return set_error(FTMI_ERR_INVALID, "ftmi_" #NAME ": null argument block");      \
        return NAME(wan_args(args), (hipStream_t)stream);                                           \
    }
FTMI_WAN_ENTRY(wan_ln_fwd)
FTMI_WAN_ENTRY(wan_ln_bwd)
FTMI_WAN_ENTRY(wan_rms_rope_fwd)
FTMI_WAN_ENTRY(wan_rms_rope_bwd)
FTMI_WAN_ENTRY(wan_gate_res_fwd)
FTMI_WAN_ENTRY(wan_gate_res_bwd)
FTMI_WAN_ENTRY(wan_colsum)
#undef FTMI_WAN_ENTRY

size_t ftmi_wan_block_saved_bytes(const ftmi_wan_block_config* cfg) { return cfg ? wan_block_saved_bytes(*cfg) : 0; }
size_t ftmi_wan_block_scratch_bytes(const ftmi_wan_block_config* cfg) { return cfg ? wan_block_scratch_bytes(*cfg) : 0; }
size_t ftmi_wan_block_param_elements(const ftmi_wan_block_config* cfg) { return cfg ? wan_block_param_elements(*cfg) : 0; }
int ftmi_wan_block_forward(const ftmi_wan_block_config* cfg, const void* params, const void* x, const void* enc, const float* mod, const float* rope_cos,
                           const float* rope_sin, void* out, void* saved, size_t saved_bytes, ftmi_stream stream) {
    if (!cfg || !params || !x || !enc || !mod || !rope_cos || !rope_sin || !out || !saved) return set_error(FTMI_ERR_INVALID, "ftmi_wan_block_forward: null argument");
    return wan_block_forward(*cfg, (const bf16_t*)params, (const bf16_t*)x, (const bf16_t*)enc, mod, rope_cos, rope_sin, (bf16_t*)out, saved, saved_bytes,
                             (hipStream_t)stream);
}
int ftmi_wan_block_backward(const ftmi_wan_block_config* cfg, const void* params, float* grads, const void* x, const void* enc, const float* mod,
                            const float* rope_cos, const float* rope_sin, const void* dout, void* dx, void* denc, float* dmod, void* saved, size_t saved_bytes,
                            void* scratch, size_t scratch_bytes, ftmi_stream stream) {
    if (!cfg || !params || !grads || !x || !enc || !mod || !rope_cos || !rope_sin || !dout || !dx || !denc || !dmod || !saved || !scratch)
        return set_error(FTMI_ERR_INVALID, "ftmi_wan_block_backward: null argument");
    return wan_block_backward(*cfg, (const bf16_t*)params, grads, (const bf16_t*)x, (const bf16_t*)enc, mod, rope_cos, rope_sin, (const bf16_t*)dout, (bf16_t*)dx,
                              (bf16_t*)denc, dmod, saved, saved_bytes, scratch, scratch_bytes, (hipStream_t)stream);
}

int ftmi_lora_refresh(const float* a_f32, const float* b_f32, void* lora_a_sp, void* lora_bt_sp, void* lora_b_ext, void* lora_at_ext,
                      void* lora_at_qkv_ext, int L, int r, int D, ftmi_stream stream) {
    return ftmi_lora_refresh_n(a_f32, b_f32, lora_a_sp, lora_bt_sp, lora_b_ext, lora_at_ext, lora_at_qkv_ext, L, 8, r, D, stream);
}

int ftmi_lora_refresh_n(const float* a_f32, const float* b_f32, void* lora_a_sp, void* lora_bt_sp, void* lora_b_ext, void* lora_at_ext,
                        void* lora_at_qkv_ext, int L, int nadp, int r, int D, ftmi_stream stream) {
    if (!a_f32 || !b_f32 || !lora_a_sp || !lora_bt_sp || !lora_b_ext || !lora_at_ext || !lora_at_qkv_ext || nadp < 3)
        return set_error(FTMI_ERR_INVALID, "ftmi_lora_refresh: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long per = (long)r * D;
    LoraSplitArgs a;  // A [r, D]: row planes of A, column planes of A^T
    a.w = a_f32; a.rows = r; a.cols = D; a.nmat = L * nadp; a.in_bstride = per;
    a.sp = (bf16_t*)lora_a_sp; a.sp_bstride = 2 * per;
    a.t_ext = (bf16_t*)lora_at_ext; a.t_ext_bstride = 3 * per; a.ld_t_ext = 3L * r;
    int rc = lora_split(a, st);
    if (rc) return rc;
    LoraSplitArgs b;  // B [D, r]: column planes of B, row planes of B^T
    b.w = b_f32; b.rows = D; b.cols = r; b.nmat = L * nadp; b.in_bstride = per;
    b.ext = (bf16_t*)lora_b_ext; b.ext_bstride = 3 * per; b.ld_ext = 3L * r;
    b.t_sp = (bf16_t*)lora_bt_sp; b.t_sp_bstride = 2 * per;
    rc = lora_split(b, st);
    if (rc) return rc;
    LoraSplitArgs q;  // adapters 0,1,2 of every block side by side: [D, 9r]
    q.w = a_f32; q.rows = r; q.cols = D; q.nmat = L * 3; q.inner_n = 3; q.in_bstride = (long)nadp * per; q.in_istride = per;
    q.t_ext = (bf16_t*)lora_at_qkv_ext; q.t_ext_bstride = 9 * per; q.t_ext_istride = 3L * r; q.ld_t_ext = 9L * r;
    return lora_split(q, st);
}

int ftmi_lora_split(const float* w, int rows, int cols, void* sp, void* ext, void* t_sp, void* t_ext, ftmi_stream stream) {
    if (!w) return set_error(FTMI_ERR_INVALID, "ftmi_lora_split: null argument");
    LoraSplitArgs a;
    a.w = w; a.rows = rows; a.cols = cols; a.nmat = 1;
    a.sp = (bf16_t*)sp; a.ext = (bf16_t*)ext; a.ld_ext = 3L * cols; a.t_sp = (bf16_t*)t_sp; a.t_ext = (bf16_t*)t_ext; a.ld_t_ext = 3L * rows;
    return lora_split(a, (hipStream_t)stream);
}

}  // extern "C"
