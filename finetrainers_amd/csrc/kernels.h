// Internal (C++) launcher interface of libftmi355.  The public C ABI is include/ftmi355.h; this
// header is what api.hip and the DiT orchestrator (ltx_dit.hip) use to launch the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>

#include "../../include/ftmi355.h"

namespace ftmi {

typedef uint16_t bf16_t;

int set_error(int code, const char* msg);
int check_launch(const char* what);
// tuning switch from the environment (read once per call site: `static const int v = env_int(...)` is thread-safe)
int env_int(const char* name, int dflt);

// A switch that tests flip inside one process (FTMI_ATTN_PL, FTMI_ATTN_FEWKEYS, FTMI_SKINNY4: bit-identity tests compare two kernels through the same
// entry point).  The environment is read ONCE, when the first launch constructs the switch; a launch costs one relaxed atomic load -- no getenv on the launch
// path.  ftmi_reload_switches() (C ABI; tests call it after changing the environment) re-reads every switch constructed so far.  The stand-alone lab
// harnesses (tools/*_lab.hip, -DFTMI_LAB) re-read on every call: they drive their variants with setenv().
class EnvSwitch {
public:
    EnvSwitch(const char* name, int dflt) : name_(name), dflt_(dflt), v_(env_int(name, dflt)) {
        EnvSwitch* head = list().load(std::memory_order_acquire);
        do next_ = head;
        while (!list().compare_exchange_weak(head, this, std::memory_order_acq_rel));
    }
    int get() const {
#ifdef FTMI_LAB
        return env_int(name_, dflt_);
#else
        return v_.load(std::memory_order_relaxed);
#endif
    }
    static int reload_all() {
        int n = 0;
        for (EnvSwitch* s = list().load(std::memory_order_acquire); s; s = s->next_, ++n) s->v_.store(env_int(s->name_, s->dflt_), std::memory_order_relaxed);
        return n;
    }

private:
    static std::atomic<EnvSwitch*>& list() {
        static std::atomic<EnvSwitch*> head{nullptr};
        return head;
    }
    const char* name_;
    int dflt_;
    std::atomic<int> v_;
    EnvSwitch* next_ = nullptr;
};

enum { EPI_STORE = 0, EPI_GELU = 1, EPI_RESID = 2, EPI_DGELU = 3 };

// ---- in-stream HIP-event profiler (bench.py's live roofline numbers) ----
enum { PROF_GEMM_NT = 0, PROF_GEMM_TN = 1, PROF_ATTN_FWD = 2, PROF_ATTN_BWD = 3, PROF_GEMM_SKINNY = 4, PROF_NCLASS = 5 };
bool prof_enabled();
bool prof_begin(int kclass, double flops, hipStream_t st);  // true if this launch is sampled (events recorded)
void prof_end(int kclass, hipStream_t st);
struct ProfScope {
    int k;
    hipStream_t st;
    bool on;
    ProfScope(int kclass, double flops, hipStream_t s) : k(kclass), st(s), on(false) {
        if (prof_enabled()) on = prof_begin(k, flops, st);
    }
    ~ProfScope() {
        if (on) prof_end(k, st);
    }
};

struct GemmNtArgs {
    const bf16_t* gate2 = nullptr;  // EPI_RESID with out2: out2 = bf(out * gate2[b])  (the consumer's gate multiply, fused)
    long gate2_bstride = 0;
    const bf16_t* X = nullptr;  // [M, ldx]
    long ldx = 0;
    const bf16_t* W = nullptr;  // [N, ldw]   (K-contiguous rows)
    long ldw = 0;
    int M = 0, N = 0, K = 0;
    int xk_grp_n = 0;  // X column offset = (n0 / xk_grp_n) * xk_grp_stride   (grouped dXA of fused q,k,v)
    long xk_grp_stride = 0;
    // W rows live in groups of w_grp_n consecutive rows, w_grp_stride elements apart (0 = plain [N, ldw]); same for W2.
    // Lets one launch span the per-block slices of a stacked [L, 8, ...] LoRA tensor (all blocks' text-side K/V projections).
    int w_grp_n = 0;
    long w_grp_stride = 0;
    int w2_grp_n = 0;
    long w2_grp_stride = 0;
    // K-extension (fused LoRA up-projection): acc += X2 . W2^T after the base result was rounded to bf16
    const bf16_t* X2 = nullptr;
    long ldx2 = 0;
    const bf16_t* W2 = nullptr;
    long ldw2 = 0;
    int K2 = 0;
    int x2_grp_n = 0;  // X2 column offset = (n0 / x2_grp_n) * x2_grp_stride   (fused q,k,v: one XA slice per projection)
    long x2_grp_stride = 0;
    // epilogue
    const bf16_t* bias = nullptr;  // [N]
    float alpha = 1.f;
    bf16_t* out = nullptr;
    long ldo = 0;
    bf16_t* out2 = nullptr;  // EPI_GELU: pre-activation z
    long ldo2 = 0;
    const bf16_t* resid = nullptr;  // EPI_RESID
    long ldr = 0;
    const bf16_t* gate = nullptr;  // EPI_RESID: gate[b * gate_bstride + n], b = m / rows_per_batch
    long gate_bstride = 0;
    int rows_per_batch = 0;
    const bf16_t* aux = nullptr;  // EPI_DGELU: z
    long ldaux = 0;
    int epi = EPI_STORE;
    int variant = 0;  // 0 = register-staged tiles, 1 = direct global->LDS (BK64), 2/3 = BK32 direct-to-LDS
    // fp32-equivalent LoRA down-projection (skinny kernel only).  split_r > 0: W holds 2N rows -- the bf16 hi and lo planes of an
    // fp32 matrix interleaved in groups of 32 rows ([hi 0-31 | lo 0-31 | hi 32-63 | ...], N counts BOTH planes) -- the two partial
    // products are added in fp32 and the fp32 result t is written as three bf16 column planes per group of split_r outputs:
    // out[:, 3 split_r g + {0, split_r, 2 split_r} + i] = (hi(t), lo(t), hi(t)), the X2 operand layout of the K-extension.
    int split_r = 0;
    // tile -> XCD rasterisation (filled by the launcher): the 8 XCDs form a map_gm x map_gn grid, each owning a
    // map_rm x map_rn rectangle of output tiles, so the tiles resident on one XCD share few operand panels in its L2
    int map_gm = 1, map_gn = 8, map_rm = 0, map_rn = 0;
#ifdef FTMI_TRACE
    // FTMI_TRACE=1 builds only (libftmi355_trace.so, tools/nt_trace.py): per-workgroup phase stamps, 16 x u64 per workgroup -- see NT_STAMP in gemm.hip
    unsigned long long* trace = nullptr;
#endif
};
int gemm_nt(const GemmNtArgs& a, hipStream_t st);
// Round 6: a LoRA down-projection (split hi/lo mode: `down.out` is `gemm.X2`) and the GEMM that consumes it through its K-extension as ONE launch -- the
// down-projection's workgroups lead the grid, the GEMM's tiles wait (once, two K stages before their extension) on per-row-tile counters in `flags`
// (ints, >= ceil(M / 64) of them, owned by the caller: zero them once, then pass `*expect` back in for every following launch on the same flags).
// Falls back to the two launches (down-projection, then GEMM) wherever the pair is not eligible or FTMI_FUSE_DOWN=0.
int gemm_nt_lora_fused(const GemmNtArgs& gemm, const GemmNtArgs& down, int* flags, int* expect, hipStream_t st);
int gemm_fused_status();  // 1 if a poll inside a fused launch ever gave up (synchronises the device: tests only)
int gemm_nt_plan(int M, int N, int K, int K2, int epi);  // the automatic kernel choice as a pure host function (tests)
// persistent 256 x 256 stream-K form of the same contract (gemm_sk.hip); gemm_nt() routes eligible launches to it
bool gemm_nt_sk_eligible(const GemmNtArgs& a);
int gemm_nt_sk(const GemmNtArgs& a, hipStream_t st);
int gemm_nt_sk_status();
int gemm_nt_sk_trace(unsigned long long* out, int cap);
// LoRA down-projection (split hi/lo mode), third generation (gemm_skinny.hip, FTMI_EXPERIMENTAL builds only): 64 x 128 tiles, K cut across
// workgroups when the tiles alone would leave most CUs idle.  Measured slower than gemm_nt_skinny2_kernel; selected with FTMI_SKINNY3=1.
bool gemm_nt_skinny3_eligible(const GemmNtArgs& a);
int gemm_nt_skinny3_slices(int M, int N, int K);
int gemm_nt_skinny3(const GemmNtArgs& a, hipStream_t st);
bool sk_build_work(int ntiles, int G, int nk, int ov, int minp, int pc, int ac, int* work);

struct GemmTnArgs {
    const bf16_t* U = nullptr;  // [M, ldu], P columns used
    long ldu = 0;
    const bf16_t* V = nullptr;  // [M, ldv], Q columns used
    long ldv = 0;
    float* C = nullptr;  // [P, ldc] fp32, accumulated with atomics
    long ldc = 0;
    int M = 0, P = 0, Q = 0;
    int v_grp_p = 0;  // V column offset = (p0 / v_grp_p) * v_grp_stride
    long v_grp_stride = 0;
    int u_grp_p = 0;  // U columns live in groups: column of p = (p / u_grp_p) * u_grp_stride + p % u_grp_p  (0 = contiguous)
    long u_grp_stride = 0;
    // fp32-equivalent operands: an operand given as bf16 (hi, lo) column planes u_fold / v_fold elements apart contributes
    // hi^T.other + lo^T.other to the same C (at most one of the two may be folded)
    long u_fold = 0, v_fold = 0;
    int msteps_per_split = 0;  // filled by the launcher
    int xcd_groups = 0;        // filled by the launcher: > 0 = 1-D grid in the XCD-aware order of gemm_tn2_kernel (number of (split, batch) groups)
    float scale = 1.f;
    // batched form (blockIdx.y = batch index): element strides between consecutive problems (0 = shared operand)
    int batch = 1;
    long u_bstride = 0, v_bstride = 0, c_bstride = 0;
};
int gemm_tn(const GemmTnArgs& a, hipStream_t st);
// fp8 (e4m3fn) weight storage: exact up-cast to bf16, plain [rows, cols] or transposed [cols, rows] (cast.hip)
int fp8_upcast(const uint8_t* src, bf16_t* dst, int rows, int cols, int transpose, hipStream_t st);

// ---- attention -----------------------------------------------------------------------------------
struct AttnArgs {
    int B = 0, H = 0, Sq = 0, Sk = 0;
    int d = 64;  // head dim: 64 (forward + backward) or 128 (forward)
    // element strides (d contiguous)
    const bf16_t* q = nullptr;
    long q_sb = 0, q_sh = 0, q_ss = 0;
    const bf16_t* k = nullptr;
    long k_sb = 0, k_sh = 0, k_ss = 0;
    const bf16_t* v = nullptr;
    long v_sb = 0, v_sh = 0, v_ss = 0;
    bf16_t* o = nullptr;  // forward output / backward input
    long o_sb = 0, o_sh = 0, o_ss = 0;
    float* lse2 = nullptr;        // [B, H, Sq] log2-domain log-sum-exp of the scaled scores
    const float* kbias = nullptr;  // additive key bias (natural units), may be null: element (b, h, j) at kbias[b * kb_sb + h * kb_sh + j]
    long kb_sb = 0, kb_sh = 0;     // (kb_sh = 0: one bias row per sample, shared by the heads -- LTX's text mask)
    float scale = 0.125f;
    // backward only
    const bf16_t* dout = nullptr;
    long do_sb = 0, do_sh = 0, do_ss = 0;
    float* delta = nullptr;  // [B, H, Sq]
    bf16_t* dq = nullptr;
    long dq_sb = 0, dq_sh = 0, dq_ss = 0;
    bf16_t* dk = nullptr;
    long dk_sb = 0, dk_sh = 0, dk_ss = 0;
    bf16_t* dv = nullptr;
    long dv_sb = 0, dv_sh = 0, dv_ss = 0;
};
int attn_fwd(const AttnArgs& a, hipStream_t st);
int attn_bwd(const AttnArgs& a, hipStream_t st);  // delta pre-pass + dK/dV kernel + dQ kernel

// ---- row-wise / elementwise ------------------------------------------------------------------------
// ada[l][b][slot][D]: slots 0..5 = table[i] + temb[b][i] (shift_msa, scale_msa, gate_msa, shift_mlp,
// scale_mlp, gate_mlp), 6 = 1 + scale_msa, 7 = 1 + scale_mlp (all rounded to bf16 like the eager graph)
int ada_prep(const bf16_t* tables, const bf16_t* temb, bf16_t* ada, int L, int B, int D, hipStream_t st);
// out table for the final norm: ada_out[b][slot][D]: 0 = shift, 1 = scale, 2 = 1 + scale
int ada_out_prep(const bf16_t* table2, const bf16_t* emb, bf16_t* ada_out, int B, int D, hipStream_t st);

// y = bf(bf(norm(x)) * onep[b]) + shift[b]   (norm = RMS (no affine) or LayerNorm (no affine))
void rowwise_set_valid_width(int dv);  // zero-padded narrow rows: over how many channels norm_modulate / qknorm_rope take their mean (0 = all; thread-local)
int norm_modulate_fwd(const bf16_t* x, const bf16_t* shift, const bf16_t* onep, long mod_bstride, bf16_t* y, int rows,
                      int rows_per_batch, int D, float eps, int layernorm, hipStream_t st);
// dx_out = (dres ? dres : 0) + norm_bwd(x, bf(dy * onep[b]))
int norm_modulate_bwd(const bf16_t* x, const bf16_t* dy, const bf16_t* onep, long mod_bstride, const bf16_t* dres,
                      bf16_t* dx, int rows, int rows_per_batch, int D, float eps, int layernorm, hipStream_t st,
                      const bf16_t* gate2 = nullptr, long gate2_bstride = 0, bf16_t* dx2 = nullptr);  // dx2 = bf(dx * gate2[b]) (optional)

// affine RMSNorm over the full width (+ optional interleaved-pair RoPE), x row stride ldx
// w_rows > 1: row i uses weight row (i % w_rows) of a [w_rows, D] table (rows interleaved over blocks)
int qknorm_rope_fwd(const bf16_t* x, long ldx, const bf16_t* w, const float* cos_t, const float* sin_t, bf16_t* y, long ldy,
                    int rows, int rows_per_batch, int D, float eps, hipStream_t st, int w_rows = 1, const bf16_t* x2 = nullptr, const bf16_t* w2 = nullptr,
                    bf16_t* y2 = nullptr);  // x2 / w2 / y2: a second tensor set with the same strides processed by the same launch (q and k)
int qknorm_rope_bwd(const bf16_t* x, long ldx, const bf16_t* w, const float* cos_t, const float* sin_t, const bf16_t* dy,
                    long lddy, bf16_t* dx, long lddx, int rows, int rows_per_batch, int D, float eps, hipStream_t st, int w_rows = 1, const bf16_t* x2 = nullptr,
                    const bf16_t* w2 = nullptr, const bf16_t* dy2 = nullptr, bf16_t* dx2 = nullptr, int row_grp = 0, int row_grp_span = 0);

// latents [B,C,F*H*W] bf16 -> x_t, target packed [B, S, C]
int noise_pack(const bf16_t* latents, const bf16_t* noise, const float* mean, const float* std_, const float* sigma,
               const float* sigma_first, int first_frame_tokens, bf16_t* xt, bf16_t* target, int B, int C, int S,
               hipStream_t st);

// loss (fp32 scalar, accumulated) and dpred
int mse_loss_fwd_bwd(const bf16_t* pred, const bf16_t* target, const float* weight, float* loss, bf16_t* dpred, int B,
                     long per_sample, float grad_scale, float* partials, hipStream_t st);

// CogVideoX DDIM noising (mode 0: x0 = bf(a * scale), out = bf(sa x0) + bf(so b)) / get_velocity (mode 1: out = bf(sa b) - bf(so a))
int ddim_mix(const bf16_t* a, const bf16_t* b, const float* sa, const float* so, float scale, bf16_t* x0_out, bf16_t* out, int B, long per_sample,
             int mode, hipStream_t st);
// x = mean + exp(0.5 * clamp(logvar)) * eps from moments [B][2][half] (mean | logvar), bf16 op by op
int posterior_sample(const bf16_t* moments, const bf16_t* eps, bf16_t* out, int B, long half, hipStream_t st);

// timestep sinusoid (256 channels, flip_sin_to_cos) of t = float(timestep)
int timestep_sinusoid(const float* tval, bf16_t* out, int B, hipStream_t st);
// y[r][n] = act_out(sum_k act_in(x[r][k]) W[n][k] + bias[n]);  rows <= 8
int small_linear(const bf16_t* x, const bf16_t* W, const bf16_t* bias, bf16_t* y, int rows, int N, int K, int silu_in,
                 int silu_out, hipStream_t st);

// ---- optimiser ---------------------------------------------------------------------------------------
int sumsq(const float* g, long n, float* out /* [0] result, [1] ticket (both zeroed by the caller), [2..2049] block partials */, hipStream_t st);
int clip_scale(float* g, long n, const float* sumsq_in, float max_norm, float* grad_norm_out, hipStream_t st);  // in-place clip by a precomputed sum of squares
int adamw_clip_step(float* p, const float* g, float* m, float* v, long n, const float* sumsq_in, float max_norm, float lr,
                    float beta1, float beta2, float eps, float wd, int step, float* grad_norm_out, hipStream_t st);
// bf16 (hi, lo) working copies of nmat fp32 matrices W [rows, cols] (hi = bf16(w), lo = bf16(w - hi); hi + lo carries 16 mantissa bits):
//   sp  [2 rows, cols]   hi / lo planes of W interleaved in groups of 32 rows (operand of the split skinny GEMM)      (may be null)
//   ext [rows, ld_ext]   [hi | hi | lo] along the columns, written at column offset ext_col0 (K-extension operand)   (may be null)
//   t_sp [2 cols, rows], t_ext [cols, ld_t_ext]: the same two layouts of W^T                                            (may be null)
struct LoraSplitArgs {
    const float* w = nullptr;
    int rows = 0, cols = 0, nmat = 0;
    long in_bstride = 0;
    bf16_t* sp = nullptr;    long sp_bstride = 0;
    bf16_t* ext = nullptr;   long ext_bstride = 0, ld_ext = 0;
    bf16_t* t_sp = nullptr;  long t_sp_bstride = 0;
    bf16_t* t_ext = nullptr; long t_ext_bstride = 0, ld_t_ext = 0;
    // matrices are numbered m = outer * inner_n + inner (inner_n = 0: flat): *_bstride applies per outer, *_istride per inner
    int inner_n = 0;
    long in_istride = 0, t_ext_istride = 0;
};
int lora_split(const LoraSplitArgs& a, hipStream_t st);
// plain bf16 transpose [rows, cols] -> [cols, rows]
int transpose_bf16(const bf16_t* in, bf16_t* out, int rows, int cols, hipStream_t st);

// ---- CogVideoX row-wise kernels (cogvideox.hip): any row width D % 64 == 0, D <= 4096 ----------------------------------------------
// Tokens [B, rows_per_batch, D]; seg0 > 0: the first seg0 tokens of a sample (text) use modulation row 2b, the others (video) row 2b + 1 of
// [B, 2, D] tables; seg0 == 0: one row per sample, tables [B, D].
struct CogLnArgs {
    const bf16_t* x = nullptr;      // input rows (head_ln: row stride ld)
    const bf16_t* w = nullptr;      // LayerNorm weight [D] (head_ln: [64])
    const bf16_t* b = nullptr;      // LayerNorm bias
    const bf16_t* shift = nullptr;  // modulation shift
    const bf16_t* onep = nullptr;   // modulation (1 + scale), or the gate of gate_residual
    const bf16_t* dy = nullptr;
    const bf16_t* dres = nullptr;   // ln_mod_bwd: gradient on the residual branch; gate_residual: the residual input
    const float* cos = nullptr;     // head_ln: rotary tables fp32 [S, 64] for the rows at position >= seg0 of a sample (null: no rotary embedding)
    const float* sin = nullptr;
    bf16_t* y = nullptr;
    bf16_t* dx = nullptr;
    int rows = 0, D = 0, rows_per_batch = 1, seg0 = 0;
    long ld = 0;                 // head_ln: row stride of x
    long ld_dy = 0, ld_out = 0;  // head_ln: row strides of dy and of the output (y / dx); 0 = ld
    float eps = 1e-5f;
    int head_dim = 64;           // head_ln: 64 (LayerNorm per head, CogVideoX) or 128 with rms (RMSNorm per head, HunyuanVideo); rotary tables [S, head_dim]
    int rms = 0;
};
int cog_ln_mod_fwd(const CogLnArgs& a, hipStream_t st);    // y = bf(bf(LN(x; w, b)) * onep) + shift
int cog_ln_mod_bwd(const CogLnArgs& a, hipStream_t st);    // dx = [dres +] LN'(x)[bf(dy * onep) * w]
int cog_head_ln_fwd(const CogLnArgs& a, hipStream_t st);   // per 64-channel head: y = LN(x; w[64], b[64])
int cog_head_ln_bwd(const CogLnArgs& a, hipStream_t st);
int cog_gate_residual(const CogLnArgs& a, hipStream_t st); // y = [dres +] bf(onep * x)
size_t cog_workspace_bytes(const ftmi_cog_config& c);
int cog_blocks_forward(const ftmi_cog_config& c, const ftmi_cog_weights& w, const bf16_t* tokens_in, const bf16_t* temb_silu, bf16_t* tokens_out, void* ws,
                       size_t ws_bytes, hipStream_t st);
int cog_blocks_backward(const ftmi_cog_config& c, const ftmi_cog_weights& w, const bf16_t* tokens_in, const bf16_t* d_out, bf16_t* d_in, float* grad_a,
                        float* grad_b, void* ws, size_t ws_bytes, int l_hi, int l_lo, int accumulate, hipStream_t st);
int cog_mod_tables(const bf16_t* mod, bf16_t* tables, int L2, int B, int D, hipStream_t st);  // linear(silu(temb)) rows -> (shift, 1 + scale, gate) x (text, video)
// ---- Wan-T2V row-wise kernels (wan.hip): one argument block for the seven launchers --------------------------------------------------------
struct WanRowArgs {
    const bf16_t* x = nullptr;   // input rows [rows, ld_x]   (gate_res_bwd: d out)
    long ld_x = 0;
    const bf16_t* w = nullptr;   // LayerNorm / RMSNorm weight [D] (LayerNorm: null = no affine)
    const bf16_t* b = nullptr;   // LayerNorm bias [D]
    const float* shift = nullptr;  // fp32 [B, mod_bstride]: modulation shift (null = not modulated)
    const float* scale = nullptr;  // fp32: modulation scale (ln) / gate (gate_res)
    long mod_bstride = 0;
    const bf16_t* dy = nullptr;  // backward: gradient of the output rows; gate_res: the y operand
    long ld_dy = 0;
    const bf16_t* dres = nullptr;  // ln_bwd: gradient arriving on the residual branch (row stride ld_y), added in bf16
    bf16_t* y = nullptr;         // output rows (forward: y, backward: dx / dy)
    long ld_y = 0;
    float* red1 = nullptr;       // column sums, += (see wan.hip)
    float* red2 = nullptr;
    int red_per_batch = 0;       // 1: one row of sums per sample ([B, D]), 0: one row for all
    const float* rope_cos = nullptr;  // fp32 [rows_per_batch, head_dim / 2]
    const float* rope_sin = nullptr;
    int head_dim = 0;
    int rows = 0, D = 0, rows_per_batch = 0;
    float eps = 1e-6f;
};
int wan_ln_fwd(const WanRowArgs& a, hipStream_t st);
int wan_ln_bwd(const WanRowArgs& a, hipStream_t st);
int wan_rms_rope_fwd(const WanRowArgs& a, hipStream_t st);
int wan_rms_rope_bwd(const WanRowArgs& a, hipStream_t st);
int wan_gate_res_fwd(const WanRowArgs& a, hipStream_t st);
int wan_gate_res_bwd(const WanRowArgs& a, hipStream_t st);
int wan_colsum(const WanRowArgs& a, hipStream_t st);
// Wan block orchestrator (wan_dit.hip)
size_t wan_block_saved_bytes(const ftmi_wan_block_config& c);
size_t wan_block_scratch_bytes(const ftmi_wan_block_config& c);
size_t wan_block_param_elements(const ftmi_wan_block_config& c);
int wan_block_forward(const ftmi_wan_block_config& c, const bf16_t* params, const bf16_t* x, const bf16_t* enc, const float* mod, const float* rope_cos,
                      const float* rope_sin, bf16_t* out, void* saved, size_t saved_bytes, hipStream_t st);
int wan_block_backward(const ftmi_wan_block_config& c, const bf16_t* params, float* grads, const bf16_t* x, const bf16_t* enc, const float* mod,
                       const float* rope_cos, const float* rope_sin, const bf16_t* dout, bf16_t* dx, bf16_t* denc, float* dmod, void* saved,
                       size_t saved_bytes, void* scratch, size_t scratch_bytes, hipStream_t st);
int adamw_bf16_step(bf16_t* p, const float* g, bf16_t* m, bf16_t* v, long n, const float* sumsq_in, float max_norm, float lr, float beta1, float beta2,
                    float eps, float wd, int step, float* grad_norm_out, hipStream_t st);

// HunyuanVideo single-stream block orchestrator (hy_dit.hip)
size_t hy_single_saved_bytes(const ftmi_hy_single_config& c);
size_t hy_single_scratch_bytes(const ftmi_hy_single_config& c);
int hy_single_forward(const ftmi_hy_single_config& c, const ftmi_hy_single_weights& w, const bf16_t* x, const bf16_t* temb_silu, const float* key_bias,
                      const float* rope_cos, const float* rope_sin, bf16_t* out, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                      hipStream_t st);
int hy_single_backward(const ftmi_hy_single_config& c, const ftmi_hy_single_weights& w, const bf16_t* x, const bf16_t* dout, const float* key_bias,
                       const float* rope_cos, const float* rope_sin, const bf16_t* ones_rows, bf16_t* dx, float* grad_a, float* grad_b, void* saved,
                       size_t saved_bytes, void* scratch, size_t scratch_bytes, hipStream_t st);

size_t hy_dual_saved_bytes(const ftmi_hy_dual_config& c);
size_t hy_dual_scratch_bytes(const ftmi_hy_dual_config& c);
int hy_dual_forward(const ftmi_hy_dual_config& c, const ftmi_hy_dual_weights& w, const bf16_t* x_v, const bf16_t* x_t, const bf16_t* temb_silu, const float* key_bias,
                    const float* rope_cos, const float* rope_sin, bf16_t* out_v, bf16_t* out_t, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                    hipStream_t st);
int hy_dual_backward(const ftmi_hy_dual_config& c, const ftmi_hy_dual_weights& w, const bf16_t* x_v, const bf16_t* x_t, const bf16_t* dout_v, const bf16_t* dout_t,
                     const float* key_bias, const float* rope_cos, const float* rope_sin, const bf16_t* ones_row, bf16_t* dx_v, bf16_t* dx_t, float* grad_a,
                     float* grad_b, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, hipStream_t st);

int cog_patch_permute(const bf16_t* src, bf16_t* dst, int B, int F, int C, int H, int W, int p, int to_tokens, hipStream_t st);  // latents <-> patch tokens

}  // namespace ftmi
