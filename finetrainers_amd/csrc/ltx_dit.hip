// LTX-Video DiT forward / backward orchestrator: one C call launches the whole 28-block forward (or
// backward) on the caller's stream -- O(1) host work per step instead of the reference's ~3-4k eager
// launches.  All activations the backward needs are stashed in the caller-provided workspace (about
// 0.4 GB per block at B=2, S=2688: trivial against 288 GB of HBM, so there is no activation recompute).
//
// Restates, op for op: finetrainers/patches/models/ltx_video/patch.py:38-127 (model-level forward) and
// the upstream LTXVideoTransformerBlock it iterates, with the peft LoRA branches of
// to_q/to_k/to_v/to_out.0 fused into the projections, and the autograd backward of the same graph with
// frozen base weights (dgrad only) and trainable LoRA A/B.
#include <string.h>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

struct WsLayout {
    size_t total = 0;
    size_t tsin, t1, emb, temb, ada, ada_out, cap_h, e, hs;
    // text-side K/V of the cross-attention for ALL blocks (the text stream `e` is block-independent): one launch each
    size_t kv2_all, k2n_all, xa_kv2_all, g_kv2_all, g_k2n_all, dxa_kv2_all;
    size_t blk0, blk_stride;
    // per-block (offsets relative to the block base)
    size_t n1, qkv, qrot, krot, o1, lse1, xa_qkv, xa_o, h1, q2raw, q2n, o2, lse2, xa_q2, xa_o2, h2, z;
    // per-block backward stash: the dY / dXA operands of the LoRA weight gradients, consumed by batched launches at the end
    size_t g_o2, g_q2, g_o, g_qkv, dxa_o2, dxa_q2, dxa_o, dxa_qkv;
    // scratch
    size_t s_n2, s_g, s_ln, s_dh0, s_dh1, s_d1, s_d2, s_d3, s_dqr, s_dkr, s_dbig, s_delta;
    size_t sk_flags;
    size_t blk_slot;  // bytes of one block slot (blk_stride is 0 under gradient checkpointing: every block uses the same slot)
};

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

WsLayout make_layout(const ftmi_ltx_config& c) {
    WsLayout w;
    const size_t M = (size_t)c.B * c.S, Mt = (size_t)c.B * c.T, D = c.D, r = c.r > 0 ? c.r : 64;
    const size_t e2 = 2;  // bf16
    Bump g;
    w.tsin = g.take((size_t)c.B * 256 * e2);
    w.t1 = g.take((size_t)c.B * D * e2);
    w.emb = g.take((size_t)c.B * D * e2);
    w.temb = g.take((size_t)c.B * 6 * D * e2);
    w.ada = g.take((size_t)c.L * c.B * 8 * D * e2);
    w.ada_out = g.take((size_t)c.B * 3 * D * e2);
    w.cap_h = g.take(Mt * D * e2);
    w.e = g.take(Mt * D * e2);
    w.hs = g.take((size_t)(c.L + 1) * M * D * e2);
    w.kv2_all = g.take(Mt * (size_t)c.L * 2 * D * e2);      // [Mt][L*2D]  (k | v per block)
    w.k2n_all = g.take(Mt * (size_t)c.L * D * e2);          // [Mt][L][D]
    w.xa_kv2_all = g.take(Mt * (size_t)c.L * 2 * 3 * r * e2);   // [Mt][L][k|v][hi|lo|hi][r]
    w.g_kv2_all = g.take(Mt * (size_t)c.L * 2 * D * e2);
    w.g_k2n_all = g.take(Mt * (size_t)c.L * D * e2);
    w.dxa_kv2_all = g.take(Mt * (size_t)c.L * 2 * 3 * r * e2);
    w.sk_flags = g.take(4096);  // 1024 int counters: one per 64-row tile of the token dimension (fused down-projection + GEMM launches, gemm_nt_lora_fused)
    Bump b;
    w.n1 = b.take(M * D * e2);
    w.qkv = b.take(M * 3 * D * e2);
    w.qrot = b.take(M * D * e2);
    w.krot = b.take(M * D * e2);
    w.o1 = b.take(M * D * e2);
    w.lse1 = b.take((size_t)c.B * c.H * c.S * 4);
    w.xa_qkv = b.take(M * 9 * r * e2);  // [M][q|k|v][hi|lo|hi][r]: fp32-equivalent s * x A^T as bf16 planes
    w.xa_o = b.take(M * 3 * r * e2);
    w.h1 = b.take(M * D * e2);
    w.q2raw = b.take(M * D * e2);
    w.q2n = b.take(M * D * e2);
    w.o2 = b.take(M * D * e2);
    w.lse2 = b.take((size_t)c.B * c.H * c.S * 4);
    w.xa_q2 = b.take(M * 3 * r * e2);
    w.xa_o2 = b.take(M * 3 * r * e2);
    w.h2 = b.take(M * D * e2);
    w.z = b.take(M * (size_t)c.D_ff * e2);
    w.g_o2 = b.take(M * D * e2);
    w.g_q2 = b.take(M * D * e2);
    w.g_o = b.take(M * D * e2);
    w.g_qkv = b.take(M * 3 * D * e2);
    w.dxa_o2 = b.take(M * 3 * r * e2);
    w.dxa_q2 = b.take(M * 3 * r * e2);
    w.dxa_o = b.take(M * 3 * r * e2);
    w.dxa_qkv = b.take(M * 9 * r * e2);
    // gradient checkpointing (cfg.checkpoint, the reference's --gradient_checkpointing: utils/activation_checkpoint.py:24-49 wraps every block): ONE block slot
    // instead of L -- a block keeps only its input (the residual stream hs[l], kept for every block either way) and its forward runs again inside its backward
    w.blk_slot = b.off;
    w.blk_stride = c.checkpoint ? 0 : b.off;
    w.blk0 = g.take(w.blk_slot * (c.checkpoint ? 1 : c.L));
    w.s_n2 = g.take(M * D * e2);
    w.s_g = g.take(M * (size_t)c.D_ff * e2);
    w.s_ln = g.take(M * D * e2);
    w.s_dh0 = g.take(M * D * e2);
    w.s_dh1 = g.take(M * D * e2);
    w.s_d1 = g.take(M * D * e2);
    w.s_d2 = g.take(M * D * e2);
    w.s_d3 = g.take(M * D * e2);
    w.s_dqr = g.take(M * D * e2);
    w.s_dkr = g.take(M * D * e2);
    w.s_dbig = g.take(M * (size_t)c.D_ff * e2);
    w.s_delta = g.take((size_t)c.B * c.H * c.S * 4);
    w.total = g.off;
    return w;
}

int check_cfg(const ftmi_ltx_config& c) {
    if (c.B <= 0 || c.S <= 0 || c.T <= 0 || c.L <= 0) return set_error(FTMI_ERR_INVALID, "ltx: empty problem");
    if (c.D != 2048 || c.H * 64 != c.D) return set_error(FTMI_ERR_UNSUPPORTED, "ltx: kernels are built for width 2048 = 32 heads x 64");
    if (c.B > 8) return set_error(FTMI_ERR_UNSUPPORTED, "ltx: per-rank batch must be <= 8 (timestep-embedding kernel)");
    if (c.r < 0 || (c.r % 64) != 0) return set_error(FTMI_ERR_UNSUPPORTED, "ltx: LoRA rank must be 0 or a multiple of 64");
    if ((c.C_in % 64) || (c.C_out % 64) || (c.D_ff % 128) || (c.D_cap % 64))
        return set_error(FTMI_ERR_UNSUPPORTED, "ltx: channel counts must be multiples of 64");
    if (c.d_valid < 0 || c.d_valid > c.D || c.head_dim_valid < 0 || c.head_dim_valid > 64)
        return set_error(FTMI_ERR_INVALID, "ltx: d_valid / head_dim_valid describe a model narrower than the layout (0 = full width)");
    return 0;
}

inline const bf16_t* P(const void* base, size_t elem_off) { return reinterpret_cast<const bf16_t*>(base) + elem_off; }
inline bf16_t* W(void* ws, size_t byte_off) { return reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ws) + byte_off); }
inline float* WF(void* ws, size_t byte_off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + byte_off); }

#define FTMI_TRY(x)            \
    do {                       \
        int _rc = (x);         \
        if (_rc) return _rc;   \
    } while (0)

// plain linear helper
int linear(const bf16_t* X, long ldx, int M, const bf16_t* Wt, long ldw, int N, int K, const bf16_t* bias, bf16_t* out, long ldo,
           int variant, hipStream_t st, float alpha = 1.f) {
    GemmNtArgs a;
    a.X = X; a.ldx = ldx; a.W = Wt; a.ldw = ldw; a.M = M; a.N = N; a.K = K;
    a.bias = bias; a.alpha = alpha; a.out = out; a.ldo = ldo; a.variant = variant;
    return gemm_nt(a, st);
}

// LoRA down-projection at fp32-equivalent precision: t = alpha * X . Wf^T for an fp32 matrix Wf given as interleaved bf16 (hi, lo) row
// planes `w_sp` ([2 nout, K], kernels.h LoraSplitArgs), t kept as bf16 planes (hi | lo | hi) per group of r outputs: out [M, 3 nout].
// The reference runs this product in fp32 (trainer/sft_trainer/trainer.py:132-136 casts the LoRA parameters to fp32).
GemmNtArgs lora_down_args(const bf16_t* X, long ldx, int M, const bf16_t* w_sp, int nout, int K, int r, float alpha, bf16_t* out, int xk_grp_stride = 0) {
    GemmNtArgs a;
    a.X = X; a.ldx = ldx; a.W = w_sp; a.ldw = K; a.M = M; a.N = 2 * nout; a.K = K; a.alpha = alpha;
    if (xk_grp_stride > 0) { a.xk_grp_n = 2 * r; a.xk_grp_stride = xk_grp_stride; }  // output group g (one adapter) reads X columns g * stride ...
    a.split_r = r; a.out = out; a.ldo = 3L * nout; a.variant = 8;
    return a;
}
int lora_down(const bf16_t* X, long ldx, int M, const bf16_t* w_sp, int nout, int K, int r, float alpha, bf16_t* out, hipStream_t st,
              int xk_grp_stride = 0) {
    return gemm_nt(lora_down_args(X, ldx, M, w_sp, nout, K, r, alpha, out, xk_grp_stride), st);
}
// A projection with its LoRA: the down-projection `dn` (writes a.X2) and the GEMM `a` (K-extension over X2) -- one fused launch where the pair is eligible
// (gemm_nt_lora_fused: FTMI_FUSE_DOWN=1), else the two launches of rounds 1-5.  `fx` = the workspace's row-tile counters + the running expectation of this call.
struct FuseCtx { int* flags; int expect; };
int lora_gemm(const GemmNtArgs& a, const GemmNtArgs& dn, FuseCtx& fx, hipStream_t st) {
    return gemm_nt_lora_fused(a, dn, (a.M + 63) / 64 <= 1024 ? fx.flags : nullptr, &fx.expect, st);
}

AttnArgs attn_args(const ftmi_ltx_config& c, int Sq, int Sk) {
    AttnArgs a;
    a.B = c.B; a.H = c.H; a.Sq = Sq; a.Sk = Sk;
    a.scale = (c.head_dim_valid > 0 && c.head_dim_valid < 64) ? 1.0f / sqrtf((float)c.head_dim_valid) : 0.125f;  // (a zero-padded narrow head: the scale of its true width)
    return a;
}
inline void set3(long& sb, long& sh, long& ss, long rows_per_batch, long ld) {
    sb = rows_per_batch * ld;
    sh = 64;
    ss = ld;
}

}  // namespace

size_t ltx_workspace_bytes(const ftmi_ltx_config& c) { return make_layout(c).total; }

int ltx_workspace_offset(const ftmi_ltx_config& c, const char* name, int layer, size_t* off) {
    const WsLayout L = make_layout(c);
    const size_t M = (size_t)c.B * c.S;
    struct E { const char* n; size_t o; };
    const E globals[] = {{"kv2_all", L.kv2_all}, {"k2n_all", L.k2n_all}, {"xa_kv2_all", L.xa_kv2_all}, {"e", L.e}, {"emb", L.emb}, {"temb", L.temb}, {"ada", L.ada}, {"ada_out", L.ada_out}, {"tsin", L.tsin}};
    for (const E& g : globals)
        if (!strcmp(name, g.n)) { *off = g.o; return 0; }
    if (!strcmp(name, "hs")) {
        if (layer < 0 || layer > c.L) return set_error(FTMI_ERR_INVALID, "workspace_offset: layer out of range");
        *off = L.hs + (size_t)layer * M * c.D * 2;
        return 0;
    }
    const E blocks[] = {{"n1", L.n1}, {"qkv", L.qkv}, {"qrot", L.qrot}, {"krot", L.krot}, {"o1", L.o1}, {"lse1", L.lse1},
                        {"xa_qkv", L.xa_qkv}, {"xa_o", L.xa_o}, {"h1", L.h1}, {"q2raw", L.q2raw}, {"q2n", L.q2n},
                        {"o2", L.o2}, {"lse2", L.lse2}, {"xa_q2", L.xa_q2},
                        {"xa_o2", L.xa_o2}, {"h2", L.h2}, {"z", L.z}};
    if (layer < 0 || layer >= c.L) return set_error(FTMI_ERR_INVALID, "workspace_offset: layer out of range");
    for (const E& b : blocks)
        if (!strcmp(name, b.n)) { *off = L.blk0 + L.blk_stride * layer + b.o; return 0; }
    return set_error(FTMI_ERR_INVALID, "workspace_offset: unknown name");
}

// One transformer block of the forward (steps 1-13 of patches/models/ltx_video/patch.py:82-123 + the upstream block): reads hs[l], writes hs[l+1] and the
// block's activations into its slot.  Called by the forward for every block and -- under gradient checkpointing -- again by the backward right before a
// block's gradient computation (deterministic kernels: the recomputed activations are the forward's, bit for bit).
static int ltx_block_forward(const ftmi_ltx_config& c, const ftmi_ltx_weights& w, const WsLayout& L, void* ws, int l, const float* key_bias, hipStream_t st, FuseCtx& fx) {
    const int M = c.B * c.S, D = c.D, r = c.r, V = c.gemm_variant;
    const long D2 = (long)D * D;
    const float s = c.lora_scale;
    {
        char* blk = reinterpret_cast<char*>(ws) + L.blk0 + L.blk_stride * l;
        const bf16_t* h0 = W(ws, L.hs) + (size_t)l * M * D;
        bf16_t* hout = W(ws, L.hs) + (size_t)(l + 1) * M * D;
        const bf16_t* ada = W(ws, L.ada) + (size_t)l * c.B * 8 * D;  // [B][8][D]
        const long ab = 8L * D;
        const bf16_t* la = w.lora_a_sp ? P(w.lora_a_sp, (size_t)l * 8 * 2 * r * D) : nullptr;   // [8][2r][D]  (hi, lo) planes of A
        const bf16_t* lb = w.lora_b_ext ? P(w.lora_b_ext, (size_t)l * 8 * D * 3 * r) : nullptr;  // [8][D][3r]  [B_hi | B_hi | B_lo]
        bf16_t* n1 = W(blk, L.n1);
        bf16_t* qkv = W(blk, L.qkv);

        // 1. norm1 + AdaLN modulate
        FTMI_TRY(norm_modulate_fwd(h0, ada + 0 * D, ada + 6 * D, ab, n1, M, c.S, D, c.eps_norm, 0, st));
        // 2-3. fused q,k,v projection (+ LoRA)
        {
            GemmNtArgs a;
            a.X = n1; a.ldx = D; a.W = P(w.w_qkv, (size_t)l * 3 * D2); a.ldw = D; a.M = M; a.N = 3 * D; a.K = D;
            a.bias = P(w.b_qkv, (size_t)l * 3 * D); a.out = qkv; a.ldo = 3 * D; a.variant = V;
            GemmNtArgs dn;
            if (r > 0) {
                dn = lora_down_args(n1, D, M, la, 3 * r, D, r, s, W(blk, L.xa_qkv));
                a.X2 = W(blk, L.xa_qkv); a.ldx2 = 9 * r; a.W2 = lb; a.ldw2 = 3 * r; a.K2 = 3 * r; a.x2_grp_n = D; a.x2_grp_stride = 3 * r;
            }
            FTMI_TRY(r > 0 ? lora_gemm(a, dn, fx, st) : gemm_nt(a, st));
        }
        // 4. QK RMSNorm across heads + RoPE
        FTMI_TRY(qknorm_rope_fwd(qkv, 3 * D, P(w.norm_q, (size_t)l * D), w.rope_cos, w.rope_sin, W(blk, L.qrot), D, M, c.S, D, c.eps_qk, st, 1,
                                 qkv + D, P(w.norm_k, (size_t)l * D), W(blk, L.krot)));  // q and k in one launch
        // 5. self-attention
        {
            AttnArgs a = attn_args(c, c.S, c.S);
            a.q = W(blk, L.qrot); set3(a.q_sb, a.q_sh, a.q_ss, c.S, D);
            a.k = W(blk, L.krot); set3(a.k_sb, a.k_sh, a.k_ss, c.S, D);
            a.v = qkv + 2 * D;    set3(a.v_sb, a.v_sh, a.v_ss, c.S, 3 * D);
            a.o = W(blk, L.o1);   set3(a.o_sb, a.o_sh, a.o_ss, c.S, D);
            a.lse2 = WF(blk, L.lse1);
            FTMI_TRY(attn_fwd(a, st));
        }
        // 6. to_out (+ LoRA), gate * residual
        {
            GemmNtArgs a;
            a.X = W(blk, L.o1); a.ldx = D; a.W = P(w.w_o, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D;
            a.bias = P(w.b_o, (size_t)l * D); a.out = W(blk, L.h1); a.ldo = D; a.variant = V;
            a.epi = EPI_RESID; a.resid = h0; a.ldr = D; a.gate = ada + 2 * D; a.gate_bstride = ab; a.rows_per_batch = c.S;
            GemmNtArgs dn;
            if (r > 0) {
                dn = lora_down_args(W(blk, L.o1), D, M, la + 3L * 2 * r * D, r, D, r, s, W(blk, L.xa_o));
                a.X2 = W(blk, L.xa_o); a.ldx2 = 3 * r; a.W2 = lb + 3L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r;
            }
            FTMI_TRY(r > 0 ? lora_gemm(a, dn, fx, st) : gemm_nt(a, st));
        }
        const bf16_t* h1 = W(blk, L.h1);
        // 7. cross-attention query (no pre-norm, no RoPE)
        {
            GemmNtArgs a;
            a.X = h1; a.ldx = D; a.W = P(w.w_q2, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D;
            a.bias = P(w.b_q2, (size_t)l * D); a.out = W(blk, L.q2raw); a.ldo = D; a.variant = V;
            GemmNtArgs dn;
            if (r > 0) {
                dn = lora_down_args(h1, D, M, la + 4L * 2 * r * D, r, D, r, s, W(blk, L.xa_q2));
                a.X2 = W(blk, L.xa_q2); a.ldx2 = 3 * r; a.W2 = lb + 4L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r;
            }
            FTMI_TRY(r > 0 ? lora_gemm(a, dn, fx, st) : gemm_nt(a, st));
            FTMI_TRY(qknorm_rope_fwd(W(blk, L.q2raw), D, P(w.norm_q2, (size_t)l * D), nullptr, nullptr, W(blk, L.q2n), D, M, c.S, D, c.eps_qk, st));
        }
        // 9. cross-attention with the text-mask bias
        {
            AttnArgs a = attn_args(c, c.S, c.T);
            a.q = W(blk, L.q2n);          set3(a.q_sb, a.q_sh, a.q_ss, c.S, D);
            a.k = W(ws, L.k2n_all) + (size_t)l * D;              set3(a.k_sb, a.k_sh, a.k_ss, c.T, (long)c.L * D);
            a.v = W(ws, L.kv2_all) + (size_t)l * 2 * D + D;      set3(a.v_sb, a.v_sh, a.v_ss, c.T, (long)c.L * 2 * D);
            a.o = W(blk, L.o2);           set3(a.o_sb, a.o_sh, a.o_ss, c.S, D);
            a.lse2 = WF(blk, L.lse2);
            a.kbias = key_bias; a.kb_sb = c.T;
            FTMI_TRY(attn_fwd(a, st));
        }
        // 10. to_out (+ LoRA), residual (no gate)
        {
            GemmNtArgs a;
            a.X = W(blk, L.o2); a.ldx = D; a.W = P(w.w_o2, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D;
            a.bias = P(w.b_o2, (size_t)l * D); a.out = W(blk, L.h2); a.ldo = D; a.variant = V;
            a.epi = EPI_RESID; a.resid = h1; a.ldr = D;
            GemmNtArgs dn;
            if (r > 0) {
                dn = lora_down_args(W(blk, L.o2), D, M, la + 7L * 2 * r * D, r, D, r, s, W(blk, L.xa_o2));
                a.X2 = W(blk, L.xa_o2); a.ldx2 = 3 * r; a.W2 = lb + 7L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r;
            }
            FTMI_TRY(r > 0 ? lora_gemm(a, dn, fx, st) : gemm_nt(a, st));
        }
        const bf16_t* h2 = W(blk, L.h2);
        // 11-13. norm2 + modulate, feed-forward, gate * residual
        FTMI_TRY(norm_modulate_fwd(h2, ada + 3 * D, ada + 7 * D, ab, W(ws, L.s_n2), M, c.S, D, c.eps_norm, 0, st));
        {
            GemmNtArgs a;
            a.X = W(ws, L.s_n2); a.ldx = D; a.W = P(w.w_ff1, (size_t)l * c.D_ff * D); a.ldw = D; a.M = M; a.N = c.D_ff; a.K = D;
            a.bias = P(w.b_ff1, (size_t)l * c.D_ff); a.out = W(ws, L.s_g); a.ldo = c.D_ff; a.out2 = W(blk, L.z); a.ldo2 = c.D_ff;
            a.epi = EPI_GELU; a.variant = V;
            FTMI_TRY(gemm_nt(a, st));
        }
        {
            GemmNtArgs a;
            a.X = W(ws, L.s_g); a.ldx = c.D_ff; a.W = P(w.w_ff2, (size_t)l * D * c.D_ff); a.ldw = c.D_ff; a.M = M; a.N = D; a.K = c.D_ff;
            a.bias = P(w.b_ff2, (size_t)l * D); a.out = hout; a.ldo = D; a.variant = V;
            a.epi = EPI_RESID; a.resid = h2; a.ldr = D; a.gate = ada + 5 * D; a.gate_bstride = ab; a.rows_per_batch = c.S;
            FTMI_TRY(gemm_nt(a, st));
        }
    }
    return 0;
}

int ltx_forward(const ftmi_ltx_config& c, const ftmi_ltx_weights& w, const bf16_t* x_t, const bf16_t* text, const float* key_bias,
                const float* sigma, bf16_t* pred, void* ws, size_t ws_bytes, hipStream_t st) {
    FTMI_TRY(check_cfg(c));
    struct ValidWidth {  // (normalisations of a zero-padded narrow model take their mean over d_valid channels; reset when the pass has queued its launches)
        explicit ValidWidth(int dv) { rowwise_set_valid_width(dv); }
        ~ValidWidth() { rowwise_set_valid_width(0); }
    } valid_width_guard(c.d_valid);
    const WsLayout L = make_layout(c);
    if (ws_bytes < L.total) return set_error(FTMI_ERR_INVALID, "ltx_forward: workspace too small");
    const int M = c.B * c.S, Mt = c.B * c.T, D = c.D, r = c.r, V = c.gemm_variant;
    const long D2 = (long)D * D;
    const float s = c.lora_scale;

    // ---- conditioning (one row per sample: every token of a sample shares its timestep) ----
    FTMI_TRY(timestep_sinusoid(sigma, W(ws, L.tsin), c.B, st));
    FTMI_TRY(small_linear(W(ws, L.tsin), P(w.time_l1_w, 0), P(w.time_l1_b, 0), W(ws, L.t1), c.B, D, 256, 0, 0, st));
    FTMI_TRY(small_linear(W(ws, L.t1), P(w.time_l2_w, 0), P(w.time_l2_b, 0), W(ws, L.emb), c.B, D, D, 1, 0, st));
    FTMI_TRY(small_linear(W(ws, L.emb), P(w.time_lin_w, 0), P(w.time_lin_b, 0), W(ws, L.temb), c.B, 6 * D, D, 1, 0, st));
    FTMI_TRY(ada_prep(P(w.tables, 0), W(ws, L.temb), W(ws, L.ada), c.L, c.B, D, st));
    FTMI_TRY(ada_out_prep(P(w.table_out, 0), W(ws, L.emb), W(ws, L.ada_out), c.B, D, st));

    // ---- proj_in, caption projection ----
    FTMI_TRY(linear(x_t, c.C_in, M, P(w.proj_in_w, 0), c.C_in, D, c.C_in, P(w.proj_in_b, 0), W(ws, L.hs), D, V, st));
    {
        GemmNtArgs a;
        a.X = text; a.ldx = c.D_cap; a.W = P(w.cap_l1_w, 0); a.ldw = c.D_cap; a.M = Mt; a.N = D; a.K = c.D_cap;
        a.bias = P(w.cap_l1_b, 0); a.out = W(ws, L.cap_h); a.ldo = D; a.epi = EPI_GELU; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
        FTMI_TRY(linear(W(ws, L.cap_h), D, Mt, P(w.cap_l2_w, 0), D, D, D, P(w.cap_l2_b, 0), W(ws, L.e), D, V, st));
    }
    const bf16_t* e = W(ws, L.e);

    // ---- cross-attention keys/values of EVERY block (the text stream does not change across blocks) ----
    {
        const long ldkv = (long)c.L * 2 * D;
        GemmNtArgs a;
        a.X = e; a.ldx = D; a.W = P(w.w_kv2, 0); a.ldw = D; a.M = Mt; a.N = c.L * 2 * D; a.K = D;
        a.bias = P(w.b_kv2, 0); a.out = W(ws, L.kv2_all); a.ldo = ldkv; a.variant = V;
        if (r > 0) {
            GemmNtArgs x;  // XA[:, (l,k|v)] = s * e A_{l,k|v}^T : adapters 5,6 of block l are 2 * 2r consecutive plane rows, blocks 8 * 2r * D apart
            x.X = e; x.ldx = D; x.W = P(w.lora_a_sp, 5L * 2 * r * D); x.ldw = D; x.w_grp_n = 4 * r; x.w_grp_stride = 16L * r * D;
            x.M = Mt; x.N = c.L * 4 * r; x.K = D; x.alpha = s; x.split_r = r; x.out = W(ws, L.xa_kv2_all); x.ldo = (long)c.L * 6 * r; x.variant = V;
            FTMI_TRY(gemm_nt(x, st));
            a.X2 = W(ws, L.xa_kv2_all); a.ldx2 = (long)c.L * 6 * r; a.x2_grp_n = D; a.x2_grp_stride = 3 * r; a.K2 = 3 * r;
            a.W2 = P(w.lora_b_ext, 5L * D * 3 * r); a.ldw2 = 3 * r; a.w2_grp_n = 2 * D; a.w2_grp_stride = 8L * D * 3 * r;
        }
        FTMI_TRY(gemm_nt(a, st));
        // k2 = norm_k(k2raw): rows ordered (token, block): row i = t * L + l reads kv2_all + i * 2D, weight row l
        FTMI_TRY(qknorm_rope_fwd(W(ws, L.kv2_all), 2 * D, P(w.norm_k2, 0), nullptr, nullptr, W(ws, L.k2n_all), D, Mt * c.L, Mt * c.L, D,
                                 c.eps_qk, st, c.L));
    }

    FuseCtx fx{reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + L.sk_flags), 0};
    if (hipMemsetAsync(fx.flags, 0, 4096, st) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "ltx_forward: memset of the row-tile counters failed");
    for (int l = 0; l < c.L; ++l) FTMI_TRY(ltx_block_forward(c, w, L, ws, l, key_bias, st, fx));

    // ---- tail: LayerNorm + modulate + proj_out ----
    const bf16_t* hL = W(ws, L.hs) + (size_t)c.L * M * D;
    const bf16_t* ao = W(ws, L.ada_out);
    FTMI_TRY(norm_modulate_fwd(hL, ao, ao + 2 * D, 3L * D, W(ws, L.s_ln), M, c.S, D, c.eps_norm, 1, st));
    FTMI_TRY(linear(W(ws, L.s_ln), D, M, P(w.proj_out_w, 0), D, c.C_out, D, P(w.proj_out_b, 0), pred, c.C_out, V, st));
    return 0;
}

// Backward of blocks [l_lo, l_hi) in descending order (the tail first when l_hi == L).  When the call returns (stream order) the LoRA
// gradients of exactly those blocks are FINAL -- all 8 adapters, the text-side ones included -- so a data-parallel caller can start
// the gradient exchange of that block range while the next call computes the blocks below it (DDP's bucketed overlap,
// finetrainers/parallel/ptd.py:462-463, without a reducer).  State between calls lives in the workspace; l_hi..l_lo must tile L..0.
int ltx_backward_range(const ftmi_ltx_config& c, const ftmi_ltx_weights& w, const bf16_t* text, const float* key_bias, const bf16_t* dpred,
                       float* grad_a, float* grad_b, void* ws, size_t ws_bytes, int l_hi, int l_lo, int accumulate, hipStream_t st) {
    (void)text;
    FTMI_TRY(check_cfg(c));
    struct ValidWidth {  // (normalisations of a zero-padded narrow model take their mean over d_valid channels; reset when the pass has queued its launches)
        explicit ValidWidth(int dv) { rowwise_set_valid_width(dv); }
        ~ValidWidth() { rowwise_set_valid_width(0); }
    } valid_width_guard(c.d_valid);
    const WsLayout L = make_layout(c);
    if (ws_bytes < L.total) return set_error(FTMI_ERR_INVALID, "ltx_backward: workspace too small");
    if (l_lo < 0 || l_hi > c.L || l_lo >= l_hi) return set_error(FTMI_ERR_INVALID, "ltx_backward: bad block range");
    if (c.r > 0 && !accumulate && l_hi == c.L) {  // .grad was None: the weight-gradient kernels accumulate (split-token atomics), so start from zero
        const size_t nbytes = (size_t)c.L * 8 * c.r * c.D * sizeof(float);
        if (hipMemsetAsync(grad_a, 0, nbytes, st) != hipSuccess || hipMemsetAsync(grad_b, 0, nbytes, st) != hipSuccess)
            return set_error(FTMI_ERR_LAUNCH, "ltx_backward: memset of the gradient buffer failed");
    }
    const int M = c.B * c.S, Mt = c.B * c.T, D = c.D, r = c.r, V = c.gemm_variant;
    const long D2 = (long)D * D;
    const float s = c.lora_scale;
    const bf16_t* e = W(ws, L.e);

    bf16_t* dh[2] = {W(ws, L.s_dh0), W(ws, L.s_dh1)};
    bf16_t* d1 = W(ws, L.s_d1);
    bf16_t* d2 = W(ws, L.s_d2);
    bf16_t* dO = W(ws, L.s_d3);

    // ---- tail ----
    if (l_hi == c.L) {
        FTMI_TRY(linear(dpred, c.C_out, M, P(w.proj_out_w_t, 0), c.C_out, D, c.C_out, nullptr, d1, D, V, st));
        const bf16_t* hL = W(ws, L.hs) + (size_t)c.L * M * D;
        const bf16_t* ao = W(ws, L.ada_out);
        // the gated copy bf(dh * gate_mlp) that opens the last block's backward is written by the same kernel (into dO, idle here)
        const bf16_t* ada_last = W(ws, L.ada) + (size_t)(c.L - 1) * c.B * 8 * D;
        FTMI_TRY(norm_modulate_bwd(hL, d1, ao + 2 * D, 3L * D, nullptr, dh[0], M, c.S, D, c.eps_norm, 1, st, ada_last + 5 * D, 8L * D, dO));
    }
    int cur = (c.L - l_hi) & 1;  // the gradient of the residual stream ping-pongs between two buffers, one flip per finished block

    // LoRA adapter backward, critical-path half: dXA = s * dY B (needed at once by the dgrad K-extension).  The weight
    // gradients dB += dY^T XA and dA += dXA^T X only feed the gradient buffer, so dY / dXA are kept per block and all 28
    // blocks of one adapter are reduced by ONE batched launch after the loop (fills the GPU instead of 28 latency-bound ones).
    auto lora_dxa = [&](const bf16_t* dY, long lddy, int rows, int nadp, int adp, int l, bf16_t* dxa_out) -> GemmNtArgs {
        const bf16_t* lbt = P(w.lora_bt_sp, ((size_t)l * 8 + adp) * 2 * r * D);  // [nadp * 2r][D]: (hi, lo) planes of B^T
        return lora_down_args(dY, lddy, rows, lbt, nadp * r, D, r, s, dxa_out, nadp > 1 ? D : 0);
    };
    FuseCtx fx{reinterpret_cast<int*>(reinterpret_cast<char*>(ws) + L.sk_flags), 0};  // row-tile counters of the fused down-projection + GEMM launches of this call
    if (r > 0 && hipMemsetAsync(fx.flags, 0, 4096, st) != hipSuccess) return set_error(FTMI_ERR_LAUNCH, "ltx_backward: memset of the row-tile counters failed");

    // LoRA weight gradients dB += dY^T XA, dA += dXA^T X only feed the gradient buffer: ONE batched launch per adapter group over all
    // blocks of this call's range, after its block loop (fills the GPU instead of 28 latency-bound launches per adapter).
    auto lora_wgrad = [&](int l0, int nb, hipStream_t s2) -> int {
        char* blk0 = reinterpret_cast<char*>(ws) + L.blk0 + L.blk_stride * l0;
        const long bs = (long)(L.blk_stride / 2);  // block stride in bf16 elements
        struct G { size_t dy; long lddy; int rows, nadp, adp; size_t xa, dxa; const bf16_t* x; long ldx, x_bs; };
        const G groups[4] = {
            {L.g_o2, D, M, 1, 7, L.xa_o2, L.dxa_o2, W(blk0, L.o2), D, bs},
            {L.g_q2, D, M, 1, 4, L.xa_q2, L.dxa_q2, W(blk0, L.h1), D, bs},
            {L.g_o, D, M, 1, 3, L.xa_o, L.dxa_o, W(blk0, L.o1), D, bs},
            {L.g_qkv, 3L * D, M, 3, 0, L.xa_qkv, L.dxa_qkv, W(blk0, L.n1), D, bs},
        };
        float* ga = grad_a + (size_t)l0 * 8 * r * D;
        float* gb = grad_b + (size_t)l0 * 8 * D * r;
        for (const G& gr : groups) {
            GemmTnArgs t;  // dB[l] += dY[l]^T XA[l]
            t.U = W(blk0, gr.dy); t.ldu = gr.lddy; t.V = W(blk0, gr.xa); t.ldv = (long)gr.nadp * 3 * r; t.v_fold = r;  // XA = hi + lo planes
            t.C = gb + (size_t)gr.adp * D * r; t.ldc = r; t.M = gr.rows; t.P = gr.nadp * D; t.Q = r;
            if (gr.nadp > 1) { t.v_grp_p = D; t.v_grp_stride = 3 * r; }
            t.batch = nb; t.u_bstride = bs; t.v_bstride = bs; t.c_bstride = 8L * D * r;
            FTMI_TRY(gemm_tn(t, s2));
            GemmTnArgs u;  // dA[l] += dXA[l]^T X[l]
            u.U = W(blk0, gr.dxa); u.ldu = (long)gr.nadp * 3 * r; u.u_fold = r; u.V = gr.x; u.ldv = gr.ldx;
            if (gr.nadp > 1) { u.u_grp_p = r; u.u_grp_stride = 3 * r; }
            u.C = ga + (size_t)gr.adp * r * D; u.ldc = D; u.M = gr.rows; u.P = gr.nadp * r; u.Q = D;
            u.batch = nb; u.u_bstride = bs; u.v_bstride = gr.x_bs; u.c_bstride = 8L * r * D;
            FTMI_TRY(gemm_tn(u, s2));
        }
        return 0;
    };
    for (int l = l_hi - 1; l >= l_lo; --l) {
        char* blk = reinterpret_cast<char*>(ws) + L.blk0 + L.blk_stride * l;
        const bf16_t* h0 = W(ws, L.hs) + (size_t)l * M * D;
        const bf16_t* ada = W(ws, L.ada) + (size_t)l * c.B * 8 * D;
        const long ab = 8L * D;
        const bf16_t* lat = w.lora_at_ext ? P(w.lora_at_ext, (size_t)l * 8 * D * 3 * r) : nullptr;  // [8][D][3r]  [A^T_hi | A^T_hi | A^T_lo]
        const bf16_t* h1 = W(blk, L.h1);
        const bf16_t* h2 = W(blk, L.h2);
        const bf16_t* dhin = dh[cur];
        if (c.checkpoint) FTMI_TRY(ltx_block_forward(c, w, L, ws, l, key_bias, st, fx));  // the block's activations again, into the one slot

        // ---- feed-forward ----
        // (dO holds bf(dhin * gate_mlp): written by the kernel that produced dhin)
        {
            GemmNtArgs a;
            a.X = dO; a.ldx = D; a.W = P(w.w_ff2_t, (size_t)l * c.D_ff * D); a.ldw = D; a.M = M; a.N = c.D_ff; a.K = D;
            a.out = W(ws, L.s_dbig); a.ldo = c.D_ff; a.epi = EPI_DGELU; a.aux = W(blk, L.z); a.ldaux = c.D_ff; a.variant = V;
            FTMI_TRY(gemm_nt(a, st));
        }
        FTMI_TRY(linear(W(ws, L.s_dbig), c.D_ff, M, P(w.w_ff1_t, (size_t)l * D * c.D_ff), c.D_ff, D, c.D_ff, nullptr, d2, D, V, st));
        bf16_t* d3 = W(blk, L.g_o2);  // dh2: also the dY of attn2.to_out
        FTMI_TRY(norm_modulate_bwd(h2, d2, ada + 7 * D, ab, dhin, d3, M, c.S, D, c.eps_norm, 0, st));

        // ---- cross-attention ----
        {
            GemmNtArgs a;
            a.X = d3; a.ldx = D; a.W = P(w.w_o2_t, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = d1; a.ldo = D; a.variant = V;
            if (r > 0) { a.X2 = W(blk, L.dxa_o2); a.ldx2 = 3 * r; a.W2 = lat + 7L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r; }
            FTMI_TRY(r > 0 ? lora_gemm(a, lora_dxa(d3, D, M, 1, 7, l, W(blk, L.dxa_o2)), fx, st) : gemm_nt(a, st));  // d1 = dO2
        }
        {
            AttnArgs a = attn_args(c, c.S, c.T);
            a.q = W(blk, L.q2n);          set3(a.q_sb, a.q_sh, a.q_ss, c.S, D);
            a.k = W(ws, L.k2n_all) + (size_t)l * D;              set3(a.k_sb, a.k_sh, a.k_ss, c.T, (long)c.L * D);
            a.v = W(ws, L.kv2_all) + (size_t)l * 2 * D + D;      set3(a.v_sb, a.v_sh, a.v_ss, c.T, (long)c.L * 2 * D);
            a.o = W(blk, L.o2);           set3(a.o_sb, a.o_sh, a.o_ss, c.S, D);
            a.lse2 = WF(blk, L.lse2); a.kbias = key_bias; a.kb_sb = c.T;
            a.dout = d1;                  set3(a.do_sb, a.do_sh, a.do_ss, c.S, D);
            a.delta = WF(ws, L.s_delta);
            a.dq = d2;                    set3(a.dq_sb, a.dq_sh, a.dq_ss, c.S, D);
            a.dk = W(ws, L.g_k2n_all) + (size_t)l * D;            set3(a.dk_sb, a.dk_sh, a.dk_ss, c.T, (long)c.L * D);
            a.dv = W(ws, L.g_kv2_all) + (size_t)l * 2 * D + D;    set3(a.dv_sb, a.dv_sh, a.dv_ss, c.T, (long)c.L * 2 * D);
            FTMI_TRY(attn_bwd(a, st));
        }
        bf16_t* gq2 = W(blk, L.g_q2);  // dq2raw
        FTMI_TRY(qknorm_rope_bwd(W(blk, L.q2raw), D, P(w.norm_q2, (size_t)l * D), nullptr, nullptr, d2, D, gq2, D, M, c.S, D, c.eps_qk, st));
        {
            GemmNtArgs a;
            a.X = gq2; a.ldx = D; a.W = P(w.w_q2_t, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = d2; a.ldo = D; a.variant = V;
            a.epi = EPI_RESID; a.resid = d3; a.ldr = D;
            a.out2 = W(blk, L.g_o); a.ldo2 = D; a.gate2 = ada + 2 * D; a.gate2_bstride = ab; a.rows_per_batch = c.S;  // go = bf(dh1 * gate_msa), fused
            if (r > 0) { a.X2 = W(blk, L.dxa_q2); a.ldx2 = 3 * r; a.W2 = lat + 4L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r; }
            FTMI_TRY(r > 0 ? lora_gemm(a, lora_dxa(gq2, D, M, 1, 4, l, W(blk, L.dxa_q2)), fx, st) : gemm_nt(a, st));  // d2 = dh1
        }

        // ---- self-attention ----
        bf16_t* go = W(blk, L.g_o);  // d(attn1.to_out output)
        {
            GemmNtArgs a;
            a.X = go; a.ldx = D; a.W = P(w.w_o_t, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = dO; a.ldo = D; a.variant = V;
            if (r > 0) { a.X2 = W(blk, L.dxa_o); a.ldx2 = 3 * r; a.W2 = lat + 3L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r; }
            FTMI_TRY(r > 0 ? lora_gemm(a, lora_dxa(go, D, M, 1, 3, l, W(blk, L.dxa_o)), fx, st) : gemm_nt(a, st));
        }
        bf16_t* dqkv = W(blk, L.g_qkv);
        const bf16_t* qkv = W(blk, L.qkv);
        {
            AttnArgs a = attn_args(c, c.S, c.S);
            a.q = W(blk, L.qrot); set3(a.q_sb, a.q_sh, a.q_ss, c.S, D);
            a.k = W(blk, L.krot); set3(a.k_sb, a.k_sh, a.k_ss, c.S, D);
            a.v = qkv + 2 * D;    set3(a.v_sb, a.v_sh, a.v_ss, c.S, 3 * D);
            a.o = W(blk, L.o1);   set3(a.o_sb, a.o_sh, a.o_ss, c.S, D);
            a.lse2 = WF(blk, L.lse1);
            a.dout = dO;          set3(a.do_sb, a.do_sh, a.do_ss, c.S, D);
            a.delta = WF(ws, L.s_delta);
            a.dq = W(ws, L.s_dqr); set3(a.dq_sb, a.dq_sh, a.dq_ss, c.S, D);
            a.dk = W(ws, L.s_dkr); set3(a.dk_sb, a.dk_sh, a.dk_ss, c.S, D);
            a.dv = dqkv + 2 * D;   set3(a.dv_sb, a.dv_sh, a.dv_ss, c.S, 3 * D);
            FTMI_TRY(attn_bwd(a, st));
        }
        FTMI_TRY(qknorm_rope_bwd(qkv, 3 * D, P(w.norm_q, (size_t)l * D), w.rope_cos, w.rope_sin, W(ws, L.s_dqr), D, dqkv, 3 * D, M, c.S, D, c.eps_qk, st, 1,
                                 qkv + D, P(w.norm_k, (size_t)l * D), W(ws, L.s_dkr), dqkv + D));  // q and k in one launch
        if (r > 0 && l == 0) FTMI_TRY(gemm_nt(lora_dxa(dqkv, 3 * D, M, 3, 0, l, W(blk, L.dxa_qkv)), st));  // (block 0 has no input gradient: the dXA alone, for dA)
        if (l > 0) {
            GemmNtArgs a;
            a.X = dqkv; a.ldx = 3 * D; a.W = P(w.w_qkv_t, (size_t)l * 3 * D2); a.ldw = 3 * D; a.M = M; a.N = D; a.K = 3 * D; a.out = d1; a.ldo = D; a.variant = V;
            if (r > 0) { a.X2 = W(blk, L.dxa_qkv); a.ldx2 = 9 * r; a.W2 = P(w.lora_at_qkv_ext, (size_t)l * D * 9 * r); a.ldw2 = 9 * r; a.K2 = 9 * r; }
            FTMI_TRY(r > 0 ? lora_gemm(a, lora_dxa(dqkv, 3 * D, M, 3, 0, l, W(blk, L.dxa_qkv)), fx, st) : gemm_nt(a, st));  // d1 = dn1
            const bf16_t* ada_prev = W(ws, L.ada) + (size_t)(l - 1) * c.B * 8 * D;  // the next block processed is l - 1
            FTMI_TRY(norm_modulate_bwd(h0, d1, ada + 6 * D, ab, d2, dh[cur ^ 1], M, c.S, D, c.eps_norm, 0, st, ada_prev + 5 * D, ab, dO));
            cur ^= 1;
        }
        if (c.checkpoint && r > 0) FTMI_TRY(lora_wgrad(l, 1, st));  // the slot is about to be reused: this block's weight gradients now
    }
    if (r > 0 && !c.checkpoint) FTMI_TRY(lora_wgrad(l_lo, l_hi - l_lo, st));
    const int nb = l_hi - l_lo;

    // ---- text side of the cross-attention, all blocks at once (nothing upstream of `e` needs a gradient) ----
    if (r > 0) {
        // d(k2raw) = RMSNorm backward of d(k2n); rows ordered (token, block) like the forward
        // (row i of this call = (token i / nb, block l_lo + i % nb): rows of one token are L apart in the all-block arrays)
        FTMI_TRY(qknorm_rope_bwd(W(ws, L.kv2_all) + (size_t)l_lo * 2 * D, 2 * D, P(w.norm_k2, (size_t)l_lo * D), nullptr, nullptr,
                                 W(ws, L.g_k2n_all) + (size_t)l_lo * D, D, W(ws, L.g_kv2_all) + (size_t)l_lo * 2 * D, 2 * D,
                                 Mt * nb, Mt * nb, D, c.eps_qk, st, nb, nullptr, nullptr, nullptr, nullptr, nb, c.L));
        GemmNtArgs a;  // dXA[:, (l,k|v)] = s * dY[:, (l,k|v) slice] B_{l,k|v}
        a.X = W(ws, L.g_kv2_all) + (size_t)l_lo * 2 * D; a.ldx = (long)c.L * 2 * D; a.xk_grp_n = 2 * r; a.xk_grp_stride = D;
        a.W = P(w.lora_bt_sp, ((size_t)l_lo * 8 + 5) * 2 * r * D); a.ldw = D; a.w_grp_n = 4 * r; a.w_grp_stride = 16L * r * D;
        a.M = Mt; a.N = nb * 4 * r; a.K = D; a.alpha = s; a.split_r = r; a.out = W(ws, L.dxa_kv2_all) + (size_t)l_lo * 6 * r; a.ldo = (long)c.L * 6 * r; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
    }

    // ---- LoRA weight gradients of the text-side adapters (all blocks in one launch each) ----
    if (r > 0) {
        {   // attn2.to_k / to_v: operands are column slices of the all-block arrays (batch stride = one block's columns)
            GemmTnArgs t;
            t.U = W(ws, L.g_kv2_all) + (size_t)l_lo * 2 * D; t.ldu = (long)c.L * 2 * D; t.V = W(ws, L.xa_kv2_all) + (size_t)l_lo * 6 * r; t.ldv = (long)c.L * 6 * r; t.v_fold = r;
            t.C = grad_b + ((size_t)l_lo * 8 + 5) * D * r; t.ldc = r; t.M = Mt; t.P = 2 * D; t.Q = r; t.v_grp_p = D; t.v_grp_stride = 3 * r;
            t.batch = nb; t.u_bstride = 2L * D; t.v_bstride = 6L * r; t.c_bstride = 8L * D * r;
            FTMI_TRY(gemm_tn(t, st));
            GemmTnArgs u;
            u.U = W(ws, L.dxa_kv2_all) + (size_t)l_lo * 6 * r; u.ldu = (long)c.L * 6 * r; u.u_fold = r; u.u_grp_p = r; u.u_grp_stride = 3 * r; u.V = e; u.ldv = D;
            u.C = grad_a + ((size_t)l_lo * 8 + 5) * r * D; u.ldc = D; u.M = Mt; u.P = 2 * r; u.Q = D;
            u.batch = nb; u.u_bstride = 6L * r; u.v_bstride = 0; u.c_bstride = 8L * r * D;
            FTMI_TRY(gemm_tn(u, st));
        }
    }
    return 0;
}

}  // namespace ftmi
