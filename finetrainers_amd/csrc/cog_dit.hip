// CogVideoX DiT block stack, forward / backward orchestrator: one C call launches all L blocks (or a range of blocks of the backward) on the
// caller's stream; every activation the backward needs lives in the caller-provided workspace (no recompute).  The same design as ltx_dit.hip,
// for the joint text+video block of CogVideoX (BASELINE config 3):
//
//   tokens [B, N = T + S, D] (text first)  ->  for each block:
//     n1 = LN(x) * (1 + scale) + shift            (CogVideoXLayerNormZero; text rows and video rows have their own shift / scale / gate)
//     q|k|v = n1 W_qkv^T + b  (+ LoRA, one fused GEMM with a grouped K-extension)    q, k <- per-head LayerNorm [+ RoPE on the video rows]
//     o = softmax(q k^T / 8) v   over all N joint tokens
//     h1 = x + gate * (o W_o^T + b + LoRA)
//     out = h1 + gate_ff * FF(LN(h1) * (1 + scale_ff) + shift_ff)                 (GELU-tanh, over text and video rows alike)
//
// Reference: [upstream] diffusers CogVideoXBlock as driven by finetrainers/models/cogvideox/base_specification.py:296-333, restated in
// oracle/cogvideox.py; the backward is the autograd backward of that graph with frozen base weights (dgrads only) and trainable LoRA A / B.
// The three projections q, k, v read the same n1: their input gradients are summed in one fp32 accumulator of the fused dgrad GEMM (the eager
// graph adds three bf16 tensors).
#include <string.h>

#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

struct CogLayout {
    size_t total = 0;
    size_t mod_raw, tables, hs, blk0, blk_stride;
    size_t n1, qkv, qn, kn, o, lse, xa_qkv, xa_o, h1, n2, z;  // per block, forward
    size_t g_qkv, g_o, dxa_qkv, dxa_o;                        // per block, backward stash for the batched weight gradients
    size_t s_act, s_f, s_big, s_d1, s_d2, s_d3, s_dq, s_dk, s_dh0, s_dh1, s_delta;
};

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

CogLayout make_layout(const ftmi_cog_config& c) {
    CogLayout w;
    const size_t N = (size_t)c.T + c.S, M = (size_t)c.B * N, D = c.D, r = c.r > 0 ? c.r : 64, e2 = 2;
    Bump g;
    w.mod_raw = g.take((size_t)c.B * c.L * 2 * 6 * D * e2);
    w.tables = g.take((size_t)c.L * 2 * 3 * c.B * 2 * D * e2);
    w.hs = g.take((size_t)(c.L > 1 ? c.L - 1 : 1) * M * D * e2);  // inputs of blocks 1 .. L-1
    Bump b;
    w.n1 = b.take(M * D * e2);
    w.qkv = b.take(M * 3 * D * e2);
    w.qn = b.take(M * D * e2);
    w.kn = b.take(M * D * e2);
    w.o = b.take(M * D * e2);
    w.lse = b.take((size_t)c.B * c.H * N * 4);
    w.xa_qkv = b.take(M * 9 * r * e2);
    w.xa_o = b.take(M * 3 * r * e2);
    w.h1 = b.take(M * D * e2);
    w.n2 = b.take(M * D * e2);
    w.z = b.take(M * (size_t)c.D_ff * e2);
    w.g_qkv = b.take(M * 3 * D * e2);
    w.g_o = b.take(M * D * e2);
    w.dxa_qkv = b.take(M * 9 * r * e2);
    w.dxa_o = b.take(M * 3 * r * e2);
    w.blk_stride = b.off;
    w.blk0 = g.take(w.blk_stride * c.L);
    w.s_act = g.take(M * (size_t)c.D_ff * e2);
    w.s_f = g.take(M * D * e2);
    w.s_big = g.take(M * (size_t)c.D_ff * e2);
    w.s_d1 = g.take(M * D * e2);
    w.s_d2 = g.take(M * D * e2);
    w.s_d3 = g.take(M * D * e2);
    w.s_dq = g.take(M * D * e2);
    w.s_dk = g.take(M * D * e2);
    w.s_dh0 = g.take(M * D * e2);
    w.s_dh1 = g.take(M * D * e2);
    w.s_delta = g.take((size_t)c.B * c.H * N * 4);
    w.total = g.off;
    return w;
}

int check_cfg(const ftmi_cog_config& c) {
    if (c.B <= 0 || c.S <= 0 || c.T < 0 || c.L <= 0) return set_error(FTMI_ERR_INVALID, "cog: empty problem");
    if (c.H * 64 != c.D || c.D % 128 != 0 || c.D > 4096) return set_error(FTMI_ERR_UNSUPPORTED, "cog: width must be heads x 64, a multiple of 128, at most 4096");
    if (c.r < 0 || (c.r % 64) != 0) return set_error(FTMI_ERR_UNSUPPORTED, "cog: LoRA rank must be 0 or a multiple of 64");
    if ((c.D_ff % 128) || (c.D_temb % 64)) return set_error(FTMI_ERR_UNSUPPORTED, "cog: D_ff must be a multiple of 128, the time-embedding width of 64");
    return 0;
}

inline const bf16_t* P(const void* base, size_t elem_off) { return reinterpret_cast<const bf16_t*>(base) + elem_off; }
inline bf16_t* W(void* ws, size_t byte_off) { return reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ws) + byte_off); }
inline float* WF(void* ws, size_t byte_off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + byte_off); }

#define FTMI_TRY(x)          \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

// fp32-equivalent LoRA down-projection (see ltx_dit.hip lora_down): out [M, 3 nout] = (hi | lo | hi) planes of alpha * X . Wf^T
int lora_down(const bf16_t* X, long ldx, int M, const bf16_t* w_sp, int nout, int K, int r, float alpha, bf16_t* out, hipStream_t st, int xk_grp_stride = 0) {
    GemmNtArgs a;
    a.X = X; a.ldx = ldx; a.W = w_sp; a.ldw = K; a.M = M; a.N = 2 * nout; a.K = K; a.alpha = alpha;
    if (xk_grp_stride > 0) { a.xk_grp_n = 2 * r; a.xk_grp_stride = xk_grp_stride; }
    a.split_r = r; a.out = out; a.ldo = 3L * nout; a.variant = 8;
    return gemm_nt(a, st);
}

struct Tables {  // the six modulation tables of one block, each [B][2][D]
    const bf16_t *shift1, *onep1, *gate1, *shift2, *onep2, *gate2;
};
Tables tables_of(const ftmi_cog_config& c, void* ws, const CogLayout& L, int l) {
    const size_t one = (size_t)c.B * 2 * c.D;
    const bf16_t* t = W(ws, L.tables) + (size_t)l * 6 * one;
    return {t, t + one, t + 2 * one, t + 3 * one, t + 4 * one, t + 5 * one};
}

CogLnArgs ln_args(const ftmi_cog_config& c, const bf16_t* x, int which_norm, int l, const ftmi_cog_weights& w) {
    CogLnArgs a;
    a.x = x; a.rows = c.B * (c.T + c.S); a.D = c.D; a.rows_per_batch = c.T + c.S; a.seg0 = c.T; a.eps = c.eps_norm;
    a.w = P(w.norm_w, ((size_t)l * 2 + which_norm) * c.D);
    a.b = P(w.norm_b, ((size_t)l * 2 + which_norm) * c.D);
    return a;
}

int gate_res(const ftmi_cog_config& c, const bf16_t* res, const bf16_t* y, const bf16_t* gate, bf16_t* out, hipStream_t st) {
    CogLnArgs a;
    a.x = y; a.onep = gate; a.dres = res; a.y = out; a.rows = c.B * (c.T + c.S); a.D = c.D; a.rows_per_batch = c.T + c.S; a.seg0 = c.T;
    return cog_gate_residual(a, st);
}

AttnArgs attn_args(const ftmi_cog_config& c) {
    AttnArgs a;
    a.B = c.B; a.H = c.H; a.Sq = a.Sk = c.T + c.S; a.scale = 0.125f; a.d = 64;
    return a;
}
inline void set3(long& sb, long& sh, long& ss, long rows_per_batch, long ld) {
    sb = rows_per_batch * ld;
    sh = 64;
    ss = ld;
}

}  // namespace

size_t cog_workspace_bytes(const ftmi_cog_config& c) { return make_layout(c).total; }

int cog_blocks_forward(const ftmi_cog_config& c, const ftmi_cog_weights& w, const bf16_t* tokens_in, const bf16_t* temb_silu, bf16_t* tokens_out, void* ws,
                       size_t ws_bytes, hipStream_t st) {
    FTMI_TRY(check_cfg(c));
    const CogLayout L = make_layout(c);
    if (ws_bytes < L.total) return set_error(FTMI_ERR_INVALID, "cog_blocks_forward: workspace too small");
    const int N = c.T + c.S, M = c.B * N, D = c.D, r = c.r, V = c.gemm_variant;
    const long D2 = (long)D * D;
    const float s = c.lora_scale;

    // modulation of every LayerNorm-zero of every block: one GEMM over the stacked [L, 2, 6D, D_temb] weights, one table-building pass
    {
        GemmNtArgs a;
        a.X = temb_silu; a.ldx = c.D_temb; a.W = P(w.mod_w, 0); a.ldw = c.D_temb; a.M = c.B; a.N = c.L * 2 * 6 * D; a.K = c.D_temb;
        a.bias = P(w.mod_b, 0); a.out = W(ws, L.mod_raw); a.ldo = (long)c.L * 2 * 6 * D; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
        FTMI_TRY(cog_mod_tables(W(ws, L.mod_raw), W(ws, L.tables), c.L * 2, c.B, D, st));
    }

    for (int l = 0; l < c.L; ++l) {
        char* blk = reinterpret_cast<char*>(ws) + L.blk0 + L.blk_stride * l;
        const bf16_t* h0 = l == 0 ? tokens_in : W(ws, L.hs) + (size_t)(l - 1) * M * D;
        bf16_t* hout = l == c.L - 1 ? tokens_out : W(ws, L.hs) + (size_t)l * M * D;
        const Tables t = tables_of(c, ws, L, l);
        const bf16_t* la = w.lora_a_sp ? P(w.lora_a_sp, (size_t)l * 4 * 2 * r * D) : nullptr;   // [4][2r][D]
        const bf16_t* lb = w.lora_b_ext ? P(w.lora_b_ext, (size_t)l * 4 * D * 3 * r) : nullptr;  // [4][D][3r]
        bf16_t* n1 = W(blk, L.n1);
        bf16_t* qkv = W(blk, L.qkv);

        {  // LayerNorm-zero 1
            CogLnArgs a = ln_args(c, h0, 0, l, w);
            a.shift = t.shift1; a.onep = t.onep1; a.y = n1;
            FTMI_TRY(cog_ln_mod_fwd(a, st));
        }
        {  // fused q|k|v projection (+ LoRA)
            GemmNtArgs a;
            a.X = n1; a.ldx = D; a.W = P(w.w_qkv, (size_t)l * 3 * D2); a.ldw = D; a.M = M; a.N = 3 * D; a.K = D;
            a.bias = P(w.b_qkv, (size_t)l * 3 * D); a.out = qkv; a.ldo = 3 * D; a.variant = V;
            if (r > 0) {
                FTMI_TRY(lora_down(n1, D, M, la, 3 * r, D, r, s, W(blk, L.xa_qkv), st));
                a.X2 = W(blk, L.xa_qkv); a.ldx2 = 9 * r; a.W2 = lb; a.ldw2 = 3 * r; a.K2 = 3 * r; a.x2_grp_n = D; a.x2_grp_stride = 3 * r;
            }
            FTMI_TRY(gemm_nt(a, st));
        }
        for (int qk = 0; qk < 2; ++qk) {  // per-head LayerNorm of q and k (+ RoPE on the video rows)
            CogLnArgs a;
            a.x = qkv + (size_t)qk * D; a.ld = 3 * D; a.y = W(blk, qk ? L.kn : L.qn); a.ld_out = D; a.rows = M; a.D = D; a.eps = c.eps_qk;
            a.w = P(w.qk_norm, ((size_t)l * 4 + qk * 2) * 64); a.b = P(w.qk_norm, ((size_t)l * 4 + qk * 2 + 1) * 64);
            a.cos = w.rope_cos; a.sin = w.rope_sin; a.rows_per_batch = N; a.seg0 = w.rope_cos ? c.T : 0;
            FTMI_TRY(cog_head_ln_fwd(a, st));
        }
        {  // joint attention
            AttnArgs a = attn_args(c);
            a.q = W(blk, L.qn); set3(a.q_sb, a.q_sh, a.q_ss, N, D);
            a.k = W(blk, L.kn); set3(a.k_sb, a.k_sh, a.k_ss, N, D);
            a.v = qkv + 2 * D;  set3(a.v_sb, a.v_sh, a.v_ss, N, 3 * D);
            a.o = W(blk, L.o);  set3(a.o_sb, a.o_sh, a.o_ss, N, D);
            a.lse2 = WF(blk, L.lse);
            FTMI_TRY(attn_fwd(a, st));
        }
        {  // to_out (+ LoRA), gated residual
            GemmNtArgs a;
            a.X = W(blk, L.o); a.ldx = D; a.W = P(w.w_o, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D;
            a.bias = P(w.b_o, (size_t)l * D); a.out = W(ws, L.s_f); a.ldo = D; a.variant = V;
            if (r > 0) {
                FTMI_TRY(lora_down(W(blk, L.o), D, M, la + 3L * 2 * r * D, r, D, r, s, W(blk, L.xa_o), st));
                a.X2 = W(blk, L.xa_o); a.ldx2 = 3 * r; a.W2 = lb + 3L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r;
            }
            FTMI_TRY(gemm_nt(a, st));
            FTMI_TRY(gate_res(c, h0, W(ws, L.s_f), t.gate1, W(blk, L.h1), st));
        }
        {  // LayerNorm-zero 2, feed-forward, gated residual
            CogLnArgs a = ln_args(c, W(blk, L.h1), 1, l, w);
            a.shift = t.shift2; a.onep = t.onep2; a.y = W(blk, L.n2);
            FTMI_TRY(cog_ln_mod_fwd(a, st));
            GemmNtArgs f1;
            f1.X = W(blk, L.n2); f1.ldx = D; f1.W = P(w.w_ff1, (size_t)l * c.D_ff * D); f1.ldw = D; f1.M = M; f1.N = c.D_ff; f1.K = D;
            f1.bias = P(w.b_ff1, (size_t)l * c.D_ff); f1.out = W(ws, L.s_act); f1.ldo = c.D_ff; f1.out2 = W(blk, L.z); f1.ldo2 = c.D_ff;
            f1.epi = EPI_GELU; f1.variant = V;
            FTMI_TRY(gemm_nt(f1, st));
            GemmNtArgs f2;
            f2.X = W(ws, L.s_act); f2.ldx = c.D_ff; f2.W = P(w.w_ff2, (size_t)l * D * c.D_ff); f2.ldw = c.D_ff; f2.M = M; f2.N = D; f2.K = c.D_ff;
            f2.bias = P(w.b_ff2, (size_t)l * D); f2.out = W(ws, L.s_f); f2.ldo = D; f2.variant = V;
            FTMI_TRY(gemm_nt(f2, st));
            FTMI_TRY(gate_res(c, W(blk, L.h1), W(ws, L.s_f), t.gate2, hout, st));
        }
    }
    return 0;
}

// Blocks [l_lo, l_hi) of the backward, l_hi - 1 first.  d_out = gradient of block l_hi - 1's output when l_hi == L, otherwise the state left in the workspace by
// the previous call is continued.  When the call returns, grad_a / grad_b of its blocks are final (bucketed all-reduce); d_in (may be NULL) receives the gradient
// of block l_lo's input when l_lo == 0.
int cog_blocks_backward(const ftmi_cog_config& c, const ftmi_cog_weights& w, const bf16_t* tokens_in, const bf16_t* d_out, bf16_t* d_in, float* grad_a,
                        float* grad_b, void* ws, size_t ws_bytes, int l_hi, int l_lo, int accumulate, hipStream_t st) {
    FTMI_TRY(check_cfg(c));
    const CogLayout L = make_layout(c);
    if (ws_bytes < L.total) return set_error(FTMI_ERR_INVALID, "cog_blocks_backward: workspace too small");
    if (l_lo < 0 || l_hi > c.L || l_lo >= l_hi) return set_error(FTMI_ERR_INVALID, "cog_blocks_backward: bad block range");
    const int N = c.T + c.S, M = c.B * N, D = c.D, r = c.r, V = c.gemm_variant;
    const long D2 = (long)D * D;
    const float s = c.lora_scale;
    if (r > 0 && !accumulate) {  // the weight-gradient kernels accumulate: this range's slices start from zero
        const size_t off = (size_t)l_lo * 4 * r * D, nbytes = (size_t)(l_hi - l_lo) * 4 * r * D * sizeof(float);
        if (hipMemsetAsync(grad_a + off, 0, nbytes, st) != hipSuccess || hipMemsetAsync(grad_b + off, 0, nbytes, st) != hipSuccess)
            return set_error(FTMI_ERR_LAUNCH, "cog_blocks_backward: memset of the gradient buffer failed");
    }
    bf16_t* dh[2] = {W(ws, L.s_dh0), W(ws, L.s_dh1)};
    bf16_t *d1 = W(ws, L.s_d1), *d2 = W(ws, L.s_d2), *d3 = W(ws, L.s_d3);
    int cur = (c.L - l_hi) & 1;  // the gradient of the token stream ping-pongs between two buffers, one flip per finished block

    for (int l = l_hi - 1; l >= l_lo; --l) {
        char* blk = reinterpret_cast<char*>(ws) + L.blk0 + L.blk_stride * l;
        const bf16_t* h0 = l == 0 ? tokens_in : W(ws, L.hs) + (size_t)(l - 1) * M * D;
        const Tables t = tables_of(c, ws, L, l);
        const bf16_t* lat = w.lora_at_ext ? P(w.lora_at_ext, (size_t)l * 4 * D * 3 * r) : nullptr;  // [4][D][3r]
        const bf16_t* lbt = w.lora_bt_sp ? P(w.lora_bt_sp, (size_t)l * 4 * 2 * r * D) : nullptr;    // [4][2r][D]
        const bf16_t* dout = l == c.L - 1 ? d_out : dh[cur];
        bf16_t* dx = (l == 0 && d_in) ? d_in : dh[cur ^ 1];
        const bf16_t* qkv = W(blk, L.qkv);
        bf16_t* gqkv = W(blk, L.g_qkv);
        bf16_t* go = W(blk, L.g_o);

        // ---- feed-forward branch ----
        FTMI_TRY(gate_res(c, nullptr, dout, t.gate2, d1, st));  // d f = gate_ff * d out
        {
            GemmNtArgs a;
            a.X = d1; a.ldx = D; a.W = P(w.w_ff2_t, (size_t)l * c.D_ff * D); a.ldw = D; a.M = M; a.N = c.D_ff; a.K = D;
            a.out = W(ws, L.s_big); a.ldo = c.D_ff; a.epi = EPI_DGELU; a.aux = W(blk, L.z); a.ldaux = c.D_ff; a.variant = V;
            FTMI_TRY(gemm_nt(a, st));
            GemmNtArgs b;
            b.X = W(ws, L.s_big); b.ldx = c.D_ff; b.W = P(w.w_ff1_t, (size_t)l * D * c.D_ff); b.ldw = c.D_ff; b.M = M; b.N = D; b.K = c.D_ff;
            b.out = d2; b.ldo = D; b.variant = V;
            FTMI_TRY(gemm_nt(b, st));
            CogLnArgs n = ln_args(c, W(blk, L.h1), 1, l, w);
            n.onep = t.onep2; n.dy = d2; n.dres = dout; n.dx = d3;  // d3 = d h1
            FTMI_TRY(cog_ln_mod_bwd(n, st));
        }
        // ---- attention branch ----
        FTMI_TRY(gate_res(c, nullptr, d3, t.gate1, go, st));  // d(to_out output), kept for its weight gradient
        if (r > 0) FTMI_TRY(lora_down(go, D, M, lbt + 3L * 2 * r * D, r, D, r, s, W(blk, L.dxa_o), st));
        {
            GemmNtArgs a;
            a.X = go; a.ldx = D; a.W = P(w.w_o_t, (size_t)l * D2); a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = d1; a.ldo = D; a.variant = V;
            if (r > 0) { a.X2 = W(blk, L.dxa_o); a.ldx2 = 3 * r; a.W2 = lat + 3L * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r; }
            FTMI_TRY(gemm_nt(a, st));  // d1 = d o
        }
        {
            AttnArgs a = attn_args(c);
            a.q = W(blk, L.qn); set3(a.q_sb, a.q_sh, a.q_ss, N, D);
            a.k = W(blk, L.kn); set3(a.k_sb, a.k_sh, a.k_ss, N, D);
            a.v = qkv + 2 * D;  set3(a.v_sb, a.v_sh, a.v_ss, N, 3 * D);
            a.o = W(blk, L.o);  set3(a.o_sb, a.o_sh, a.o_ss, N, D);
            a.lse2 = WF(blk, L.lse);
            a.dout = d1;              set3(a.do_sb, a.do_sh, a.do_ss, N, D);
            a.delta = WF(ws, L.s_delta);
            a.dq = W(ws, L.s_dq);     set3(a.dq_sb, a.dq_sh, a.dq_ss, N, D);
            a.dk = W(ws, L.s_dk);     set3(a.dk_sb, a.dk_sh, a.dk_ss, N, D);
            a.dv = gqkv + 2 * D;      set3(a.dv_sb, a.dv_sh, a.dv_ss, N, 3 * D);
            FTMI_TRY(attn_bwd(a, st));
        }
        for (int qk = 0; qk < 2; ++qk) {  // head LayerNorm (+ RoPE) backward into the q / k thirds of d(q|k|v)
            CogLnArgs a;
            a.x = qkv + (size_t)qk * D; a.ld = 3 * D; a.dy = W(ws, qk ? L.s_dk : L.s_dq); a.ld_dy = D; a.dx = gqkv + (size_t)qk * D; a.ld_out = 3 * D;
            a.rows = M; a.D = D; a.eps = c.eps_qk; a.w = P(w.qk_norm, ((size_t)l * 4 + qk * 2) * 64);
            a.cos = w.rope_cos; a.sin = w.rope_sin; a.rows_per_batch = N; a.seg0 = w.rope_cos ? c.T : 0;
            FTMI_TRY(cog_head_ln_bwd(a, st));
        }
        if (r > 0) FTMI_TRY(lora_down(gqkv, 3 * D, M, lbt, 3 * r, D, r, s, W(blk, L.dxa_qkv), st, D));
        {
            GemmNtArgs a;  // d n1 = d(q|k|v) W_qkv + d XA A: the three paths in one fp32 accumulator
            a.X = gqkv; a.ldx = 3 * D; a.W = P(w.w_qkv_t, (size_t)l * 3 * D2); a.ldw = 3 * D; a.M = M; a.N = D; a.K = 3 * D; a.out = d2; a.ldo = D; a.variant = V;
            if (r > 0) { a.X2 = W(blk, L.dxa_qkv); a.ldx2 = 9 * r; a.W2 = P(w.lora_at_qkv_ext, (size_t)l * D * 9 * r); a.ldw2 = 9 * r; a.K2 = 9 * r; }
            FTMI_TRY(gemm_nt(a, st));
            CogLnArgs n = ln_args(c, h0, 0, l, w);
            n.onep = t.onep1; n.dy = d2; n.dres = d3; n.dx = dx;
            FTMI_TRY(cog_ln_mod_bwd(n, st));
        }
        cur ^= 1;
    }

    // LoRA weight gradients dB += dY^T XA, dA += dXA^T X: one batched launch per adapter group over the blocks of this range
    if (r > 0) {
        const int nb = l_hi - l_lo;
        char* blk0 = reinterpret_cast<char*>(ws) + L.blk0 + L.blk_stride * l_lo;
        const long bs = (long)(L.blk_stride / 2);
        struct G { size_t dy; long lddy; int nadp, adp; size_t xa, dxa, x; };
        const G groups[2] = {{L.g_o, D, 1, 3, L.xa_o, L.dxa_o, L.o}, {L.g_qkv, 3L * D, 3, 0, L.xa_qkv, L.dxa_qkv, L.n1}};
        float* ga = grad_a + (size_t)l_lo * 4 * r * D;
        float* gb = grad_b + (size_t)l_lo * 4 * D * r;
        for (const G& gr : groups) {
            GemmTnArgs tb;  // dB[l] += dY[l]^T XA[l]
            tb.U = W(blk0, gr.dy); tb.ldu = gr.lddy; tb.V = W(blk0, gr.xa); tb.ldv = (long)gr.nadp * 3 * r; tb.v_fold = r;
            tb.C = gb + (size_t)gr.adp * D * r; tb.ldc = r; tb.M = M; tb.P = gr.nadp * D; tb.Q = r;
            if (gr.nadp > 1) { tb.v_grp_p = D; tb.v_grp_stride = 3 * r; }
            tb.batch = nb; tb.u_bstride = bs; tb.v_bstride = bs; tb.c_bstride = 4L * D * r;
            FTMI_TRY(gemm_tn(tb, st));
            GemmTnArgs ta;  // dA[l] += dXA[l]^T X[l]
            ta.U = W(blk0, gr.dxa); ta.ldu = (long)gr.nadp * 3 * r; ta.u_fold = r; ta.V = W(blk0, gr.x); ta.ldv = D;
            if (gr.nadp > 1) { ta.u_grp_p = r; ta.u_grp_stride = 3 * r; }
            ta.C = ga + (size_t)gr.adp * r * D; ta.ldc = D; ta.M = M; ta.P = gr.nadp * r; ta.Q = D;
            ta.batch = nb; ta.u_bstride = bs; ta.v_bstride = bs; ta.c_bstride = 4L * r * D;
            FTMI_TRY(gemm_tn(ta, st));
        }
    }
    return 0;
}

}  // namespace ftmi
