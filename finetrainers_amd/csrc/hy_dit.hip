// HunyuanVideo single-stream block (40 of the model's 60 blocks), forward / backward orchestrator: ONE C call per direction launches the block's
// kernels on the caller's stream out of two caller-owned buffers -- `saved` (what the backward reads again: per block, lives from the forward to
// the backward, or is rebuilt inside the backward under gradient checkpointing) and `scratch` (transients: one buffer shared by all blocks).
//
//   tokens x [B, N = T + S, D] (text first), temb_silu [B, D] = silu(conditioning vector):
//     (shift, scale, gate) = temb_silu W_mod^T + b                                   AdaLayerNormZeroSingle
//     n    = LN(x) * (1 + scale) + shift
//     mlp  = gelu_tanh(n W_mlp^T + b)                                                 (pre-activation kept)
//     q|k|v = n W^T + b + LoRA        q, k <- per-head RMSNorm, rotary embedding on the video rows
//     o    = softmax(q k^T / sqrt(128) + key_bias) v        over the joint sequence (padded text keys carry -inf)
//     out  = x + gate * ([o | mlp] W_out^T + b)
//
// Reference: [upstream] diffusers HunyuanVideoSingleTransformerBlock as driven by finetrainers/models/hunyuan_video/base_specification.py:294-330,
// restated in oracle/hunyuan.py (SingleStreamBlock); the backward is the autograd backward of that graph with frozen base weights (input
// gradients only) and trainable fp32 LoRA A / B on to_q / to_k / to_v.  The kernel sequence is exactly the one finetrainers_amd/hunyuan_video/block.py
// issues from Python (which stays as the second implementation the tests compare this one with, bit for bit); what the C call removes is ~45 host
// round trips per block and direction, six LoRA operand splits (done once per call here) and torch's allocation of every intermediate.
#include "common.hip.h"
#include "kernels.h"

namespace ftmi {

namespace {

struct Bump {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off += (bytes + 255) & ~(size_t)255;
        return o;
    }
};

struct HyLayout {
    // saved
    size_t mod, shift, onep, gate, n, q, k, v, qn, kn, o, lse, pre, xa, saved_total;
    // scratch, forward
    size_t cat, y, a_sp, b_ext;
    // scratch, backward
    size_t dy, d_o, dpre, dn_mlp, dqn, dkn, dv, dq, dk, dn_q, dn_k, dn_v, dn, dxa, delta, bt_sp, at_ext, scratch_total;
};

HyLayout make_layout(const ftmi_hy_single_config& c) {
    HyLayout w;
    const size_t N = (size_t)c.T + c.S, M = (size_t)c.B * N, D = c.D, mlp = c.mlp, r = c.r > 0 ? c.r : 0, e2 = 2;
    Bump s;
    w.mod = s.take((size_t)c.B * 3 * D * e2);
    w.shift = s.take((size_t)c.B * D * e2);
    w.onep = s.take((size_t)c.B * D * e2);
    w.gate = s.take((size_t)c.B * D * e2);
    w.n = s.take(M * D * e2);
    w.q = s.take(M * D * e2);
    w.k = s.take(M * D * e2);
    w.v = s.take(M * D * e2);
    w.qn = s.take(M * D * e2);
    w.kn = s.take(M * D * e2);
    w.o = s.take(M * D * e2);
    w.lse = s.take((size_t)c.B * c.H * N * 4);
    w.pre = s.take(M * mlp * e2);
    w.xa = s.take(3 * M * 3 * r * e2);
    w.saved_total = s.off;
    Bump f;  // forward and backward transients overlay each other
    w.cat = f.take(M * (D + mlp) * e2);
    w.y = f.take(M * D * e2);
    w.a_sp = f.take(3 * 2 * r * D * e2);
    w.b_ext = f.take(3 * D * 3 * r * e2);
    Bump b;
    w.dy = b.take(M * D * e2);
    w.d_o = b.take(M * D * e2);
    w.dpre = b.take(M * mlp * e2);
    w.dn_mlp = b.take(M * D * e2);
    w.dqn = b.take(M * D * e2);
    w.dkn = b.take(M * D * e2);
    w.dv = b.take(M * D * e2);
    w.dq = b.take(M * D * e2);
    w.dk = b.take(M * D * e2);
    w.dn_q = b.take(M * D * e2);
    w.dn_k = b.take(M * D * e2);
    w.dn_v = b.take(M * D * e2);
    w.dn = b.take(M * D * e2);
    w.dxa = b.take(M * 3 * r * e2);
    w.delta = b.take((size_t)c.B * c.H * N * 4);
    w.bt_sp = b.take(3 * 2 * r * D * e2);
    w.at_ext = b.take(3 * D * 3 * r * e2);
    w.scratch_total = f.off > b.off ? f.off : b.off;
    return w;
}

int check_cfg(const ftmi_hy_single_config& c) {
    if (c.B <= 0 || c.S <= 0 || c.T < 0) return set_error(FTMI_ERR_INVALID, "hy_single: empty problem");
    if (c.H * 128 != c.D || c.D % 128 != 0 || c.D > 4096) return set_error(FTMI_ERR_UNSUPPORTED, "hy_single: width must be heads x 128, at most 4096");
    if (c.r < 0 || (c.r % 64) != 0) return set_error(FTMI_ERR_UNSUPPORTED, "hy_single: LoRA rank must be 0 or a multiple of 64 (pad smaller ranks with zeros)");
    if (c.mlp <= 0 || (c.mlp % 128)) return set_error(FTMI_ERR_UNSUPPORTED, "hy_single: the MLP width must be a multiple of 128");
    return 0;
}

inline bf16_t* W(void* ws, size_t byte_off) { return reinterpret_cast<bf16_t*>(reinterpret_cast<char*>(ws) + byte_off); }
inline float* WF(void* ws, size_t byte_off) { return reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + byte_off); }
inline const bf16_t* C16(const void* p) { return reinterpret_cast<const bf16_t*>(p); }

#define FTMI_TRY(x)          \
    do {                     \
        int _rc = (x);       \
        if (_rc) return _rc; \
    } while (0)

// mod [B, 3D] = (shift | scale | gate) -> three contiguous [B, D] tables: shift, bf(1 + scale), gate   (the eager graph's `1 + scale` is a bf16 op)
__global__ __launch_bounds__(256) void hy_mod3_kernel(const bf16_t* __restrict__ mod, bf16_t* __restrict__ shift, bf16_t* __restrict__ onep,
                                                      bf16_t* __restrict__ gate, int B, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * D) return;
    const int b = i / D, d = i - b * D;
    const bf16_t* m = mod + (size_t)b * 3 * D;
    shift[i] = m[d];
    onep[i] = f2bf(1.0f + bf2f(m[D + d]));
    gate[i] = m[2 * D + d];
}

int lora_down(const bf16_t* X, long ldx, int M, const bf16_t* w_sp, int r, int K, float alpha, bf16_t* out, hipStream_t st) {
    GemmNtArgs d;
    d.X = X; d.ldx = ldx; d.W = w_sp; d.ldw = K; d.M = M; d.N = 2 * r; d.K = K; d.alpha = alpha; d.split_r = r;
    d.out = out; d.ldo = 3L * r; d.variant = 8;
    return gemm_nt(d, st);
}

AttnArgs attn_args(const ftmi_hy_single_config& c, const float* key_bias) {
    AttnArgs a;
    const long N = (long)c.T + c.S, D = c.D;
    a.B = c.B; a.H = c.H; a.Sq = (int)N; a.Sk = (int)N; a.d = 128;
    a.scale = 0.08838834764831845f;  // 1 / sqrt(128)
    a.q_sb = a.k_sb = a.v_sb = a.o_sb = N * D;
    a.q_sh = a.k_sh = a.v_sh = a.o_sh = 128;
    a.q_ss = a.k_ss = a.v_ss = a.o_ss = D;
    a.kbias = key_bias; a.kb_sb = N; a.kb_sh = 0;
    return a;
}

}  // namespace

size_t hy_single_saved_bytes(const ftmi_hy_single_config& c) { return make_layout(c).saved_total; }
size_t hy_single_scratch_bytes(const ftmi_hy_single_config& c) { return make_layout(c).scratch_total; }

// out == nullptr: the recomputation pass of gradient checkpointing -- stops after the attention (nothing downstream is read by the backward)
int hy_single_forward(const ftmi_hy_single_config& c, const ftmi_hy_single_weights& w, const bf16_t* x, const bf16_t* temb_silu, const float* key_bias,
                      const float* rope_cos, const float* rope_sin, bf16_t* out, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                      hipStream_t st) {
    FTMI_TRY(check_cfg(c));
    const HyLayout L = make_layout(c);
    if (saved_bytes < L.saved_total || scratch_bytes < L.scratch_total) return set_error(FTMI_ERR_INVALID, "hy_single_forward: buffer too small");
    if (c.r > 0 && (!w.lora_a || !w.lora_b)) return set_error(FTMI_ERR_INVALID, "hy_single_forward: LoRA rank without adapters");
    const int N = c.T + c.S, M = c.B * N, D = c.D, mlp = c.mlp, r = c.r, V = c.gemm_variant;
    const float s = c.lora_scale;
    bf16_t *shift = W(saved, L.shift), *onep = W(saved, L.onep), *gate = W(saved, L.gate), *n = W(saved, L.n);

    {   // modulation
        GemmNtArgs a;
        a.X = temb_silu; a.ldx = D; a.W = C16(w.norm_lin_w); a.ldw = D; a.M = c.B; a.N = 3 * D; a.K = D; a.bias = C16(w.norm_lin_b);
        a.out = W(saved, L.mod); a.ldo = 3 * D; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
        hipLaunchKernelGGL(hy_mod3_kernel, dim3((c.B * D + 255) / 256), dim3(256), 0, st, W(saved, L.mod), shift, onep, gate, c.B, D);
        FTMI_TRY(check_launch("hy_mod3"));
    }
    {   // n = LN(x) * (1 + scale) + shift   (LayerNorm without affine: ones / zeros)
        CogLnArgs a;
        a.x = x; a.w = C16(w.ones); a.b = C16(w.zeros); a.shift = shift; a.onep = onep; a.y = n; a.rows = M; a.D = D; a.rows_per_batch = N; a.seg0 = 0; a.eps = c.eps;
        FTMI_TRY(cog_ln_mod_fwd(a, st));
    }
    bf16_t* cat = W(scratch, L.cat);
    {   // MLP branch: gelu_tanh(n W_mlp^T + b) straight into the [attention | MLP] feature buffer, pre-activation kept
        GemmNtArgs a;
        a.X = n; a.ldx = D; a.W = C16(w.proj_mlp_w); a.ldw = D; a.M = M; a.N = mlp; a.K = D; a.bias = C16(w.proj_mlp_b);
        a.out = cat + D; a.ldo = D + mlp; a.out2 = W(saved, L.pre); a.ldo2 = mlp; a.epi = EPI_GELU; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
    }
    if (r > 0) {  // operand copies of the fp32 adapters, once per call: A as (hi, lo) row planes, B as [hi | hi | lo] K-extension columns
        LoraSplitArgs sa;
        sa.w = w.lora_a; sa.rows = r; sa.cols = D; sa.nmat = 3; sa.in_bstride = (long)r * D; sa.sp = W(scratch, L.a_sp); sa.sp_bstride = 2L * r * D;
        FTMI_TRY(lora_split(sa, st));
        LoraSplitArgs sb;
        sb.w = w.lora_b; sb.rows = D; sb.cols = r; sb.nmat = 3; sb.in_bstride = (long)D * r; sb.ext = W(scratch, L.b_ext); sb.ext_bstride = 3L * D * r; sb.ld_ext = 3 * r;
        FTMI_TRY(lora_split(sb, st));
    }
    const void* wts[3] = {w.wq, w.wk, w.wv};
    const void* bs[3] = {w.bq, w.bk, w.bv};
    const size_t outs[3] = {L.q, L.k, L.v};
    for (int i = 0; i < 3; ++i) {
        GemmNtArgs a;
        a.X = n; a.ldx = D; a.W = C16(wts[i]); a.ldw = D; a.M = M; a.N = D; a.K = D; a.bias = C16(bs[i]); a.out = W(saved, outs[i]); a.ldo = D; a.variant = V;
        if (r > 0) {
            bf16_t* xa = W(saved, L.xa) + (size_t)i * M * 3 * r;
            FTMI_TRY(lora_down(n, D, M, W(scratch, L.a_sp) + (size_t)i * 2 * r * D, r, D, s, xa, st));
            a.X2 = xa; a.ldx2 = 3 * r; a.W2 = W(scratch, L.b_ext) + (size_t)i * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r;
        }
        FTMI_TRY(gemm_nt(a, st));
    }
    for (int i = 0; i < 2; ++i) {  // per-head RMSNorm + rotary embedding on the video rows of q and k
        CogLnArgs a;
        a.x = W(saved, i ? L.k : L.q); a.w = C16(i ? w.norm_k_w : w.norm_q_w); a.y = W(saved, i ? L.kn : L.qn); a.rows = M; a.D = D; a.ld = D; a.ld_out = D;
        a.eps = c.eps; a.head_dim = 128; a.rms = 1; a.cos = rope_cos; a.sin = rope_sin; a.seg0 = rope_cos ? c.T : 0; a.rows_per_batch = rope_cos ? N : M;
        FTMI_TRY(cog_head_ln_fwd(a, st));
    }
    {
        AttnArgs a = attn_args(c, key_bias);
        a.q = W(saved, L.qn); a.k = W(saved, L.kn); a.v = W(saved, L.v); a.o = W(saved, L.o); a.lse2 = WF(saved, L.lse);
        FTMI_TRY(attn_fwd(a, st));
    }
    if (!out) return 0;
    if (hipMemcpy2DAsync(cat, (size_t)(D + mlp) * 2, W(saved, L.o), (size_t)D * 2, (size_t)D * 2, M, hipMemcpyDeviceToDevice, st) != hipSuccess)
        return set_error(FTMI_ERR_LAUNCH, "hy_single_forward: copy of the attention features failed");
    {
        GemmNtArgs a;
        a.X = cat; a.ldx = D + mlp; a.W = C16(w.proj_out_w); a.ldw = D + mlp; a.M = M; a.N = D; a.K = D + mlp; a.bias = C16(w.proj_out_b);
        a.out = W(scratch, L.y); a.ldo = D; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
    }
    {   // out = x + bf(gate * y)
        CogLnArgs a;
        a.x = W(scratch, L.y); a.onep = gate; a.dres = x; a.y = out; a.rows = M; a.D = D; a.rows_per_batch = N; a.seg0 = 0;
        FTMI_TRY(cog_gate_residual(a, st));
    }
    return 0;
}

// dx <- gradient of the block's input; grad_a [3, r, D] / grad_b [3, D, r] fp32 are ADDED to (.grad semantics).  ones_rows: bf16 [B, D] of 1.0.
int hy_single_backward(const ftmi_hy_single_config& c, const ftmi_hy_single_weights& w, const bf16_t* x, const bf16_t* dout, const float* key_bias,
                       const float* rope_cos, const float* rope_sin, const bf16_t* ones_rows, bf16_t* dx, float* grad_a, float* grad_b, void* saved,
                       size_t saved_bytes, void* scratch, size_t scratch_bytes, hipStream_t st) {
    FTMI_TRY(check_cfg(c));
    const HyLayout L = make_layout(c);
    if (saved_bytes < L.saved_total || scratch_bytes < L.scratch_total) return set_error(FTMI_ERR_INVALID, "hy_single_backward: buffer too small");
    if (c.r > 0 && (!w.lora_a || !w.lora_b || !grad_a || !grad_b)) return set_error(FTMI_ERR_INVALID, "hy_single_backward: LoRA rank without adapters / gradient buffers");
    if (!w.wq_t || !w.wk_t || !w.wv_t || !w.proj_mlp_w_t || !w.proj_out_w_t) return set_error(FTMI_ERR_INVALID, "hy_single_backward: transposed weights missing");
    const int N = c.T + c.S, M = c.B * N, D = c.D, mlp = c.mlp, r = c.r, V = c.gemm_variant;
    const float s = c.lora_scale;
    const bf16_t *onep = W(saved, L.onep), *gate = W(saved, L.gate), *n = W(saved, L.n);
    bf16_t* dy = W(scratch, L.dy);
    {   // d(proj_out output) = gate * d out
        CogLnArgs a;
        a.x = dout; a.onep = gate; a.y = dy; a.rows = M; a.D = D; a.rows_per_batch = N; a.seg0 = 0;
        FTMI_TRY(cog_gate_residual(a, st));
    }
    const bf16_t* wt = C16(w.proj_out_w_t);  // [D + mlp, D]
    {
        GemmNtArgs a;  // gradient of the attention features
        a.X = dy; a.ldx = D; a.W = wt; a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = W(scratch, L.d_o); a.ldo = D; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
        GemmNtArgs b;  // (gradient of the MLP features) * gelu'(pre)
        b.X = dy; b.ldx = D; b.W = wt + (size_t)D * D; b.ldw = D; b.M = M; b.N = mlp; b.K = D; b.out = W(scratch, L.dpre); b.ldo = mlp;
        b.epi = EPI_DGELU; b.aux = W(saved, L.pre); b.ldaux = mlp; b.variant = V;
        FTMI_TRY(gemm_nt(b, st));
        GemmNtArgs d;
        d.X = W(scratch, L.dpre); d.ldx = mlp; d.W = C16(w.proj_mlp_w_t); d.ldw = mlp; d.M = M; d.N = D; d.K = mlp; d.out = W(scratch, L.dn_mlp); d.ldo = D; d.variant = V;
        FTMI_TRY(gemm_nt(d, st));
    }
    {
        AttnArgs a = attn_args(c, key_bias);
        a.q = W(saved, L.qn); a.k = W(saved, L.kn); a.v = W(saved, L.v); a.o = W(saved, L.o); a.lse2 = WF(saved, L.lse);
        a.dout = W(scratch, L.d_o); a.dq = W(scratch, L.dqn); a.dk = W(scratch, L.dkn); a.dv = W(scratch, L.dv); a.delta = WF(scratch, L.delta);
        a.do_sb = a.dq_sb = a.dk_sb = a.dv_sb = (long)N * D;
        a.do_sh = a.dq_sh = a.dk_sh = a.dv_sh = 128;
        a.do_ss = a.dq_ss = a.dk_ss = a.dv_ss = D;
        FTMI_TRY(attn_bwd(a, st));
    }
    for (int i = 0; i < 2; ++i) {
        CogLnArgs a;
        a.x = W(saved, i ? L.k : L.q); a.w = C16(i ? w.norm_k_w : w.norm_q_w); a.dy = W(scratch, i ? L.dkn : L.dqn); a.dx = W(scratch, i ? L.dk : L.dq);
        a.rows = M; a.D = D; a.ld = D; a.ld_dy = D; a.ld_out = D; a.eps = c.eps; a.head_dim = 128; a.rms = 1;
        a.cos = rope_cos; a.sin = rope_sin; a.seg0 = rope_cos ? c.T : 0; a.rows_per_batch = rope_cos ? N : M;
        FTMI_TRY(cog_head_ln_bwd(a, st));
    }
    if (r > 0) {
        LoraSplitArgs sb;  // B^T as (hi, lo) row planes: operand of dxa = s * dy B
        sb.w = w.lora_b; sb.rows = D; sb.cols = r; sb.nmat = 3; sb.in_bstride = (long)D * r; sb.t_sp = W(scratch, L.bt_sp); sb.t_sp_bstride = 2L * r * D;
        FTMI_TRY(lora_split(sb, st));
        LoraSplitArgs sa;  // A^T as K-extension columns: dx += dxa A
        sa.w = w.lora_a; sa.rows = r; sa.cols = D; sa.nmat = 3; sa.in_bstride = (long)r * D; sa.t_ext = W(scratch, L.at_ext); sa.t_ext_bstride = 3L * D * r; sa.ld_t_ext = 3 * r;
        FTMI_TRY(lora_split(sa, st));
    }
    const bf16_t* dys[3] = {W(scratch, L.dq), W(scratch, L.dk), W(scratch, L.dv)};
    const void* wts[3] = {w.wq_t, w.wk_t, w.wv_t};
    const size_t dns[3] = {L.dn_q, L.dn_k, L.dn_v};
    for (int i = 0; i < 3; ++i) {
        bf16_t* dxa = W(scratch, L.dxa);
        const bf16_t* xa = W(saved, L.xa) + (size_t)i * M * 3 * r;
        if (r > 0) FTMI_TRY(lora_down(dys[i], D, M, W(scratch, L.bt_sp) + (size_t)i * 2 * r * D, r, D, s, dxa, st));
        GemmNtArgs a;  // dn_i = dy_i W_i (+ dxa A_i)
        a.X = dys[i]; a.ldx = D; a.W = C16(wts[i]); a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = W(scratch, dns[i]); a.ldo = D; a.variant = V;
        if (r > 0) { a.X2 = dxa; a.ldx2 = 3 * r; a.W2 = W(scratch, L.at_ext) + (size_t)i * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r; }
        FTMI_TRY(gemm_nt(a, st));
        if (r > 0) {
            GemmTnArgs t;  // dB_i += dy_i^T xa_i
            t.U = dys[i]; t.ldu = D; t.V = xa; t.ldv = 3 * r; t.v_fold = r; t.C = grad_b + (size_t)i * D * r; t.ldc = r; t.M = M; t.P = D; t.Q = r;
            FTMI_TRY(gemm_tn(t, st));
            GemmTnArgs u;  // dA_i += dxa^T n
            u.U = dxa; u.ldu = 3 * r; u.u_fold = r; u.V = n; u.ldv = D; u.C = grad_a + (size_t)i * r * D; u.ldc = D; u.M = M; u.P = r; u.Q = D;
            FTMI_TRY(gemm_tn(u, st));
        }
    }
    // the four consumers of n: their gradients add as bf16 tensors, in autograd's order (MLP + v, + k, + q)
    const size_t adds[3] = {L.dn_v, L.dn_k, L.dn_q};
    const bf16_t* acc = W(scratch, L.dn_mlp);
    for (int i = 0; i < 3; ++i) {
        CogLnArgs a;
        a.x = W(scratch, adds[i]); a.onep = ones_rows; a.dres = acc; a.y = W(scratch, L.dn); a.rows = M; a.D = D; a.rows_per_batch = N; a.seg0 = 0;
        FTMI_TRY(cog_gate_residual(a, st));
        acc = W(scratch, L.dn);
    }
    {
        CogLnArgs a;
        a.x = x; a.w = C16(w.ones); a.onep = onep; a.dy = W(scratch, L.dn); a.dres = dout; a.dx = dx; a.rows = M; a.D = D; a.rows_per_batch = N; a.seg0 = 0; a.eps = c.eps;
        FTMI_TRY(cog_ln_mod_bwd(a, st));
    }
    return 0;
}

}  // namespace ftmi
