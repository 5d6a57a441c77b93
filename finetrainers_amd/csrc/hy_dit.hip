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

// ------------------------------------------------------------------------------------------------------------------------------------------------
// Dual-stream block (20 of the 60 blocks), ONE sample per call (modulation, attention and text mask are per sample):
//   video x_v [S, D] and text x_t [T, D] keep their own AdaLayerNormZero modulation (6 rows each), q / k / v projections (LoRA on the VIDEO stream's
//   to_q / to_k / to_v / to_out.0), per-head q / k RMSNorm (rotary embedding on the video rows) and GELU-tanh feed-forward; q, k, v of both streams are
//   laid out as ONE joint sequence [text | video] for the attention.
// Reference: [upstream] diffusers HunyuanVideoTransformerBlock, restated in oracle/hunyuan.py (DualStreamBlock); kernel sequence = the one
// finetrainers_amd/hunyuan_video/block.py (_DualBlockFunction) issues from Python.
namespace {

struct HyDualLayout {
    // saved: tables[s][i] (s = 0 video, 1 text; i: 0 shift, 1 gate, 2 shift_mlp, 3 gate_mlp, 4 1 + scale, 5 1 + scale_mlp), each [D]
    size_t mod, tables, n_v, n_t, q_v, k_v, q_t, k_t, qj, kj, vj, o, lse, h_v, h_t, pre_v, pre_t, xa, saved_total;
    // scratch, forward
    size_t a_v, a_t, n2_v, n2_t, act_v, act_t, f_v, f_t, a_sp, b_ext;
    // scratch, backward
    size_t df_v, df_t, dg_v, dg_t, dn2_v, dn2_t, dh_v, dh_t, da_v, da_t, doj, dqj, dkj, dvj, delta, dq_v, dk_v, dn_q, dn_k, dn_vv, dn_v, dq_t, dk_t, t1, t2, t3,
        dn_t, dxa, bt_sp, at_ext, scratch_total;
};

HyDualLayout make_dual_layout(const ftmi_hy_dual_config& c) {
    HyDualLayout w;
    const size_t S = c.S, T = c.T, N = S + T, D = c.D, mlp = c.mlp, r = c.r > 0 ? c.r : 0, e2 = 2;
    Bump s;
    w.mod = s.take(2 * 6 * D * e2);
    w.tables = s.take(2 * 6 * D * e2);
    w.n_v = s.take(S * D * e2);
    w.n_t = s.take(T * D * e2);
    w.q_v = s.take(S * D * e2);
    w.k_v = s.take(S * D * e2);
    w.q_t = s.take(T * D * e2);
    w.k_t = s.take(T * D * e2);
    w.qj = s.take(N * D * e2);
    w.kj = s.take(N * D * e2);
    w.vj = s.take(N * D * e2);
    w.o = s.take(N * D * e2);
    w.lse = s.take((size_t)c.H * N * 4);
    w.h_v = s.take(S * D * e2);
    w.h_t = s.take(T * D * e2);
    w.pre_v = s.take(S * mlp * e2);
    w.pre_t = s.take(T * mlp * e2);
    w.xa = s.take(4 * S * 3 * r * e2);
    w.saved_total = s.off;
    Bump f;
    w.a_v = f.take(S * D * e2);
    w.a_t = f.take(T * D * e2);
    w.n2_v = f.take(S * D * e2);
    w.n2_t = f.take(T * D * e2);
    w.act_v = f.take(S * mlp * e2);
    w.act_t = f.take(T * mlp * e2);
    w.f_v = f.take(S * D * e2);
    w.f_t = f.take(T * D * e2);
    w.a_sp = f.take(4 * 2 * r * D * e2);
    w.b_ext = f.take(4 * D * 3 * r * e2);
    Bump b;
    w.df_v = b.take(S * D * e2);
    w.df_t = b.take(T * D * e2);
    w.dg_v = b.take(S * mlp * e2);
    w.dg_t = b.take(T * mlp * e2);
    w.dn2_v = b.take(S * D * e2);
    w.dn2_t = b.take(T * D * e2);
    w.dh_v = b.take(S * D * e2);
    w.dh_t = b.take(T * D * e2);
    w.da_v = b.take(S * D * e2);
    w.da_t = b.take(T * D * e2);
    w.doj = b.take(N * D * e2);
    w.dqj = b.take(N * D * e2);
    w.dkj = b.take(N * D * e2);
    w.dvj = b.take(N * D * e2);
    w.delta = b.take((size_t)c.H * N * 4);
    w.dq_v = b.take(S * D * e2);
    w.dk_v = b.take(S * D * e2);
    w.dn_q = b.take(S * D * e2);
    w.dn_k = b.take(S * D * e2);
    w.dn_vv = b.take(S * D * e2);
    w.dn_v = b.take(S * D * e2);
    w.dq_t = b.take(T * D * e2);
    w.dk_t = b.take(T * D * e2);
    w.t1 = b.take(T * D * e2);
    w.t2 = b.take(T * D * e2);
    w.t3 = b.take(T * D * e2);
    w.dn_t = b.take(T * D * e2);
    w.dxa = b.take(S * 3 * r * e2);
    w.bt_sp = b.take(4 * 2 * r * D * e2);
    w.at_ext = b.take(4 * D * 3 * r * e2);
    w.scratch_total = f.off > b.off ? f.off : b.off;
    return w;
}

int check_dual_cfg(const ftmi_hy_dual_config& c) {
    if (c.S <= 0 || c.T <= 0) return set_error(FTMI_ERR_INVALID, "hy_dual: empty problem");
    if (c.H * 128 != c.D || c.D % 128 != 0 || c.D > 4096) return set_error(FTMI_ERR_UNSUPPORTED, "hy_dual: width must be heads x 128, at most 4096");
    if (c.r < 0 || (c.r % 64) != 0) return set_error(FTMI_ERR_UNSUPPORTED, "hy_dual: LoRA rank must be 0 or a multiple of 64 (pad smaller ranks with zeros)");
    if (c.mlp <= 0 || (c.mlp % 128)) return set_error(FTMI_ERR_UNSUPPORTED, "hy_dual: the feed-forward width must be a multiple of 128");
    return 0;
}

// mod [6D] = (shift, scale, gate, shift_mlp, scale_mlp, gate_mlp) -> tables (shift, gate, shift_mlp, gate_mlp, bf(1 + scale), bf(1 + scale_mlp)), for both streams
__global__ __launch_bounds__(256) void hy_mod6_kernel(const bf16_t* __restrict__ mod, bf16_t* __restrict__ tables, int D) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * D) return;
    const int sidx = i / D, d = i - sidx * D;
    const bf16_t* m = mod + (size_t)sidx * 6 * D;
    bf16_t* t = tables + (size_t)sidx * 6 * D;
    t[0 * D + d] = m[0 * D + d];
    t[1 * D + d] = m[2 * D + d];
    t[2 * D + d] = m[3 * D + d];
    t[3 * D + d] = m[5 * D + d];
    t[4 * D + d] = f2bf(1.0f + bf2f(m[1 * D + d]));
    t[5 * D + d] = f2bf(1.0f + bf2f(m[4 * D + d]));
}

int plain_linear(const bf16_t* X, int M, int K, const void* Wm, const void* bias, int N, bf16_t* out, int V, hipStream_t st) {
    GemmNtArgs a;
    a.X = X; a.ldx = K; a.W = C16(Wm); a.ldw = K; a.M = M; a.N = N; a.K = K; a.bias = C16(bias); a.out = out; a.ldo = N; a.variant = V;
    return gemm_nt(a, st);
}
int ln_mod(const bf16_t* x, const ftmi_hy_dual_weights& w, const bf16_t* shift, const bf16_t* onep, bf16_t* y, int rows, int D, float eps, hipStream_t st) {
    CogLnArgs a;
    a.x = x; a.w = C16(w.ones); a.b = C16(w.zeros); a.shift = shift; a.onep = onep; a.y = y; a.rows = rows; a.D = D; a.rows_per_batch = rows; a.seg0 = 0; a.eps = eps;
    return cog_ln_mod_fwd(a, st);
}
int ln_mod_back(const bf16_t* x, const ftmi_hy_dual_weights& w, const bf16_t* onep, const bf16_t* dy, const bf16_t* dres, bf16_t* dx, int rows, int D, float eps,
                hipStream_t st) {
    CogLnArgs a;
    a.x = x; a.w = C16(w.ones); a.onep = onep; a.dy = dy; a.dres = dres; a.dx = dx; a.rows = rows; a.D = D; a.rows_per_batch = rows; a.seg0 = 0; a.eps = eps;
    return cog_ln_mod_bwd(a, st);
}
int gate_res(const bf16_t* res, const bf16_t* y, const bf16_t* gate, bf16_t* out, int rows, int D, hipStream_t st) {  // out = [res +] bf(gate * y)
    CogLnArgs a;
    a.x = y; a.onep = gate; a.dres = res; a.y = out; a.rows = rows; a.D = D; a.rows_per_batch = rows; a.seg0 = 0;
    return cog_gate_residual(a, st);
}
int head_norm(const bf16_t* x, const void* wn, bf16_t* y, const bf16_t* dy, int rows, int D, float eps, const float* cos_t, const float* sin_t, bool backward,
              hipStream_t st) {
    CogLnArgs a;
    a.x = x; a.w = C16(wn); a.rows = rows; a.D = D; a.ld = D; a.ld_dy = D; a.ld_out = D; a.eps = eps; a.head_dim = 128; a.rms = 1;
    a.cos = cos_t; a.sin = sin_t; a.seg0 = 0; a.rows_per_batch = rows;
    if (backward) { a.dy = dy; a.dx = y; return cog_head_ln_bwd(a, st); }
    a.y = y;
    return cog_head_ln_fwd(a, st);
}
AttnArgs dual_attn_args(const ftmi_hy_dual_config& c, const float* key_bias) {
    AttnArgs a;
    const long N = (long)c.T + c.S, D = c.D;
    a.B = 1; a.H = c.H; a.Sq = (int)N; a.Sk = (int)N; a.d = 128;
    a.scale = 0.08838834764831845f;
    a.q_sb = a.k_sb = a.v_sb = a.o_sb = N * D;
    a.q_sh = a.k_sh = a.v_sh = a.o_sh = 128;
    a.q_ss = a.k_ss = a.v_ss = a.o_ss = D;
    a.kbias = key_bias; a.kb_sb = N; a.kb_sh = 0;
    return a;
}
// y = x W^T + b (+ LoRA adapter i); xa_i kept
int lora_linear_fwd(const bf16_t* X, int M, int D, const void* Wm, const void* bias, int i, int r, float s, void* scratch, const HyDualLayout& L, bf16_t* xa_all,
                    size_t xa_stride, bf16_t* out, int V, hipStream_t st) {
    GemmNtArgs a;
    a.X = X; a.ldx = D; a.W = C16(Wm); a.ldw = D; a.M = M; a.N = D; a.K = D; a.bias = C16(bias); a.out = out; a.ldo = D; a.variant = V;
    if (r > 0) {
        bf16_t* xa = xa_all + (size_t)i * xa_stride;
        FTMI_TRY(lora_down(X, D, M, W(scratch, L.a_sp) + (size_t)i * 2 * r * D, r, D, s, xa, st));
        a.X2 = xa; a.ldx2 = 3 * r; a.W2 = W(scratch, L.b_ext) + (size_t)i * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r;
    }
    return gemm_nt(a, st);
}
// dx = dy W (+ dxa A_i);  dB_i += dy^T xa_i, dA_i += dxa^T x
int lora_linear_bwd(const bf16_t* X, const bf16_t* dy, int M, int D, const void* Wt, int i, int r, float s, void* scratch, const HyDualLayout& L, const bf16_t* xa_all,
                    size_t xa_stride, bf16_t* dx, float* grad_a, float* grad_b, int V, hipStream_t st) {
    bf16_t* dxa = W(scratch, L.dxa);
    if (r > 0) FTMI_TRY(lora_down(dy, D, M, W(scratch, L.bt_sp) + (size_t)i * 2 * r * D, r, D, s, dxa, st));
    GemmNtArgs a;
    a.X = dy; a.ldx = D; a.W = C16(Wt); a.ldw = D; a.M = M; a.N = D; a.K = D; a.out = dx; a.ldo = D; a.variant = V;
    if (r > 0) { a.X2 = dxa; a.ldx2 = 3 * r; a.W2 = W(scratch, L.at_ext) + (size_t)i * D * 3 * r; a.ldw2 = 3 * r; a.K2 = 3 * r; }
    FTMI_TRY(gemm_nt(a, st));
    if (r > 0) {
        GemmTnArgs t;
        t.U = dy; t.ldu = D; t.V = xa_all + (size_t)i * xa_stride; t.ldv = 3 * r; t.v_fold = r; t.C = grad_b + (size_t)i * D * r; t.ldc = r; t.M = M; t.P = D; t.Q = r;
        FTMI_TRY(gemm_tn(t, st));
        GemmTnArgs u;
        u.U = dxa; u.ldu = 3 * r; u.u_fold = r; u.V = X; u.ldv = D; u.C = grad_a + (size_t)i * r * D; u.ldc = D; u.M = M; u.P = r; u.Q = D;
        FTMI_TRY(gemm_tn(u, st));
    }
    return 0;
}

}  // namespace

size_t hy_dual_saved_bytes(const ftmi_hy_dual_config& c) { return make_dual_layout(c).saved_total; }
size_t hy_dual_scratch_bytes(const ftmi_hy_dual_config& c) { return make_dual_layout(c).scratch_total; }

// out_v == nullptr: the recomputation pass of gradient checkpointing (stops before the second feed-forward GEMMs)
int hy_dual_forward(const ftmi_hy_dual_config& c, const ftmi_hy_dual_weights& w, const bf16_t* x_v, const bf16_t* x_t, const bf16_t* temb_silu, const float* key_bias,
                    const float* rope_cos, const float* rope_sin, bf16_t* out_v, bf16_t* out_t, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                    hipStream_t st) {
    FTMI_TRY(check_dual_cfg(c));
    const HyDualLayout L = make_dual_layout(c);
    if (saved_bytes < L.saved_total || scratch_bytes < L.scratch_total) return set_error(FTMI_ERR_INVALID, "hy_dual_forward: buffer too small");
    if (c.r > 0 && (!w.lora_a || !w.lora_b)) return set_error(FTMI_ERR_INVALID, "hy_dual_forward: LoRA rank without adapters");
    if ((out_v == nullptr) != (out_t == nullptr)) return set_error(FTMI_ERR_INVALID, "hy_dual_forward: both outputs or none");
    const int S = c.S, T = c.T, N = S + T, D = c.D, mlp = c.mlp, r = c.r, V = c.gemm_variant;
    const float s = c.lora_scale, eps = c.eps;
    bf16_t* tab = W(saved, L.tables);
    auto TV = [&](int i) { return tab + (size_t)i * D; };            // video: shift, gate, shift_mlp, gate_mlp, 1 + scale, 1 + scale_mlp
    auto TT = [&](int i) { return tab + (size_t)(6 + i) * D; };      // text
    FTMI_TRY(plain_linear(temb_silu, 1, D, w.norm1_lin_w, w.norm1_lin_b, 6 * D, W(saved, L.mod), V, st));
    FTMI_TRY(plain_linear(temb_silu, 1, D, w.norm1c_lin_w, w.norm1c_lin_b, 6 * D, W(saved, L.mod) + (size_t)6 * D, V, st));
    hipLaunchKernelGGL(hy_mod6_kernel, dim3((2 * D + 255) / 256), dim3(256), 0, st, W(saved, L.mod), tab, D);
    FTMI_TRY(check_launch("hy_mod6"));
    bf16_t *n_v = W(saved, L.n_v), *n_t = W(saved, L.n_t);
    FTMI_TRY(ln_mod(x_v, w, TV(0), TV(4), n_v, S, D, eps, st));
    FTMI_TRY(ln_mod(x_t, w, TT(0), TT(4), n_t, T, D, eps, st));
    if (r > 0) {
        LoraSplitArgs sa;
        sa.w = w.lora_a; sa.rows = r; sa.cols = D; sa.nmat = 4; sa.in_bstride = (long)r * D; sa.sp = W(scratch, L.a_sp); sa.sp_bstride = 2L * r * D;
        FTMI_TRY(lora_split(sa, st));
        LoraSplitArgs sb;
        sb.w = w.lora_b; sb.rows = D; sb.cols = r; sb.nmat = 4; sb.in_bstride = (long)D * r; sb.ext = W(scratch, L.b_ext); sb.ext_bstride = 3L * D * r; sb.ld_ext = 3 * r;
        FTMI_TRY(lora_split(sb, st));
    }
    bf16_t *qj = W(saved, L.qj), *kj = W(saved, L.kj), *vj = W(saved, L.vj), *xa = W(saved, L.xa);
    const size_t xas = (size_t)S * 3 * r;
    FTMI_TRY(lora_linear_fwd(n_v, S, D, w.wq, w.bq, 0, r, s, scratch, L, xa, xas, W(saved, L.q_v), V, st));
    FTMI_TRY(lora_linear_fwd(n_v, S, D, w.wk, w.bk, 1, r, s, scratch, L, xa, xas, W(saved, L.k_v), V, st));
    FTMI_TRY(lora_linear_fwd(n_v, S, D, w.wv, w.bv, 2, r, s, scratch, L, xa, xas, vj + (size_t)T * D, V, st));  // v of the video rows straight into the joint buffer
    FTMI_TRY(head_norm(W(saved, L.q_v), w.norm_q_w, qj + (size_t)T * D, nullptr, S, D, eps, rope_cos, rope_sin, false, st));
    FTMI_TRY(head_norm(W(saved, L.k_v), w.norm_k_w, kj + (size_t)T * D, nullptr, S, D, eps, rope_cos, rope_sin, false, st));
    FTMI_TRY(plain_linear(n_t, T, D, w.add_q_w, w.add_q_b, D, W(saved, L.q_t), V, st));
    FTMI_TRY(plain_linear(n_t, T, D, w.add_k_w, w.add_k_b, D, W(saved, L.k_t), V, st));
    FTMI_TRY(plain_linear(n_t, T, D, w.add_v_w, w.add_v_b, D, vj, V, st));
    FTMI_TRY(head_norm(W(saved, L.q_t), w.norm_added_q_w, qj, nullptr, T, D, eps, nullptr, nullptr, false, st));
    FTMI_TRY(head_norm(W(saved, L.k_t), w.norm_added_k_w, kj, nullptr, T, D, eps, nullptr, nullptr, false, st));
    bf16_t* o = W(saved, L.o);
    {
        AttnArgs a = dual_attn_args(c, key_bias);
        a.q = qj; a.k = kj; a.v = vj; a.o = o; a.lse2 = WF(saved, L.lse);
        FTMI_TRY(attn_fwd(a, st));
    }
    bf16_t *a_v = W(scratch, L.a_v), *a_t = W(scratch, L.a_t), *h_v = W(saved, L.h_v), *h_t = W(saved, L.h_t);
    FTMI_TRY(lora_linear_fwd(o + (size_t)T * D, S, D, w.wo, w.bo, 3, r, s, scratch, L, xa, xas, a_v, V, st));
    FTMI_TRY(plain_linear(o, T, D, w.add_out_w, w.add_out_b, D, a_t, V, st));
    FTMI_TRY(gate_res(x_v, a_v, TV(1), h_v, S, D, st));
    FTMI_TRY(gate_res(x_t, a_t, TT(1), h_t, T, D, st));
    bf16_t *n2_v = W(scratch, L.n2_v), *n2_t = W(scratch, L.n2_t);
    FTMI_TRY(ln_mod(h_v, w, TV(2), TV(5), n2_v, S, D, eps, st));
    FTMI_TRY(ln_mod(h_t, w, TT(2), TT(5), n2_t, T, D, eps, st));
    for (int sidx = 0; sidx < 2; ++sidx) {  // feed-forward, first GEMM: gelu_tanh, pre-activation kept
        GemmNtArgs a;
        a.X = sidx ? n2_t : n2_v; a.ldx = D; a.W = C16(sidx ? w.ffc1_w : w.ff1_w); a.ldw = D; a.M = sidx ? T : S; a.N = mlp; a.K = D; a.bias = C16(sidx ? w.ffc1_b : w.ff1_b);
        a.out = W(scratch, sidx ? L.act_t : L.act_v); a.ldo = mlp; a.out2 = W(saved, sidx ? L.pre_t : L.pre_v); a.ldo2 = mlp; a.epi = EPI_GELU; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
    }
    if (!out_v) return 0;
    FTMI_TRY(plain_linear(W(scratch, L.act_v), S, mlp, w.ff2_w, w.ff2_b, D, W(scratch, L.f_v), V, st));
    FTMI_TRY(gate_res(h_v, W(scratch, L.f_v), TV(3), out_v, S, D, st));
    FTMI_TRY(plain_linear(W(scratch, L.act_t), T, mlp, w.ffc2_w, w.ffc2_b, D, W(scratch, L.f_t), V, st));
    FTMI_TRY(gate_res(h_t, W(scratch, L.f_t), TT(3), out_t, T, D, st));
    return 0;
}

// ones_row: bf16 [D] of 1.0.  grad_a [4, r, D] / grad_b [4, D, r] fp32 are ADDED to.
int hy_dual_backward(const ftmi_hy_dual_config& c, const ftmi_hy_dual_weights& w, const bf16_t* x_v, const bf16_t* x_t, const bf16_t* dout_v, const bf16_t* dout_t,
                     const float* key_bias, const float* rope_cos, const float* rope_sin, const bf16_t* ones_row, bf16_t* dx_v, bf16_t* dx_t, float* grad_a,
                     float* grad_b, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, hipStream_t st) {
    FTMI_TRY(check_dual_cfg(c));
    const HyDualLayout L = make_dual_layout(c);
    if (saved_bytes < L.saved_total || scratch_bytes < L.scratch_total) return set_error(FTMI_ERR_INVALID, "hy_dual_backward: buffer too small");
    if (c.r > 0 && (!w.lora_a || !w.lora_b || !grad_a || !grad_b)) return set_error(FTMI_ERR_INVALID, "hy_dual_backward: LoRA rank without adapters / gradient buffers");
    if (!w.wq_t || !w.wk_t || !w.wv_t || !w.wo_t || !w.add_q_w_t || !w.add_k_w_t || !w.add_v_w_t || !w.add_out_w_t || !w.ff1_w_t || !w.ff2_w_t || !w.ffc1_w_t || !w.ffc2_w_t)
        return set_error(FTMI_ERR_INVALID, "hy_dual_backward: transposed weights missing");
    const int S = c.S, T = c.T, N = S + T, D = c.D, mlp = c.mlp, r = c.r, V = c.gemm_variant;
    const float s = c.lora_scale, eps = c.eps;
    const bf16_t* tab = W(saved, L.tables);
    auto TV = [&](int i) { return tab + (size_t)i * D; };
    auto TT = [&](int i) { return tab + (size_t)(6 + i) * D; };
    // feed-forward branches: out = h + gate_mlp * FF(LN(h) * (1 + scale_mlp) + shift_mlp)
    for (int sidx = 0; sidx < 2; ++sidx) {
        const int rows = sidx ? T : S;
        bf16_t* df = W(scratch, sidx ? L.df_t : L.df_v);
        const bf16_t* dout = sidx ? dout_t : dout_v;
        FTMI_TRY(gate_res(nullptr, dout, sidx ? TT(3) : TV(3), df, rows, D, st));
        GemmNtArgs a;  // (d of the activations) * gelu'(pre)
        a.X = df; a.ldx = D; a.W = C16(sidx ? w.ffc2_w_t : w.ff2_w_t); a.ldw = D; a.M = rows; a.N = mlp; a.K = D; a.out = W(scratch, sidx ? L.dg_t : L.dg_v); a.ldo = mlp;
        a.epi = EPI_DGELU; a.aux = W(saved, sidx ? L.pre_t : L.pre_v); a.ldaux = mlp; a.variant = V;
        FTMI_TRY(gemm_nt(a, st));
        GemmNtArgs b;
        b.X = W(scratch, sidx ? L.dg_t : L.dg_v); b.ldx = mlp; b.W = C16(sidx ? w.ffc1_w_t : w.ff1_w_t); b.ldw = mlp; b.M = rows; b.N = D; b.K = mlp;
        b.out = W(scratch, sidx ? L.dn2_t : L.dn2_v); b.ldo = D; b.variant = V;
        FTMI_TRY(gemm_nt(b, st));
        FTMI_TRY(ln_mod_back(W(saved, sidx ? L.h_t : L.h_v), w, sidx ? TT(5) : TV(5), W(scratch, sidx ? L.dn2_t : L.dn2_v), dout, W(scratch, sidx ? L.dh_t : L.dh_v), rows, D,
                             eps, st));
    }
    if (r > 0) {
        LoraSplitArgs sb;
        sb.w = w.lora_b; sb.rows = D; sb.cols = r; sb.nmat = 4; sb.in_bstride = (long)D * r; sb.t_sp = W(scratch, L.bt_sp); sb.t_sp_bstride = 2L * r * D;
        FTMI_TRY(lora_split(sb, st));
        LoraSplitArgs sa;
        sa.w = w.lora_a; sa.rows = r; sa.cols = D; sa.nmat = 4; sa.in_bstride = (long)r * D; sa.t_ext = W(scratch, L.at_ext); sa.t_ext_bstride = 3L * D * r; sa.ld_t_ext = 3 * r;
        FTMI_TRY(lora_split(sa, st));
    }
    const bf16_t *dh_v = W(scratch, L.dh_v), *dh_t = W(scratch, L.dh_t), *o = W(saved, L.o), *xa = W(saved, L.xa);
    const size_t xas = (size_t)S * 3 * r;
    bf16_t* doj = W(scratch, L.doj);
    // attention outputs: h = x + gate_msa * (o W_o^T + b)
    FTMI_TRY(gate_res(nullptr, dh_v, TV(1), W(scratch, L.da_v), S, D, st));
    FTMI_TRY(lora_linear_bwd(o + (size_t)T * D, W(scratch, L.da_v), S, D, w.wo_t, 3, r, s, scratch, L, xa, xas, doj + (size_t)T * D, grad_a, grad_b, V, st));
    FTMI_TRY(gate_res(nullptr, dh_t, TT(1), W(scratch, L.da_t), T, D, st));
    FTMI_TRY(plain_linear(W(scratch, L.da_t), T, D, w.add_out_w_t, nullptr, D, doj, V, st));
    {
        AttnArgs a = dual_attn_args(c, key_bias);
        a.q = W(saved, L.qj); a.k = W(saved, L.kj); a.v = W(saved, L.vj); a.o = W(saved, L.o); a.lse2 = WF(saved, L.lse);
        a.dout = doj; a.dq = W(scratch, L.dqj); a.dk = W(scratch, L.dkj); a.dv = W(scratch, L.dvj); a.delta = WF(scratch, L.delta);
        a.do_sb = a.dq_sb = a.dk_sb = a.dv_sb = (long)N * D;
        a.do_sh = a.dq_sh = a.dk_sh = a.dv_sh = 128;
        a.do_ss = a.dq_ss = a.dk_ss = a.dv_ss = D;
        FTMI_TRY(attn_bwd(a, st));
    }
    const bf16_t *dqj = W(scratch, L.dqj), *dkj = W(scratch, L.dkj), *dvj = W(scratch, L.dvj);
    // video stream: RMSNorm + rotary backward, the three LoRA projections
    FTMI_TRY(head_norm(W(saved, L.q_v), w.norm_q_w, W(scratch, L.dq_v), dqj + (size_t)T * D, S, D, eps, rope_cos, rope_sin, true, st));
    FTMI_TRY(head_norm(W(saved, L.k_v), w.norm_k_w, W(scratch, L.dk_v), dkj + (size_t)T * D, S, D, eps, rope_cos, rope_sin, true, st));
    const bf16_t* n_v = W(saved, L.n_v);
    FTMI_TRY(lora_linear_bwd(n_v, W(scratch, L.dq_v), S, D, w.wq_t, 0, r, s, scratch, L, xa, xas, W(scratch, L.dn_q), grad_a, grad_b, V, st));
    FTMI_TRY(lora_linear_bwd(n_v, W(scratch, L.dk_v), S, D, w.wk_t, 1, r, s, scratch, L, xa, xas, W(scratch, L.dn_k), grad_a, grad_b, V, st));
    FTMI_TRY(lora_linear_bwd(n_v, dvj + (size_t)T * D, S, D, w.wv_t, 2, r, s, scratch, L, xa, xas, W(scratch, L.dn_vv), grad_a, grad_b, V, st));
    FTMI_TRY(gate_res(W(scratch, L.dn_vv), W(scratch, L.dn_k), ones_row, W(scratch, L.dn_v), S, D, st));   // bf16 accumulation of the three gradients of n_v
    FTMI_TRY(gate_res(W(scratch, L.dn_v), W(scratch, L.dn_q), ones_row, W(scratch, L.dn_v), S, D, st));
    FTMI_TRY(ln_mod_back(x_v, w, TV(4), W(scratch, L.dn_v), dh_v, dx_v, S, D, eps, st));
    // text stream
    FTMI_TRY(head_norm(W(saved, L.q_t), w.norm_added_q_w, W(scratch, L.dq_t), dqj, T, D, eps, nullptr, nullptr, true, st));
    FTMI_TRY(head_norm(W(saved, L.k_t), w.norm_added_k_w, W(scratch, L.dk_t), dkj, T, D, eps, nullptr, nullptr, true, st));
    FTMI_TRY(plain_linear(dvj, T, D, w.add_v_w_t, nullptr, D, W(scratch, L.t1), V, st));
    FTMI_TRY(plain_linear(W(scratch, L.dk_t), T, D, w.add_k_w_t, nullptr, D, W(scratch, L.t2), V, st));
    FTMI_TRY(gate_res(W(scratch, L.t1), W(scratch, L.t2), ones_row, W(scratch, L.dn_t), T, D, st));
    FTMI_TRY(plain_linear(W(scratch, L.dq_t), T, D, w.add_q_w_t, nullptr, D, W(scratch, L.t3), V, st));
    FTMI_TRY(gate_res(W(scratch, L.dn_t), W(scratch, L.t3), ones_row, W(scratch, L.dn_t), T, D, st));
    FTMI_TRY(ln_mod_back(x_t, w, TT(4), W(scratch, L.dn_t), dh_t, dx_t, T, D, eps, st));
    return 0;
}

}  // namespace ftmi
