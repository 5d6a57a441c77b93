/* libftmi355 -- MI355X (gfx950) native kernels for the finetrainers LTX-Video LoRA SFT step.
 *
 * C ABI: plain pointers (device memory unless stated), explicit sizes/strides, a HIP stream, caller-owned
 * workspaces.  No allocation and no ownership inside the library; every call is stream-ordered (the library
 * never synchronises the device) and re-entrant.  Return value: 0 = ok, negative = error (see FTMI_ERR_*),
 * message via ftmi_last_error().
 *
 * The reference (a-r-r-o-w/finetrainers, pure Python) has no FFI; the entry points below sit behind its two
 * plugin surfaces.  Each entry names the reference interface it replaces (paths relative to the reference
 * tree):
 *
 *   ftmi_attn_fwd / ftmi_attn_bwd ...... a provider function of finetrainers/models/attention_dispatch.py:405-447
 *                                        (contract :295-362; native provider :938-962) and its autograd backward
 *   ftmi_ltx_forward / ftmi_ltx_backward[_range] the transformer call inside LTXVideoModelSpecification.forward
 *                                        (finetrainers/models/ltx_video/base_specification.py:336-342), i.e.
 *                                        _patched_LTXVideoTransformer3D_forward (finetrainers/patches/models/
 *                                        ltx_video/patch.py:38-127) + loss.backward() through it
 *                                        (finetrainers/trainer/sft_trainer/trainer.py:481)
 *   ftmi_ltx_noise_pack ................ normalise / noise / flow-match mix / pack / target of
 *                                        base_specification.py:295-320,343,427-459 + functional/diffusion.py:4-11
 *   ftmi_mse_loss ...................... trainer/sft_trainer/trainer.py:463-480 (+ d loss / d pred)
 *   ftmi_clip_adamw_step ............... utils/torch.py:99-161,299-374 (clip_grad_norm_) +
 *                                        optimizer.py:117-125 (torch.optim.AdamW step) over the flat LoRA buffer
 *   ftmi_lora_refresh / ftmi_lora_split  (new) bf16 (hi, lo) working copies of the fp32 LoRA matrices after a step
 *   ftmi_linear_lora_fwd / _bwd ........ peft lora.Linear.forward over a frozen nn.Linear
 *                                        (trainer/sft_trainer/trainer.py:121-136 injects them) and the autograd
 *                                        backward the reference gets from loss.backward() (trainer.py:481): dgrad to
 *                                        the input, weight gradients of A and B only (the base weight is frozen)
 */
#ifndef FTMI355_H
#define FTMI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTMI_OK 0
#define FTMI_ERR_INVALID (-1)     /* bad argument (ValueError on the Python side) */
#define FTMI_ERR_UNSUPPORTED (-2) /* shape/dtype outside what the gfx950 kernels cover (ValueError) */
#define FTMI_ERR_LAUNCH (-3)      /* HIP launch/runtime failure (RuntimeError) */

typedef void* ftmi_stream; /* hipStream_t */

int ftmi_version(void);
/* Copies the message of the most recent failing call MADE BY THE CALLING THREAD (messages live in a ring tagged with the failing thread's
 * id -- the trainer's forward and the autograd engine's backward run on different threads and cannot overwrite each other's message);
 * falls back to the newest message of any thread.  No thread-local state inside the library. */
int ftmi_last_error(char* buf, size_t len);

/* In-stream HIP-event profiler used by bench.py for its live roofline figures: while enabled, every stride-th launch of a
 * kernel class is bracketed by two events on the launch stream (stride 1 = every launch; bracketing all ~1500 launches
 * of a step costs ~8 % of the step, sampling keeps the measurement inside the timed region at <1 %).  Classes: 0 gemm_nt
 * (the tiled kernel), 1 gemm_tn, 2 attention forward, 3 attention backward (dQ [+ delta] + dK/dV), 4 the skinny (N <= 256) LoRA
 * down-projection GEMM.  ftmi_prof_summary waits for the recorded
 * events and returns, for one class, the summed device time / count / algorithmic FLOPs of the SAMPLED launches and the
 * count / FLOPs of ALL launches seen (reset != 0 clears the class). */
int ftmi_prof_enable(int stride);
int ftmi_prof_summary(int kernel_class, double* sampled_ms, long* sampled_launches, double* sampled_flops, long* all_launches,
                      double* all_flops, int reset);

/* ------------------------------------------------------------------------------------------------------------
 * Attention provider level.  q,k,v,out,dout,dq,dk,dv: bf16, head_dim desc.d = 64 (LTX-Video, CogVideoX) or 128 (Wan, HunyuanVideo)
 * contiguous; element (b,h,s,:) lives at
 * base + b*stride[0] + h*stride[1] + s*stride[2] (strides in elements).  lse: fp32 [B,H,Sq] (log2 domain,
 * written by fwd, read by bwd).  key_bias: optional fp32 additive bias per (batch, head, key) (an attn_mask that broadcasts over
 * queries; desc.bias_strides = {Sk, 0} for the usual [B,Sk] mask shared by the heads), NULL for none.  A bias of -inf removes the key
 * (softmax weight exactly 0).  Non-causal, no dropout.
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, H, Sq, Sk, d;
    long q_strides[3], k_strides[3], v_strides[3], o_strides[3];
    long do_strides[3], dq_strides[3], dk_strides[3], dv_strides[3]; /* backward only */
    long bias_strides[2]; /* key_bias element (b,h,j) at key_bias[b*bias_strides[0] + h*bias_strides[1] + j]; {Sk, 0} = one row per sample */
    float scale;
} ftmi_attn_desc;

int ftmi_attn_fwd(const ftmi_attn_desc* desc, const void* q, const void* k, const void* v, void* out, float* lse,
                  const float* key_bias, ftmi_stream stream);
/* delta_ws: fp32 [B,H,Sq] scratch (rowsum(dO * O), written by the dQ kernel, read by the dK/dV kernel) */
int ftmi_attn_bwd(const ftmi_attn_desc* desc, const void* q, const void* k, const void* v, const void* out,
                  const float* lse, const void* dout, void* dq, void* dk, void* dv, float* delta_ws,
                  const float* key_bias, ftmi_stream stream);

/* ------------------------------------------------------------------------------------------------------------
 * Linear (+ LoRA) building block:  y = bf16( bf16(x W^T + b) + lora_scale * (x A^T) B^T ), the LoRA branch at fp32-equivalent
 * precision (peft lora.Linear with fp32 adapter weights).  a_sp [2r,K] and b_ext [N,3r] are working copies of the fp32 A [r,K] and
 * B [N,r] made by ftmi_lora_split (sp of A, ext of B); xa_out [M,3r] receives lora_scale * x A^T as bf16 planes (hi | lo | hi) and
 * is kept for the backward.  r == 0 => plain linear (a_sp, b_ext, xa_out may be NULL).
 * ------------------------------------------------------------------------------------------------------------ */
int ftmi_linear_lora_fwd(int M, int K, int N, int r, float lora_scale, const void* x, const void* w, const void* bias,
                         const void* a_sp, const void* b_ext, void* y, void* xa_out, int variant, ftmi_stream stream);

/* Backward of the above.  dy [M,N] bf16; xa [M,3r] = the forward's xa_out; w_t [K,N] = W^T (bf16, made once with
 * ftmi_transpose_bf16); bt_sp [2r,N] = t_sp of B, at_ext [K,3r] = t_ext of A (ftmi_lora_split).
 *   dxa_ws [M,3r] bf16 scratch <- lora_scale * dy B as planes (hi | lo | hi)
 *   dx [M,K] bf16            <- bf16(bf16(dy W) + dxa A)            (may be NULL: input needs no gradient)
 *   grad_a [r,K] fp32        += dxa^T x       grad_b [N,r] fp32 += dy^T xa      (accumulated, like .grad)
 * r == 0 => dx only. */
int ftmi_linear_lora_bwd(int M, int K, int N, int r, float lora_scale, const void* x, const void* dy, const void* xa,
                         const void* w_t, const void* bt_sp, const void* at_ext, void* dxa_ws, void* dx, float* grad_a,
                         float* grad_b, int variant, ftmi_stream stream);

/* Generic building blocks (exposed for tests / incremental adoption) */
/* out[M,N] = bf16(alpha * x[M,K] w[N,K]^T + bias) ; epilogue: 0 store, 1 gelu-tanh (out2 <- pre-activation),
 * 2 out = resid + (gate ? gate[b] * y : y), 3 out = y * gelu'(aux).  out2, resid and aux are [M, N] views with row stride ld_side (0: out's ldo).
 * variant: 8 = automatic kernel / tile choice (use this); 60 = the persistent 256 x 256 stream-K kernel (FTMI_ERR_UNSUPPORTED where it does
 * not apply), 61 = automatic choice among the one-tile-per-workgroup kernels only;
 * other ids pin one kernel (bit-identical A/B partners, see gemm.hip) */
int ftmi_gemm_nt(int M, int N, int K, const void* x, long ldx, const void* w, long ldw, const void* bias, float alpha,
                 void* out, long ldo, int epilogue, void* out2, const void* resid, const void* gate, int rows_per_batch,
                 const void* aux, long ld_side, int variant, ftmi_stream stream);
/* Which kernel variant 8 takes for a plain launch of this shape (K2 = depth of a fused LoRA K-extension or 0; epilogue as above), as a pure host function
 * (no device, no launch): 80 / 86 / 87 = the 16 x 16 x 32 pipeline with 256- / 192- / 224-row tiles (1386 = 192-row tiles with the W operand on a three-slot direct-to-LDS ring: the default of the 192-row launches; 2286 = 192-row tiles with the register-staged operand prefetch, FTMI_NT16_W3=0), 42 = 192 x 128 tiles (two workgroups per CU), 47 = 256 x 256
 * (8 waves), 44 = 128 x 128, 2 = a skinny kernel (N <= 256, plain store, no extension, M >= 512: the LDS-ring kernel when K % 256 == 0, else the direct-gather one), 1 = the 128 x 64 kernel of N % 128 != 0,
 * 0 = not a tiled launch (N % 64 or K % 64).  Host tests pin the choice to DESIGN.md. */
int ftmi_gemm_nt_plan(int M, int N, int K, int K2, int epilogue);
/* ---- in-library gradient exchange (replaces what replicate(model, bucket_cap_mb=100) does for the LoRA gradients: finetrainers/parallel/ptd.py:462-463) ----
 * One process per GPU; the collectives are RCCL's (xGMI inside a node), looked up with dlopen at the first call -- libftmi355.so has no link-time dependency on
 * librccl and these entry points return FTMI_ERR_UNSUPPORTED where it cannot be found.
 *   ftmi_allreduce_unique_id : rank 0 obtains the 128-byte rendezvous id; the host side hands the bytes to every rank (any channel: torch.distributed's store,
 *                              a file, MPI).
 *   ftmi_allreduce_init      : COLLECTIVE -- every rank calls it with the same id, its rank and the world size; creates the communicator, the library's
 *                              communication stream and its events.  The handle is owned by the caller (ftmi_allreduce_destroy).
 *   ftmi_allreduce_bucket    : all-reduce (mean if average != 0, else sum) of count fp32 values IN PLACE, ordered after everything already queued on
 *                              compute_stream (event hand-over, no host sync), running on the communication stream: the caller goes on queueing the
 *                              backward of the earlier blocks while the bucket is on the wire.  Every rank must issue the same buckets in the same order
 *                              (the block-range schedule of ftmi_ltx_backward_range is a function of L alone).
 *   ftmi_allreduce_wait      : compute_stream waits (device side) for every bucket issued so far -- call before clip + AdamW.
 * Thread-safe per handle (a mutex); no thread-local state. */
typedef void* ftmi_exchange;
int ftmi_allreduce_unique_id(void* id128);
int ftmi_allreduce_init(const void* id128, int rank, int world, ftmi_exchange* out);
int ftmi_allreduce_bucket(ftmi_exchange ex, float* grad, size_t count, int average, ftmi_stream compute_stream);
int ftmi_allreduce_wait(ftmi_exchange ex, ftmi_stream compute_stream);
long ftmi_allreduce_buckets_issued(ftmi_exchange ex);
int ftmi_allreduce_version(void); /* RCCL's version code, -1 if librccl could not be loaded */
int ftmi_allreduce_destroy(ftmi_exchange ex);

/* The FTMI_* tuning switches that select between bit-identical kernels (FTMI_ATTN_PL, FTMI_ATTN_FEWKEYS, FTMI_SKINNY4) are read from the environment ONCE,
 * at the first launch that consults them -- the launch path never calls getenv.  A process that wants to compare kernels (the bit-identity tests do) changes
 * the environment and calls this: every switch consulted so far is re-read.  Returns how many were. */
int ftmi_reload_switches(void);
/* FTMI_FUSE_DOWN=1 (round 6): a LoRA down-projection and the GEMM that consumes it run as ONE launch -- the down-projection's workgroups lead the grid, the GEMM's
 * tiles wait for the rows they need on per-row-tile counters two K stages before their K-extension (csrc/gemm.hip: gemm_nt16_fused_kernel; results are the bits of
 * the two separate launches).  A wait is bounded: a poll that gives up raises a device-side status word and goes on.  This returns that word (0 = every wait of
 * every fused launch so far was satisfied); it synchronises the device -- tests only. */
int ftmi_fused_status(void);
#ifdef FTMI_EXPERIMENTAL
/* Research build only (FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build): the persistent stream-K GEMM (tools/experimental/gemm_sk.hip, variant 60) --
 * parity-green and 5-25 % slower than the shipped kernels on the step's shapes (profiles/r03_gemm_streamk.txt); not part of the product ABI. */
/* The stream-K split of the persistent GEMM as a pure host function (no device needed; tests): the last `ntiles mod n_workgroups` tiles of
 * a launch form a cost line of (nk + owner_cost) units per tile -- nk K iterations, owner_cost = what finishing the tile (LoRA extension +
 * epilogue) costs in K-iteration units -- cut into n_workgroups shares of equal cost; a cut inside a tile charges partial_cost to the workgroup
 * after it (it stores a partial) and add_cost to the tile's owner; cuts closer than min_piece iterations to a tile edge snap to it.
 * work[w] = {t0, k0, t1, k1, kinds, n_whole, contributor_mask, 0}: the share of workgroup w starts at K iteration k0 of stream-K tile t0 and
 * ends before iteration k1 of tile t1; kinds bit 0 = it opens with a partial piece (published to its owner), bit 1 = it closes with the owner
 * piece [0, k1) of tile t1; n_whole whole tiles in between; contributor_mask bit i = workgroup w + 1 + i holds a partial of that tile. */
int ftmi_gemm_sk_plan(int ntiles, int n_workgroups, int nk, int owner_cost, int min_piece, int partial_cost, int add_cost,
                      int* work /* [n_workgroups][8] */);
/* 0 = every stream-K hand-off since the previous call completed; 1 = a bounded wait for a partial gave up (results of that launch are wrong);
 * < 0 = error.  Reads and clears the word; call it after synchronising the streams that ran stream-K launches (tests and debugging only).
 * Liveness: a workgroup takes share G - 1 - blockIdx of the stream-K work, so the partials it waits for belong to workgroups that were dispatched before
 * it and publish them as their first action -- the launch makes progress whatever else occupies the device (FTMI_SK_ORDER=0 selects the first,
 * XCD-contiguous numbering, whose launches are chained across streams instead). */
int ftmi_gemm_sk_status(void);
/* Debugging aid (FTMI_SK_TRACE=1): shader-clock stamps of the last stream-K launch, out[n_workgroups][16] (start, end of each K phase,
 * hand-off waits, segment ends; 0 = unused); returns n_workgroups, 0 if nothing was traced.  Synchronises the device. */
int ftmi_gemm_sk_trace(unsigned long long* out, int capacity);
#endif /* FTMI_EXPERIMENTAL */
/* c[P,Q] (fp32) += scale * u[M,P]^T v[M,Q] */
int ftmi_gemm_tn(int M, int P, int Q, const void* u, long ldu, const void* v, long ldv, float* c, long ldc, float scale,
                 ftmi_stream stream);
/* fp8 weight storage of the layerwise up-casting recipe (finetrainers/trainer/sft_trainer/trainer.py:111-118: storage float8_e4m3fn, compute bf16):
 * dst (bf16) = exact up-cast of src [rows, cols] (OCP e4m3fn bytes); transpose != 0: dst is [cols, rows] = src^T (rows, cols multiples of 64) -- the layout
 * the input-gradient GEMMs read.  Called per block right before it runs; 1 byte read + 2 bytes written per weight. */
int ftmi_fp8_upcast(const void* src, void* dst, int rows, int cols, int transpose, ftmi_stream stream);
/* out[cols,rows] = in[rows,cols]^T (bf16); used once at load time for the dgrad copies of frozen weights */
int ftmi_transpose_bf16(const void* in, void* out, int rows, int cols, ftmi_stream stream);

/* Row-wise building blocks (width D = 2048; one wavefront per token row; every op of the eager chain fused in registers with a bf16
 * round wherever the reference's eager bf16 graph materialises a tensor).
 *  norm_modulate: y = bf16(bf16(norm(x)) * onep[b]) + shift[b], norm = RMSNorm (layernorm = 0) or LayerNorm (1), no affine -- the
 *    reference's _patched_rms_norm_forward (finetrainers/patches/dependencies/diffusers/rms_norm.py:17-29) / nn.LayerNorm followed by
 *    the AdaLN modulate `* (1 + scale) + shift` of the LTX block; b = row / rows_per_batch, shift / onep rows mod_bstride apart.
 *    bwd: dx = (dres ? dres + : ) norm_bwd(x, bf16(dy * onep[b])).
 *  qknorm_rope: y = rope(bf16(rms_norm(x) * w)) -- norm_q / norm_k (affine RMSNorm over the full width) followed by apply_rotary_emb
 *    (finetrainers/patches/models/ltx_video/patch.py:23-33); cos / sin fp32 [rows_per_batch, D/2] (one value per rotated pair) or NULL
 *    for no rotation (cross-attention).  bwd returns d x.  Row strides ldx / ldy / lddy / lddx in elements. */
int ftmi_norm_modulate_fwd(const void* x, const void* shift, const void* onep, long mod_bstride, void* y, int rows, int rows_per_batch,
                           int D, float eps, int layernorm, ftmi_stream stream);
int ftmi_norm_modulate_bwd(const void* x, const void* dy, const void* onep, long mod_bstride, const void* dres, void* dx, int rows,
                           int rows_per_batch, int D, float eps, int layernorm, ftmi_stream stream);
int ftmi_qknorm_rope_fwd(const void* x, long ldx, const void* w, const float* cos_t, const float* sin_t, void* y, long ldy, int rows,
                         int rows_per_batch, int D, float eps, ftmi_stream stream);
int ftmi_qknorm_rope_bwd(const void* x, long ldx, const void* w, const float* cos_t, const float* sin_t, const void* dy, long lddy,
                         void* dx, long lddx, int rows, int rows_per_batch, int D, float eps, ftmi_stream stream);

/* ------------------------------------------------------------------------------------------------------------
 * LTX-Video DiT level
 * ------------------------------------------------------------------------------------------------------------ */
typedef struct {
    int B, S, T;      /* batch, video tokens / sample, text tokens / sample */
    int D, H, L;      /* width (2048), heads (D/H must be 64), blocks */
    int C_in, C_out;  /* latent channels (128) */
    int D_ff, D_cap;  /* 8192, 4096 */
    int r;            /* LoRA rank: 0 or a multiple of 64 */
    float lora_scale; /* alpha / r */
    float eps_norm;   /* 1e-6 (norm1 / norm2 / norm_out) */
    float eps_qk;     /* 1e-5 (norm_q / norm_k) */
    int gemm_variant; /* 8 = automatic tile / K-loop choice (production); other ids select one fixed kernel (A/B partners, gemm.hip) */
    /* 1 = gradient checkpointing (the reference's --gradient_checkpointing, trainer/sft_trainer/trainer.py:155-157 -> utils/activation_checkpoint.py:24-49,
     * every block wrapped): the workspace holds ONE block's activations instead of L (ftmi_ltx_workspace_bytes shrinks accordingly), the forward keeps only
     * the residual stream, and the backward runs every block's forward kernels again right before its gradient kernels.  Gradients are bit-identical to
     * checkpoint = 0 (deterministic kernels).  Forward and backward of one step must use the same value. */
    int checkpoint;
    /* A NARROW model embedded in this layout by zero padding (round 6: the reference's dummy fixture, tests/models/ltx_video/base_specification.py:47-58 -- 4 heads x 8,
     * one block -- through the production kernels): every weight / bias / table zero-padded to D = 2048 (head h of the narrow model occupies channels [64 h, 64 h + hd)),
     * channel counts and the LoRA rank padded to multiples of 64 (finetrainers_amd/ltx_video/padded.py builds the image).  Padded channels then carry exact zeros
     * everywhere; what the kernels must be told is over how many channels a normalisation takes its mean (d_valid) and the true head width for the softmax scale
     * 1 / sqrt(head_dim_valid).  0 = the full width (2048 / 64). */
    int d_valid, head_dim_valid;
} ftmi_ltx_config;

/* All weights bf16 unless noted.  Per-block tensors are stacked along a leading L dimension.  "*_t" are
 * transposed copies (made once at load) so that every dgrad is the same K-contiguous GEMM as the forward.
 * LoRA order inside a block: 0 attn1.to_q, 1 attn1.to_k, 2 attn1.to_v, 3 attn1.to_out.0,
 *                            4 attn2.to_q, 5 attn2.to_k, 6 attn2.to_v, 7 attn2.to_out.0 */
typedef struct {
    const void *proj_in_w, *proj_in_b;                   /* [D,C_in] [D] */
    const void *time_l1_w, *time_l1_b;                   /* [D,256] [D] */
    const void *time_l2_w, *time_l2_b;                   /* [D,D] [D] */
    const void *time_lin_w, *time_lin_b;                 /* [6D,D] [6D] */
    const void *cap_l1_w, *cap_l1_b;                     /* [D,D_cap] [D] */
    const void *cap_l2_w, *cap_l2_b;                     /* [D,D] [D] */
    const void* tables;                                  /* [L,6,D] scale_shift_table of every block */
    const void* table_out;                               /* [2,D] */
    const void *proj_out_w, *proj_out_b, *proj_out_w_t;  /* [C_out,D] [C_out] [D,C_out] */
    const void *w_qkv, *b_qkv, *w_qkv_t;                 /* [L,3D,D] [L,3D] [L,D,3D] */
    const void *norm_q, *norm_k;                         /* [L,D] */
    const void *w_o, *b_o, *w_o_t;                       /* [L,D,D] [L,D] [L,D,D] */
    const void *w_q2, *b_q2, *w_q2_t;                    /* [L,D,D] [L,D] [L,D,D] */
    const void *w_kv2, *b_kv2;                           /* [L,2D,D] [L,2D] */
    const void *norm_q2, *norm_k2;                       /* [L,D] */
    const void *w_o2, *b_o2, *w_o2_t;                    /* [L,D,D] [L,D] [L,D,D] */
    const void *w_ff1, *b_ff1, *w_ff1_t;                 /* [L,D_ff,D] [L,D_ff] [L,D,D_ff] */
    const void *w_ff2, *b_ff2, *w_ff2_t;                 /* [L,D,D_ff] [L,D] [L,D_ff,D] */
    /* bf16 working copies of the fp32 LoRA matrices, refreshed by ftmi_lora_refresh.  The reference computes the LoRA branch in
     * fp32 (trainer/sft_trainer/trainer.py:132-136); here every fp32 LoRA value v travels as two bf16 numbers hi = bf16(v),
     * lo = bf16(v - hi) (16 mantissa bits) and every product is evaluated as hi*hi + lo*hi + hi*lo on the bf16 MFMA with fp32
     * accumulation -- fp32-equivalent to ~2^-16 relative. */
    const void* lora_a_sp;       /* [L,8,2r,D]  A   as (hi, lo) row planes interleaved per 32 rows: operand of x A^T      */
    const void* lora_bt_sp;      /* [L,8,2r,D]  B^T the same way:                                  operand of dY B        */
    const void* lora_b_ext;      /* [L,8,D,3r]  [B_hi | B_hi | B_lo]:   K-extension operand of the forward projections   */
    const void* lora_at_ext;     /* [L,8,D,3r]  [A^T_hi | A^T_hi | A^T_lo]: K-extension operand of the dgrads            */
    const void* lora_at_qkv_ext; /* [L,D,9r]    the same for the fused q|k|v dgrad (adapters 0,1,2 side by side)          */
    const float *rope_cos, *rope_sin; /* fp32 [S, D/2]: one (cos, sin) per rotated pair */
} ftmi_ltx_weights;

size_t ftmi_ltx_workspace_bytes(const ftmi_ltx_config* cfg);
/* Byte offset of a named activation inside the workspace (for tests / debugging): global names "hs" (residual
 * stream [L+1,B*S,D]; layer selects the slice), "e", "emb", "temb", "ada", "ada_out", "kv2_all" ([B*T, L*2D] text-side k|v of
 * every block), "k2n_all" ([B*T, L, D]), "xa_kv2_all"; per-block names "n1", "qkv", "qrot", "krot", "o1", "lse1", "xa_qkv",
 * "xa_o", "h1", "q2raw", "q2n", "o2", "lse2", "xa_q2", "xa_o2", "h2", "z". */
int ftmi_ltx_workspace_offset(const ftmi_ltx_config* cfg, const char* name, int layer, size_t* offset);

/* Forward of the DiT: x_t [B,S,C_in], text [B,T,D_cap], key_bias fp32 [B,T] ((1-mask) * -10000 as the reference
 * builds it), timestep fp32 [B] (= float(long(sigma*1000)), base_specification.py:320; one per sample) -> pred [B,S,C_out].
 * Activations needed by the backward are kept in ws. */
int ftmi_ltx_forward(const ftmi_ltx_config* cfg, const ftmi_ltx_weights* w, const void* x_t, const void* text,
                     const float* key_bias, const float* timestep, void* pred, void* ws, size_t ws_bytes, ftmi_stream stream);

/* Backward: dpred [B,S,C_out] bf16 -> LoRA gradients ACCUMULATED (+=) into fp32 grad_a [L,8,r,D] and grad_b [L,8,D,r]. */
int ftmi_ltx_backward(const ftmi_ltx_config* cfg, const ftmi_ltx_weights* w, const void* text, const float* key_bias,
                      const void* dpred, float* grad_a, float* grad_b, void* ws, size_t ws_bytes, ftmi_stream stream);

/* The same backward in pieces: blocks [l_lo, l_hi) in descending order (the tail first when l_hi == L).  Successive calls must tile
 * L..0 (e.g. [21,28) [14,21) [7,14) [0,7)); the state between calls lives in ws.  When a call returns (stream order) the gradients of
 * all 8 adapters of exactly those blocks are final, so a data-parallel caller starts the all-reduce of grad_a[l_lo:l_hi] /
 * grad_b[l_lo:l_hi] on its communication stream while the next call computes -- the bucketed, overlapped gradient exchange of the
 * reference's DDP (finetrainers/parallel/ptd.py:462-463: replicate(bucket_cap_mb=100)) without a reducer.
 * accumulate = 0: the whole gradient buffer is zeroed first (by the call with l_hi == L), i.e. ".grad was None";
 * accumulate = 1: gradients are added to the buffer's content (gradient accumulation). */
int ftmi_ltx_backward_range(const ftmi_ltx_config* cfg, const ftmi_ltx_weights* w, const void* text, const float* key_bias,
                            const void* dpred, float* grad_a, float* grad_b, void* ws, size_t ws_bytes, int l_hi, int l_lo,
                            int accumulate, ftmi_stream stream);

/* latents, noise [B,C,F*H*W] bf16; mean,std fp32 [C]; sigma fp32 [B]; sigma_first fp32 [B] or NULL (first-frame
 * conditioning branch: tokens < first_frame_tokens use it) -> x_t, target [B,S,C] bf16 */
int ftmi_ltx_noise_pack(const void* latents, const void* noise, const float* mean, const float* std_, const float* sigma,
                        const float* sigma_first, int first_frame_tokens, void* x_t, void* target, int B, int C, int S,
                        ftmi_stream stream);

/* ------------------------------------------------------------------------------------------------------------
 * CogVideoX spec level (first kernels of the next model family): the DDIM noising and the velocity -> x0 conversion of
 * CogVideoXModelSpecification.forward (finetrainers/models/cogvideox/base_specification.py:283-293 and :326-329; scheduler ops
 * add_noise / get_velocity of [upstream] CogVideoXDDIMScheduler).  latents / noise / sample / outputs: bf16 [B, per_sample];
 * sqrt_alpha, sqrt_one_minus_alpha: fp32 [B] holding the scheduler's bf16-rounded sqrt(alphas_cumprod[t]), sqrt(1 - alphas_cumprod[t]).
 *   add_noise:    x0 = bf16(latents * scaling_factor)  (the training target, may be NULL);  noisy = bf16(bf16(sa x0) + bf16(so noise))
 *   get_velocity: out = bf16(bf16(sa noise) - bf16(so sample))      (the reference calls it as get_velocity(model_out, noisy, t))
 * The loss weight 1 / (1 - alphas_cumprod[t]) (utils/diffusion.py:125-128) goes through ftmi_mse_loss's per-sample weight.
 * ------------------------------------------------------------------------------------------------------------ */
int ftmi_ddim_add_noise(const void* latents, const void* noise, const float* sqrt_alpha, const float* sqrt_one_minus_alpha,
                        float scaling_factor, void* x0, void* noisy, int B, long per_sample, ftmi_stream stream);
int ftmi_ddim_get_velocity(const void* sample, const void* noise, const float* sqrt_alpha, const float* sqrt_one_minus_alpha, void* out,
                           int B, long per_sample, ftmi_stream stream);

/* ---- CogVideoX DiT block, row-wise stages (call sites finetrainers/models/cogvideox/base_specification.py:296-333, arithmetic
 * [upstream] diffusers CogVideoXBlock / CogVideoXLayerNormZero / Attention(qk_norm="layer_norm"), restated in oracle/cogvideox.py).
 * Tokens are one bf16 buffer [B, rows_per_batch, D] with the text_len text tokens of a sample first (the order the joint attention
 * concatenates them in); D % 64 == 0, D <= 4096.  text_len > 0: modulation tables are [B, 2, D] (row 0 text, row 1 video);
 * text_len == 0: [B, D].  Every tensor the eager bf16 graph materialises is one bf16 rounding here.  Only x-gradients exist: with LoRA on
 * the attention projections nothing upstream of the modulation / affine parameters is trainable. */
/* y = bf(bf(LayerNorm(x; w, b, eps)) * onep) + shift        (onep = bf(1 + scale), prepared by the caller like the reference's (1 + scale)) */
int ftmi_cog_ln_mod_fwd(const void* x, const void* w, const void* b, const void* shift, const void* onep, void* y, int rows, int D,
                        int rows_per_batch, int text_len, float eps, ftmi_stream stream);
/* dx = [dres +] LayerNorm'(x)[bf(dy * onep) * w]            (dres: gradient arriving on the residual branch, may be NULL) */
int ftmi_cog_ln_mod_bwd(const void* x, const void* w, const void* onep, const void* dy, const void* dres, void* dx, int rows, int D,
                        int rows_per_batch, int text_len, float eps, ftmi_stream stream);
/* q / k LayerNorm over each head's 64 channels (w, b: [64]); x, y, dy, dx: rows of D channels, row stride ld elements.
 * rope_cos / rope_sin (fp32 [S, 64], every frequency repeated twice; NULL for the sincos-table checkpoints): rotary embedding of the rotary
 * checkpoints (CogVideoX-5b), applied after the norm to the video rows only -- token position >= text_len within its rows_per_batch-token sample --
 * as diffusers' apply_rotary_emb(use_real=True, use_real_unbind_dim=-1) does on channel pairs (2i, 2i+1). */
int ftmi_cog_head_ln_fwd(const void* x, long ld, const void* w, const void* b, void* y, int rows, int D, float eps, const float* rope_cos,
                         const float* rope_sin, int rows_per_batch, int text_len, ftmi_stream stream);
int ftmi_cog_head_ln_bwd(const void* x, long ld, const void* w, const void* dy, void* dx, int rows, int D, float eps, const float* rope_cos,
                         const float* rope_sin, int rows_per_batch, int text_len, ftmi_stream stream);
/* out = res + bf(gate * y)   (res NULL: out = bf(gate * y), which is also the y-gradient of the same op) */
int ftmi_cog_gate_residual(const void* res, const void* y, const void* gate, void* out, int rows, int D, int rows_per_batch, int text_len,
                           ftmi_stream stream);

/* The whole CogVideoX block stack as one call per direction (the LTX orchestrator's design: caller-owned workspace, every activation of the backward
 * kept, nothing recomputed; the backward runs over block ranges so a data-parallel caller can all-reduce finished ranges while the rest computes).
 * Weights are stacked per kind along a leading layer axis, bf16; "_t" = transposed copies for the dgrads (made once at load time). */
typedef struct {
    int B, T, S;      /* batch, text tokens, video tokens: token buffers are [B, T + S, D], text first */
    int D, H, L;      /* width = H x 64, blocks */
    int D_ff, D_temb; /* feed-forward width, time-embedding width */
    int r;            /* LoRA rank (0: none; multiple of 64) on to_q, to_k, to_v, to_out.0 */
    float lora_scale, eps_norm, eps_qk;
    int gemm_variant; /* 8 */
} ftmi_cog_config;
typedef struct {
    const void *mod_w, *mod_b;   /* [L,2,6D,D_temb], [L,2,6D]: norm1.linear and norm2.linear of every block */
    const void *norm_w, *norm_b; /* [L,2,D]: the LayerNorm affine of norm1, norm2 */
    const void *w_qkv, *b_qkv;   /* [L,3D,D], [L,3D]: to_q | to_k | to_v */
    const void *w_o, *b_o;       /* [L,D,D], [L,D] */
    const void* qk_norm;         /* [L,4,64]: norm_q.weight, norm_q.bias, norm_k.weight, norm_k.bias */
    const void *w_ff1, *b_ff1, *w_ff2, *b_ff2;       /* [L,D_ff,D], [L,D_ff], [L,D,D_ff], [L,D] */
    const void *w_qkv_t, *w_o_t, *w_ff1_t, *w_ff2_t; /* [L,D,3D], [L,D,D], [L,D,D_ff], [L,D_ff,D] */
    /* bf16 (hi, lo) working copies of the fp32 adapters (ftmi_lora_refresh_n with 4 adapters per block), NULL when r == 0 */
    const void *lora_a_sp, *lora_bt_sp, *lora_b_ext, *lora_at_ext, *lora_at_qkv_ext; /* [L,4,2r,D] x2, [L,4,D,3r] x2, [L,D,9r] */
    const float *rope_cos, *rope_sin; /* fp32 [S,64] for the rotary checkpoints, NULL otherwise */
} ftmi_cog_weights;
size_t ftmi_cog_workspace_bytes(const ftmi_cog_config* cfg);
/* tokens_out <- blocks(tokens_in); temb_silu [B, D_temb] bf16 = silu(time embedding) */
int ftmi_cog_blocks_forward(const ftmi_cog_config* cfg, const ftmi_cog_weights* w, const void* tokens_in, const void* temb_silu, void* tokens_out,
                            void* workspace, size_t workspace_bytes, ftmi_stream stream);
/* blocks [l_lo, l_hi) of the backward, last block first; ranges in descending order share the workspace.  d_tokens_out: gradient of the stack's output
 * (read when l_hi == L).  d_tokens_in (may be NULL): gradient of the stack's input (written when l_lo == 0).  grad_a [L,4,r,D] / grad_b [L,4,D,r] fp32: the
 * range's slices are overwritten (accumulate = 0) or added to (1) and are final when the call returns (stream order). */
int ftmi_cog_blocks_backward(const ftmi_cog_config* cfg, const ftmi_cog_weights* w, const void* tokens_in, const void* d_tokens_out, void* d_tokens_in,
                             float* grad_a, float* grad_b, void* workspace, size_t workspace_bytes, int l_hi, int l_lo, int accumulate, ftmi_stream stream);

/* latents [B, F, C, H, W] -> tokens [B, F (H/p) (W/p), C p p] (the im2col of CogVideoXPatchEmbed's Conv2d(kernel = stride = p), channel order
 * (c, py, px) like the flattened conv weight), and back (the model's final un-patchify).  [upstream] CogVideoXPatchEmbed / CogVideoXTransformer3DModel. */
int ftmi_cog_patchify(const void* latents, void* tokens, int B, int F, int C, int H, int W, int patch, ftmi_stream stream);
int ftmi_cog_unpatchify(const void* tokens, void* latents, int B, int F, int C, int H, int W, int patch, ftmi_stream stream);

/* Precomputed-latent path (finetrainers/trainer/sft_trainer/trainer.py:374: --enable_precomputation => compute_posterior = False):
 * moments [B, 2, per_sample] bf16 = the VAE posterior (mean | logvar) as finetrainers-precomputed-data stores it, eps [B, per_sample] bf16
 * the N(0,1) draw; out = mean + exp(0.5 * clamp(logvar, -30, 20)) * eps, one bf16 rounding per torch op of
 * models/ltx_video/base_specification.py:285-289 ([upstream] diffusers DiagonalGaussianDistribution.sample). */
int ftmi_posterior_sample(const void* moments, const void* eps, void* out, int B, long per_sample, ftmi_stream stream);

/* loss (device fp32 scalar) = mean_b mean w_b (pred-target)^2 ; dpred = d(loss*grad_scale)/dpred (bf16), may be NULL.
 * scratch: caller-owned device memory, >= FTMI_MSE_SCRATCH_FLOATS_PER_SAMPLE * B floats (per-workgroup partial sums, added in a fixed order:
 * the loss is bitwise reproducible); the library allocates nothing on this path, so the call is legal inside a stream capture. */
#define FTMI_MSE_SCRATCH_FLOATS_PER_SAMPLE 256
int ftmi_mse_loss(const void* pred, const void* target, const float* weight, float* loss, void* dpred, int B, long per_sample,
                  float grad_scale, float* scratch, ftmi_stream stream);

/* Global L2 clip (max_norm <= 0 disables) + AdamW over flat fp32 buffers; scratch: >= FTMI_CLIP_SCRATCH_FLOATS floats (device).
 * The norm is reduced in a fixed order (block partials in scratch, last block adds them): the same gradients give the same
 * bits on every call and on every rank, so data-parallel replicas keep identical clip coefficients.
 * grad_norm_out (device fp32, may be NULL) receives the pre-clip total norm. */
#define FTMI_CLIP_SCRATCH_FLOATS 2050
int ftmi_clip_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long n, float max_norm, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int step, float* scratch, float* grad_norm_out,
                         ftmi_stream stream);

/* In-place global L2 clip of a flat fp32 gradient buffer: grads *= min(1, max_norm / (norm + 1e-6)) (finetrainers/utils/torch.py:99-161).
 * The reference loop clips after EVERY backward (trainer/sft_trainer/trainer.py:487-492), also on the micro-steps of a gradient-
 * accumulation window where no optimiser step follows; ftmi_clip_adamw_step covers the stepping micro-step, this call the others.
 * scratch: >= FTMI_CLIP_SCRATCH_FLOATS floats; grad_norm_out (may be NULL) receives the pre-clip norm; order-fixed reduction. */
int ftmi_clip_grad_norm(float* grads, long n, float max_norm, float* scratch, float* grad_norm_out, ftmi_stream stream);

/* HunyuanVideo ([upstream] diffusers Attention(qk_norm = "rms_norm") + HunyuanVideoAttnProcessor2_0, restated in oracle/hunyuan.py): q / k
 * RMSNorm over each head's head_dim channels (128; weight [head_dim] bf16, eps 1e-6, the reference's patched F.rms_norm: one bf16 rounding), then -- on
 * the rows at position >= rope_from of a sample of rows_per_batch tokens (the video tokens of a joint [text | video] sequence) -- the rotary embedding
 * in its real form: rope_cos / rope_sin fp32 [rows_per_batch - rope_from, head_dim], every frequency repeated for its channel pair; NULL: none.
 * x / y / dy / dx: [rows, D] bf16 views with their own row strides (multiples of 8).  Only the input gradient is produced (LoRA training). */
int ftmi_head_rms_rope_fwd(const void* x, long ld, const void* w, void* y, long ld_y, int rows, int D, int head_dim, float eps, const float* rope_cos,
                           const float* rope_sin, int rows_per_batch, int rope_from, ftmi_stream stream);
int ftmi_head_rms_rope_bwd(const void* x, long ld, const void* w, const void* dy, long ld_dy, void* dx, long ld_dx, int rows, int D, int head_dim, float eps,
                           const float* rope_cos, const float* rope_sin, int rows_per_batch, int rope_from, ftmi_stream stream);

/* HunyuanVideo single-stream block (40 of the 60 blocks of the DiT; [upstream] diffusers HunyuanVideoSingleTransformerBlock as driven by
 * finetrainers/models/hunyuan_video/base_specification.py:294-330, restated in oracle/hunyuan.py SingleStreamBlock) as ONE call per direction:
 *   tokens x [B, T + S, D] bf16 (text first), temb_silu [B, D] = silu(conditioning vector), key_bias fp32 [B, T + S] (-inf on padded text keys) or NULL,
 *   rope_cos / rope_sin fp32 [S, 128] (video rows), LoRA (fp32, rank r, scale alpha / r) on to_q / to_k / to_v.
 * Buffers are the caller's: `saved` (ftmi_hy_single_saved_bytes: what the backward reads again -- keep it from the forward to the backward of THIS block,
 * or rebuild it inside the backward by calling the forward with out = NULL, which is gradient checkpointing: utils/activation_checkpoint.py:24-49) and
 * `scratch` (ftmi_hy_single_scratch_bytes: transients, may be shared by all blocks on a stream).  The weights may be views into an arena that is
 * refilled per block (fp8 storage: ftmi_fp8_upcast).  The backward ADDS to grad_a [3, r, D] / grad_b [3, D, r] (fp32) and writes dx. */
typedef struct {
    int B, T, S;      /* batch, text tokens, video tokens */
    int D, H, mlp;    /* width = H x 128, MLP width (4 D) */
    int r;            /* LoRA rank: 0 or a multiple of 64 (smaller ranks zero-padded by the caller) */
    float lora_scale; /* alpha / (the user's) r */
    float eps;        /* 1e-6: LayerNorm and q / k RMSNorm */
    int gemm_variant; /* 8 */
} ftmi_hy_single_config;
typedef struct {
    const void *norm_lin_w, *norm_lin_b;                       /* [3D, D], [3D]   norm.linear */
    const void *proj_mlp_w, *proj_mlp_b;                       /* [mlp, D], [mlp] */
    const void *wq, *bq, *wk, *bk, *wv, *bv;                   /* [D, D], [D]     attn.to_q / to_k / to_v */
    const void *norm_q_w, *norm_k_w;                           /* [128]           attn.norm_q / norm_k */
    const void *proj_out_w, *proj_out_b;                       /* [D, D + mlp], [D] */
    const void *wq_t, *wk_t, *wv_t, *proj_mlp_w_t, *proj_out_w_t; /* transposed copies [D, D] x3, [D, mlp], [D + mlp, D]: backward only */
    const float *lora_a, *lora_b;                              /* fp32 [3, r, D], [3, D, r]; NULL when r == 0 */
    const void *ones, *zeros;                                  /* bf16 [D]: the affine-free LayerNorm's weight and bias */
} ftmi_hy_single_weights;
size_t ftmi_hy_single_saved_bytes(const ftmi_hy_single_config* cfg);
size_t ftmi_hy_single_scratch_bytes(const ftmi_hy_single_config* cfg);
/* out [B, T + S, D]; out == NULL: recomputation pass (fills `saved`, stops after the attention) */
int ftmi_hy_single_forward(const ftmi_hy_single_config* cfg, const ftmi_hy_single_weights* w, const void* x, const void* temb_silu, const float* key_bias,
                           const float* rope_cos, const float* rope_sin, void* out, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes,
                           ftmi_stream stream);
/* ones_rows: bf16 [B, D] of 1.0 (the bf16 accumulation of the four gradients of the normalised tokens runs through the gate-residual kernel) */
int ftmi_hy_single_backward(const ftmi_hy_single_config* cfg, const ftmi_hy_single_weights* w, const void* x, const void* dout, const float* key_bias,
                            const float* rope_cos, const float* rope_sin, const void* ones_rows, void* dx, float* grad_a, float* grad_b, void* saved,
                            size_t saved_bytes, void* scratch, size_t scratch_bytes, ftmi_stream stream);

/* HunyuanVideo dual-stream block (the other 20 blocks; [upstream] diffusers HunyuanVideoTransformerBlock, oracle/hunyuan.py DualStreamBlock), ONE SAMPLE per
 * call: video x_v [S, D] and text x_t [T, D] keep their own modulation (temb_silu [1, D]), projections, q / k norms and feed-forward and meet in one joint
 * attention over [text | video] (key_bias fp32 [T + S] or NULL; rope_cos / rope_sin fp32 [S, 128] on the video rows); LoRA (adapters 0..3) on the video
 * stream's to_q / to_k / to_v / to_out.0.  Buffers and gradient semantics as for the single-stream block; out_v == out_t == NULL: recomputation pass. */
typedef struct {
    int T, S;         /* text tokens, video tokens of the sample */
    int D, H, mlp;    /* width = H x 128, feed-forward width */
    int r;            /* LoRA rank: 0 or a multiple of 64 */
    float lora_scale, eps;
    int gemm_variant; /* 8 */
} ftmi_hy_dual_config;
typedef struct {
    const void *norm1_lin_w, *norm1_lin_b, *norm1c_lin_w, *norm1c_lin_b;              /* [6D, D], [6D]: norm1.linear, norm1_context.linear */
    const void *wq, *bq, *wk, *bk, *wv, *bv, *wo, *bo;                                /* video: attn.to_q / to_k / to_v / to_out.0 */
    const void *add_q_w, *add_q_b, *add_k_w, *add_k_b, *add_v_w, *add_v_b, *add_out_w, *add_out_b; /* text: attn.add_*_proj, attn.to_add_out */
    const void *norm_q_w, *norm_k_w, *norm_added_q_w, *norm_added_k_w;                /* [128] */
    const void *ff1_w, *ff1_b, *ff2_w, *ff2_b, *ffc1_w, *ffc1_b, *ffc2_w, *ffc2_b;    /* [mlp, D], [mlp], [D, mlp], [D]: ff, ff_context */
    const void *wq_t, *wk_t, *wv_t, *wo_t, *add_q_w_t, *add_k_w_t, *add_v_w_t, *add_out_w_t, *ff1_w_t, *ff2_w_t, *ffc1_w_t, *ffc2_w_t; /* transposed: backward only */
    const float *lora_a, *lora_b;                                                     /* fp32 [4, r, D], [4, D, r]; NULL when r == 0 */
    const void *ones, *zeros;                                                         /* bf16 [D] */
} ftmi_hy_dual_weights;
size_t ftmi_hy_dual_saved_bytes(const ftmi_hy_dual_config* cfg);
size_t ftmi_hy_dual_scratch_bytes(const ftmi_hy_dual_config* cfg);
int ftmi_hy_dual_forward(const ftmi_hy_dual_config* cfg, const ftmi_hy_dual_weights* w, const void* x_v, const void* x_t, const void* temb_silu,
                         const float* key_bias, const float* rope_cos, const float* rope_sin, void* out_v, void* out_t, void* saved, size_t saved_bytes,
                         void* scratch, size_t scratch_bytes, ftmi_stream stream);
/* ones_row: bf16 [D] of 1.0; grad_a [4, r, D] / grad_b [4, D, r] fp32 are ADDED to */
int ftmi_hy_dual_backward(const ftmi_hy_dual_config* cfg, const ftmi_hy_dual_weights* w, const void* x_v, const void* x_t, const void* dout_v, const void* dout_t,
                          const float* key_bias, const float* rope_cos, const float* rope_sin, const void* ones_row, void* dx_v, void* dx_t, float* grad_a,
                          float* grad_b, void* saved, size_t saved_bytes, void* scratch, size_t scratch_bytes, ftmi_stream stream);

/* ---- Wan-T2V full fine-tune (BASELINE config 4; finetrainers/models/wan/base_specification.py:433-493 driving [upstream]
 * diffusers transformer_wan.py; restated in oracle/wan.py).  Every parameter trains, so the backward kernels also produce the column sums the
 * parameter gradients need; they ADD to red1 / red2 (fp32, zeroed or kept by the caller; atomics: the summation order is not fixed).
 * One argument block for the seven row-wise launchers; rows = whole samples of rows_per_batch tokens, D a multiple of 64 (<= 4096, colsum: any),
 * row strides multiples of 8.
 *   ftmi_wan_ln_fwd        y = bf(LN(float(x)) [* w + b] [* (1 + scale_b) + shift_b])     FP32LayerNorm (+ affine | + modulation), one rounding
 *   ftmi_wan_ln_bwd        y = dx = bf([dres +] bf(LN'(x)[dy * (w | 1 + scale_b)]));  red1 += sum dy (d shift | d bias), red2 += sum dy * xhat
 *                          (d scale | d weight); red_per_batch = 1 for the modulated norms ([B, D] sums)
 *   ftmi_wan_rms_rope_fwd  n = bf(x * rstd(x) * w) over the WHOLE row (qk_norm = "rms_norm_across_heads"), y = bf(n rotated by rope_cos/sin
 *                          [rows_per_batch, head_dim / 2], the complex pair (2k, 2k+1) of every head times cos_k + i sin_k); rope null: y = n
 *   ftmi_wan_rms_rope_bwd  y = dx;  red2 += sum dn * xhat (d weight)
 *   ftmi_wan_gate_res_fwd  y = bf(float(x) + float(dy) * scale_b)   ("dy" = the branch output; scale = fp32 gate [B, D], null: bf(x + dy))
 *   ftmi_wan_gate_res_bwd  x = d out, dy = the branch output of the forward: y = bf(d out * scale_b), red1 += sum d out * branch (d gate)
 *   ftmi_wan_colsum        red1 += sum over rows of x   (Linear bias gradients) */
typedef struct ftmi_wan_row_args {
    const void* x; long ld_x;
    const void* w; const void* b;            /* bf16 [D] */
    const float* shift; const float* scale;  /* fp32 [B, mod_bstride] */
    long mod_bstride;
    const void* dy; long ld_dy;
    const void* dres;                        /* bf16, row stride ld_y */
    void* y; long ld_y;
    float* red1; float* red2; int red_per_batch;
    const float* rope_cos; const float* rope_sin; int head_dim;
    int rows, D, rows_per_batch;
    float eps;
} ftmi_wan_row_args;
int ftmi_wan_ln_fwd(const ftmi_wan_row_args* args, ftmi_stream stream);
int ftmi_wan_ln_bwd(const ftmi_wan_row_args* args, ftmi_stream stream);
int ftmi_wan_rms_rope_fwd(const ftmi_wan_row_args* args, ftmi_stream stream);
int ftmi_wan_rms_rope_bwd(const ftmi_wan_row_args* args, ftmi_stream stream);
int ftmi_wan_gate_res_fwd(const ftmi_wan_row_args* args, ftmi_stream stream);
int ftmi_wan_gate_res_bwd(const ftmi_wan_row_args* args, ftmi_stream stream);
int ftmi_wan_colsum(const ftmi_wan_row_args* args, ftmi_stream stream);

/* One Wan-T2V DiT block (the unit FSDP-2 shards, finetrainers/parallel/ptd.py:466-499) as ONE call per direction.  params: the block's flat bf16 parameter
 * buffer in the order of finetrainers_amd/wan/block.py WanBlockLayout (ftmi_wan_block_param_elements elements: attn1 q|k|v weights, their biases, attn1
 * to_out, attn1 norm_q / norm_k, attn2 to_q, attn2 k|v weights, their biases, attn2 to_out, attn2 norm_q / norm_k, norm2, ffn.net.0.proj, ffn.net.2,
 * scale_shift_table) -- the local buffer or the all-gathered one; grads: fp32, same layout, ADDED to (every entry except scale_shift_table, whose
 * gradient the caller forms from dmod).  x [B, S, D], enc [B, T, D] bf16; mod fp32 [B, 6, D] = scale_shift_table + time projection; rope_cos / rope_sin
 * fp32 [S, 64].  `saved` (ftmi_wan_block_saved_bytes) must live from the forward to the backward of the block; `scratch` (ftmi_wan_block_scratch_bytes,
 * backward only) may be shared by all blocks of a stream.  dmod fp32 [6, B, D] is ADDED to (per-sample column sums of d shift / d scale / d gate). */
typedef struct {
    int B, S, T;      /* batch, video tokens, text tokens */
    int D, H, F;      /* width = H x 128, feed-forward width */
    float eps;        /* 1e-6 */
    int gemm_variant; /* 8 */
} ftmi_wan_block_config;
size_t ftmi_wan_block_saved_bytes(const ftmi_wan_block_config* cfg);
size_t ftmi_wan_block_scratch_bytes(const ftmi_wan_block_config* cfg);
size_t ftmi_wan_block_param_elements(const ftmi_wan_block_config* cfg);
int ftmi_wan_block_forward(const ftmi_wan_block_config* cfg, const void* params, const void* x, const void* enc, const float* mod, const float* rope_cos,
                           const float* rope_sin, void* out, void* saved, size_t saved_bytes, ftmi_stream stream);
int ftmi_wan_block_backward(const ftmi_wan_block_config* cfg, const void* params, float* grads, const void* x, const void* enc, const float* mod,
                            const float* rope_cos, const float* rope_sin, const void* dout, void* dx, void* denc, float* dmod, void* saved, size_t saved_bytes,
                            void* scratch, size_t scratch_bytes, ftmi_stream stream);

/* Sum of squares of a flat fp32 gradient (shard): scratch[0] <- sum g^2 (order-fixed; scratch >= FTMI_CLIP_SCRATCH_FLOATS floats).  Sharded training
 * all-reduces scratch[0] over the ranks before the optimiser call below (the reference's clip_grad_norm_ over DTensor shards, utils/torch.py:99-161). */
int ftmi_grad_sumsq(const float* grads, long n, float* scratch, ftmi_stream stream);
/* In-place clip of a flat fp32 gradient (shard) by a global norm given as a device sum of squares: grads *= min(1, max_norm / (sqrt(*sumsq) + 1e-6)).
 * The non-stepping micro-steps of a gradient-accumulation window in sharded training (the reference clips after EVERY backward,
 * trainer/sft_trainer/trainer.py:487-492); the stepping micro-step clips inside ftmi_adamw_bf16_step. */
int ftmi_clip_by_sumsq(float* grads, long n, const float* sumsq, float max_norm, float* grad_norm_out, ftmi_stream stream);
/* torch.optim.AdamW on bf16 parameters with bf16 moments (the reference's bf16 full fine-tune, optimizer.py:17-46): every torch op of the update is one
 * fp32 computation rounded to bf16.  grads: fp32 (the reduce-scattered shard), multiplied by min(1, max_norm / (sqrt(*sumsq) + 1e-6)) and rounded to
 * bf16 first (sumsq NULL: no clip).  grad_norm_out (may be NULL) receives sqrt(*sumsq). */
int ftmi_adamw_bf16_step(void* params, const float* grads, void* exp_avg, void* exp_avg_sq, long n, const float* sumsq, float max_norm, float lr,
                         float beta1, float beta2, float eps, float weight_decay, int step, float* grad_norm_out, ftmi_stream stream);

/* fp32 flat LoRA params (A region [L,8,r,D] then B region [L,8,D,r]) -> the bf16 (hi, lo) working copies of ftmi_ltx_weights */
int ftmi_lora_refresh(const float* a_f32, const float* b_f32, void* lora_a_sp, void* lora_bt_sp, void* lora_b_ext, void* lora_at_ext,
                      void* lora_at_qkv_ext, int L, int r, int D, ftmi_stream stream);
/* the same for a model with n_adapters adapters per block, the first three being q, k, v (CogVideoX: 4) */
int ftmi_lora_refresh_n(const float* a_f32, const float* b_f32, void* lora_a_sp, void* lora_bt_sp, void* lora_b_ext, void* lora_at_ext,
                        void* lora_at_qkv_ext, int L, int n_adapters, int r, int D, ftmi_stream stream);

/* The same split for ONE fp32 matrix w [rows, cols] (building block of ftmi_linear_lora_fwd/_bwd callers): any of the four outputs
 * may be NULL.  sp [2 rows, cols]: (hi, lo) row planes interleaved per 32 rows; ext [rows, 3 cols]: [hi | hi | lo];
 * t_sp [2 cols, rows], t_ext [cols, 3 rows]: the same two layouts of w^T. */
int ftmi_lora_split(const float* w, int rows, int cols, void* sp, void* ext, void* t_sp, void* t_ext, ftmi_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* FTMI355_H */
