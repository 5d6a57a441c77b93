"""HunyuanVideo single-stream DiT block (40 of the model's 60 blocks) for LoRA SFT on the MI355X -- first piece of SURVEY 8f-4 / BASELINE config 5.

Reference: [upstream] diffusers ``HunyuanVideoSingleTransformerBlock`` + ``HunyuanVideoAttnProcessor2_0`` as driven by
finetrainers/models/hunyuan_video/base_specification.py:294-330, restated in oracle/hunyuan.py (``SingleStreamBlock``).  One autograd Function, everything
inside a call through the C ABI: AdaLN-zero-single modulation (``ftmi_cog_ln_mod_*`` with one table row per sample), the MLP branch (GELU-tanh GEMM
epilogue) and q / k / v (fused LoRA GEMMs) off the same normalised tokens, per-head RMSNorm + rotary embedding on the video rows
(``ftmi_head_rms_rope_*``), head_dim-128 attention over the joint sequence with the padded text keys masked by a per-sample key bias, ONE output GEMM over
the concatenated [attention | MLP] features, gated residual.

Token layout: ONE buffer [B, T + S, D] with the T text tokens of a sample FIRST (the order the CogVideoX kernels use; the reference concatenates
[video | text] -- attention and every row-wise stage are invariant to the order of the tokens, the rotary embedding follows the video rows).
LoRA on to_q / to_k / to_v (the default target regex, sft_trainer/config.py:24-26, matches nothing else in this block).  The frozen weights may hold
fp8-representable values (config 5's layerwise casting is storage-only: the arithmetic is the same)."""

from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from .. import ops

bf16 = torch.bfloat16


class _SingleBlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk: "MI355XHunyuanSingleBlock", x, temb_silu, key_bias, rope_cos, rope_sin, text_len, lora_a, lora_b):
        B, N, D = x.shape
        M, H, hd, T, s = B * N, blk.heads, 128, int(text_len), blk.lora_scale
        mod = ops.gemm_nt(temb_silu, blk.norm_lin_w, blk.norm_lin_b).view(B, 3, D)  # shift, scale, gate
        shift, onep, gate = mod[:, 0].contiguous(), (1 + mod[:, 1]).contiguous(), mod[:, 2].contiguous()
        n = ops.cog_ln_mod(x, blk.ones, blk.zeros, shift, onep, 0, 1e-6)
        n2d = n.view(M, D)
        cat = torch.empty((M, D + blk.mlp_dim), dtype=bf16, device=x.device)  # [attention | MLP] features, the input of proj_out
        _, pre = ops.gemm_nt(n2d, blk.proj_mlp_w, blk.proj_mlp_b, epilogue=1, want_out2=True, out=cat[:, D:])  # GELU-tanh, pre-activation kept
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        q, xa_q = ops.linear_lora_fwd(n2d, blk.wq, blk.bq, A(0), Bm(0), s)
        k, xa_k = ops.linear_lora_fwd(n2d, blk.wk, blk.bk, A(1), Bm(1), s)
        v, xa_v = ops.linear_lora_fwd(n2d, blk.wv, blk.bv, A(2), Bm(2), s)
        rope = (rope_cos, rope_sin)
        qn = ops.head_rms_rope(q, blk.norm_q_w, hd, 1e-6, rope=rope, rows_per_batch=N, rope_from=T)
        kn = ops.head_rms_rope(k, blk.norm_k_w, hd, 1e-6, rope=rope, rows_per_batch=N, rope_from=T)
        heads = lambda t: t.view(B, N, H, hd).permute(0, 2, 1, 3)
        o, lse = ops.attn_fwd(heads(qn), heads(kn), heads(v), key_bias)
        cat[:, :D].copy_(o.permute(0, 2, 1, 3).reshape(M, D))
        y = ops.gemm_nt(cat, blk.proj_out_w, blk.proj_out_b)
        out = ops.cog_gate_residual(x, y.view(B, N, D), gate, 0)
        ctx.blk, ctx.T, ctx.rope, ctx.key_bias, ctx.has_lora = blk, T, rope, key_bias, lora_a is not None
        ctx.save_for_backward(x, n, q, k, qn, kn, v, o, lse, pre, cat, onep, gate, xa_q, xa_k, xa_v,
                              lora_a if lora_a is not None else x.new_empty(0), lora_b if lora_b is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        blk, T, rope = ctx.blk, ctx.T, ctx.rope
        x, n, q, k, qn, kn, v, o, lse, pre, cat, onep, gate, xa_q, xa_k, xa_v, lora_a, lora_b = ctx.saved_tensors
        if not ctx.has_lora:
            lora_a = lora_b = None
        B, N, D = x.shape
        M, H, hd, s = B * N, blk.heads, 128, blk.lora_scale
        dout = dout.contiguous()
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        ga = torch.zeros_like(lora_a) if lora_a is not None else None
        gb = torch.zeros_like(lora_b) if lora_b is not None else None
        GA = lambda i: None if ga is None else ga[i]
        GB = lambda i: None if gb is None else gb[i]
        dy = ops.cog_gate_residual(None, dout, gate, 0).view(M, D)  # d proj_out output = gate * d out
        wt = blk.proj_out_w_t  # [D + mlp, D]
        do = ops.gemm_nt(dy, wt[:D], None)                             # gradient of the attention features
        dpre = torch.empty_like(cat)[:, D:]                             # same row stride as pre (a view like the forward's MLP features)
        ops.gemm_nt(dy, wt[D:], None, epilogue=3, aux=pre, out=dpre)   # (gradient of the MLP features) * gelu'(pre)
        dn_mlp = ops.gemm_nt(dpre, blk.proj_mlp_w_t, None)
        heads = lambda t: t.view(B, N, H, hd).permute(0, 2, 1, 3)
        flat = lambda t: t.permute(0, 2, 1, 3).reshape(M, D)
        dqn, dkn, dv = ops.attn_bwd(heads(qn), heads(kn), heads(v), o, lse, heads(do), ctx.key_bias)
        dq = ops.head_rms_rope_bwd(q, blk.norm_q_w, flat(dqn), hd, 1e-6, rope=rope, rows_per_batch=N, rope_from=T)
        dk = ops.head_rms_rope_bwd(k, blk.norm_k_w, flat(dkn), hd, 1e-6, rope=rope, rows_per_batch=N, rope_from=T)
        n2d = n.view(M, D)
        dn_q, _, _ = ops.linear_lora_bwd(n2d, dq, xa_q, blk.wq_t, A(0), Bm(0), s, GA(0), GB(0))
        dn_k, _, _ = ops.linear_lora_bwd(n2d, dk, xa_k, blk.wk_t, A(1), Bm(1), s, GA(1), GB(1))
        dn_v, _, _ = ops.linear_lora_bwd(n2d, flat(dv), xa_v, blk.wv_t, A(2), Bm(2), s, GA(2), GB(2))
        # the four consumers of the normalised tokens: their gradients add as bf16 tensors (autograd's accumulation)
        ones = blk.ones_rows(B, x.device)
        dn = ops.cog_gate_residual(dn_mlp.view(B, N, D), dn_v.view(B, N, D), ones, 0)
        dn = ops.cog_gate_residual(dn, dn_k.view(B, N, D), ones, 0)
        dn = ops.cog_gate_residual(dn, dn_q.view(B, N, D), ones, 0)
        dx = ops.cog_ln_mod_bwd(x, blk.ones, onep, dn, 0, 1e-6, dres=dout)
        return None, dx, None, None, None, None, None, ga, gb


class MI355XHunyuanSingleBlock(nn.Module):
    """Frozen bf16 weights (+ the transposes the input-gradient GEMMs use, made once) and the fp32 LoRA adapters of to_q / to_k / to_v."""

    _KEYS = {  # diffusers HunyuanVideoSingleTransformerBlock parameter name -> buffer
        "norm.linear.weight": "norm_lin_w", "norm.linear.bias": "norm_lin_b", "proj_mlp.weight": "proj_mlp_w", "proj_mlp.bias": "proj_mlp_b",
        "attn.to_q.weight": "wq", "attn.to_q.bias": "bq", "attn.to_k.weight": "wk", "attn.to_k.bias": "bk", "attn.to_v.weight": "wv", "attn.to_v.bias": "bv",
        "attn.norm_q.weight": "norm_q_w", "attn.norm_k.weight": "norm_k_w", "proj_out.weight": "proj_out_w", "proj_out.bias": "proj_out_b",
    }

    def __init__(self, dim: int = 3072, heads: int = 24, mlp_ratio: float = 4.0, device: Optional[torch.device] = None):
        super().__init__()
        if dim != heads * 128:
            raise ValueError("HunyuanVideo blocks have heads of 128 channels")
        self.dim, self.heads, self.mlp_dim = dim, heads, int(dim * mlp_ratio)
        dev = device or torch.device("cuda", 0)
        z = lambda *shape: torch.zeros(shape, dtype=bf16, device=dev)
        for name, shape in (("norm_lin_w", (3 * dim, dim)), ("norm_lin_b", (3 * dim,)), ("proj_mlp_w", (self.mlp_dim, dim)), ("proj_mlp_b", (self.mlp_dim,)),
                            ("wq", (dim, dim)), ("bq", (dim,)), ("wk", (dim, dim)), ("bk", (dim,)), ("wv", (dim, dim)), ("bv", (dim,)),
                            ("norm_q_w", (128,)), ("norm_k_w", (128,)), ("proj_out_w", (dim, dim + self.mlp_dim)), ("proj_out_b", (dim,))):
            self.register_buffer(name, z(*shape))
        for name in ("wq_t", "wk_t", "wv_t", "proj_mlp_w_t", "proj_out_w_t"):
            self.register_buffer(name, None, persistent=False)
        self.register_buffer("ones", torch.ones(dim, dtype=bf16, device=dev), persistent=False)   # LayerNorm(elementwise_affine=False)
        self.register_buffer("zeros", torch.zeros(dim, dtype=bf16, device=dev), persistent=False)
        self.lora_A: Optional[nn.Parameter] = None  # [3, r, D]
        self.lora_B: Optional[nn.Parameter] = None  # [3, D, r]
        self.lora_scale = 0.0
        self._ones_rows: Dict[int, torch.Tensor] = {}

    def ones_rows(self, B: int, dev) -> torch.Tensor:
        if B not in self._ones_rows:
            self._ones_rows[B] = torch.ones(B, self.dim, dtype=bf16, device=dev)
        return self._ones_rows[B]

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items()}
        missing = [k for k in self._KEYS if k not in sd]
        if missing:
            raise KeyError(f"HunyuanVideo single-stream block state dict lacks {missing[:4]}")
        for k, name in self._KEYS.items():
            getattr(self, name).copy_(sd[k].to(bf16))
        for name in ("wq", "wk", "wv", "proj_mlp_w", "proj_out_w"):
            setattr(self, name + "_t", ops.transpose_bf16(getattr(self, name)))

    def add_adapter(self, r: int = 64, lora_alpha: float = 64.0) -> None:
        if r % 64 != 0:
            raise ValueError("ranks that are multiples of 64 (the LTX model shows the zero-padding route for the others)")
        dev, D = self.wq.device, self.dim
        a = torch.empty(3, r, D, dtype=torch.float32, device=dev).uniform_(-(1.0 / D) ** 0.5, (1.0 / D) ** 0.5)  # kaiming_uniform_(a = sqrt(5))
        self.lora_A, self.lora_B = nn.Parameter(a), nn.Parameter(torch.zeros(3, D, r, dtype=torch.float32, device=dev))
        self.lora_scale = float(lora_alpha) / r

    def forward(self, tokens: torch.Tensor, temb: torch.Tensor, text_len: int, image_rotary_emb, text_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``tokens`` [B, T + S, D] bf16 (text first), ``temb`` [B, D] the conditioning vector, ``image_rotary_emb`` = (cos, sin) fp32 [S, 128],
        ``text_mask`` [B, T] (1 = real token; None: all real) -> the block's output tokens in the same layout."""
        if self.wq_t is None:
            raise RuntimeError("load_diffusers_state_dict first (it also builds the transposed weights the input-gradient GEMMs use)")
        B, N, _ = tokens.shape
        key_bias = None
        if text_mask is not None:  # padded text tokens are never attended to (the reference masks every key beyond a sample's real text length)
            key_bias = torch.zeros((B, N), dtype=torch.float32, device=tokens.device)
            key_bias[:, :text_len].masked_fill_(~text_mask.to(tokens.device).bool(), float("-inf"))
        temb_silu = torch.nn.functional.silu(temb.to(bf16)).contiguous()
        cos, sin = image_rotary_emb
        return _SingleBlockFunction.apply(self, tokens.contiguous(), temb_silu, key_bias, cos.contiguous(), sin.contiguous(), int(text_len), self.lora_A, self.lora_B)
