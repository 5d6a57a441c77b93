"""One CogVideoX DiT block (LoRA SFT: forward and the backward that yields dx and the LoRA gradients) on the gfx950 kernels.

Reference: [upstream] diffusers ``CogVideoXBlock`` (CogVideoXLayerNormZero -> joint text+video attention with per-head q/k LayerNorm ->
gated residual -> CogVideoXLayerNormZero -> GELU-tanh feed-forward over the concatenated tokens -> gated residual), as driven by
``finetrainers/models/cogvideox/base_specification.py:296-333`` and restated op by op in ``oracle/cogvideox.py`` (``CogVideoXBlock``).

First cut of SURVEY 8f-1's block: the kernels are the product (MFMA GEMMs with fused LoRA / GELU / GELU' epilogues, flash attention over
the 226 + 17 550 joint tokens, the row-wise stages of ``csrc/cogvideox.hip``); the ORCHESTRATION is still Python -- ~35 C-ABI calls per
block and direction -- where the LTX path has a C++ orchestrator with a caller-owned workspace (that and the fused q|k|v projection are the
next steps).  Both checkpoint families are covered: sincos-table (2b) and rotary (5b: RoPE on the video rows of q / k, fused into the head norm).  Tokens live in one buffer ``[B, T + S, D]``, text first: the joint attention's own order,
so nothing is ever concatenated or split.  The only torch ops are on per-sample conditioning vectors (``silu(temb)`` [B, 512], ``1 + scale``
[B, 2, D]) -- no token-sized tensor is touched outside a kernel.
"""

from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops

bf16 = torch.bfloat16
LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


class _BlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk: "MI355XCogVideoXBlock", x, temb_silu, text_len, lora_a, lora_b, rope_cos=None, rope_sin=None):
        B, N, D = x.shape
        M, H, T = B * N, blk.heads, int(text_len)
        s = blk.lora_scale
        # CogVideoXLayerNormZero: (shift, scale, gate, enc_shift, enc_scale, enc_gate) = linear(silu(temb)).chunk(6)  ->  [B, 2, D] tables, text row first
        def tables(w, b):
            m = ops.gemm_nt(temb_silu, w, b).view(B, 6, D)
            return (torch.stack([m[:, 3], m[:, 0]], 1).contiguous(), (1 + torch.stack([m[:, 4], m[:, 1]], 1)).contiguous(),
                    torch.stack([m[:, 5], m[:, 2]], 1).contiguous())
        shift1, onep1, gate1 = tables(blk.norm1_lin_w, blk.norm1_lin_b)
        shift2, onep2, gate2 = tables(blk.norm2_lin_w, blk.norm2_lin_b)

        n1 = ops.cog_ln_mod(x, blk.norm1_w, blk.norm1_b, shift1, onep1, T, blk.norm_eps)
        n1_2d = n1.view(M, D)
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        q, xa_q = ops.linear_lora_fwd(n1_2d, blk.wq, blk.bq, A(0), Bm(0), s)
        k, xa_k = ops.linear_lora_fwd(n1_2d, blk.wk, blk.bk, A(1), Bm(1), s)
        v, xa_v = ops.linear_lora_fwd(n1_2d, blk.wv, blk.bv, A(2), Bm(2), s)
        rope = None if rope_cos is None else (rope_cos, rope_sin)  # rotary checkpoints: applied to the video rows of q and k after the norm
        qn = ops.cog_head_ln(q, blk.norm_q_w, blk.norm_q_b, blk.qk_eps, rope=rope, rows_per_batch=N, text_len=T)
        kn = ops.cog_head_ln(k, blk.norm_k_w, blk.norm_k_b, blk.qk_eps, rope=rope, rows_per_batch=N, text_len=T)
        heads = lambda t: t.view(B, N, H, 64).permute(0, 2, 1, 3)
        o, lse = ops.attn_fwd(heads(qn), heads(kn), heads(v))
        o_2d = o.permute(0, 2, 1, 3).reshape(M, D)  # a view: the kernel wrote [B, N, H, 64]
        ao, xa_o = ops.linear_lora_fwd(o_2d, blk.wo, blk.bo, A(3), Bm(3), s)
        h1 = ops.cog_gate_residual(x, ao.view(B, N, D), gate1, T)

        n2 = ops.cog_ln_mod(h1, blk.norm2_w, blk.norm2_b, shift2, onep2, T, blk.norm_eps)
        act, pre = ops.gemm_nt(n2.view(M, D), blk.ff1_w, blk.ff1_b, epilogue=1, want_out2=True)  # GELU-tanh, pre-activation kept for GELU'
        f = ops.gemm_nt(act, blk.ff2_w, blk.ff2_b)
        out = ops.cog_gate_residual(h1, f.view(B, N, D), gate2, T)

        ctx.blk, ctx.T, ctx.rope = blk, T, rope
        ctx.has_lora = lora_a is not None
        ctx.save_for_backward(x, n1, q, k, qn, kn, v, o, lse, h1, pre, onep1, gate1, onep2, gate2, xa_q, xa_k, xa_v, xa_o,
                              lora_a if lora_a is not None else x.new_empty(0), lora_b if lora_b is not None else x.new_empty(0))
        return out

    @staticmethod
    def backward(ctx, dout):
        blk, T = ctx.blk, ctx.T
        x, n1, q, k, qn, kn, v, o, lse, h1, pre, onep1, gate1, onep2, gate2, xa_q, xa_k, xa_v, xa_o, lora_a, lora_b = ctx.saved_tensors
        if not ctx.has_lora:
            lora_a = lora_b = None
        B, N, D = x.shape
        M, H, s = B * N, blk.heads, blk.lora_scale
        dout = dout.contiguous()
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        # LoRA gradients: inside a model they go straight into this block's slice of the model-wide flat gradient buffer, which IS lora_A.grad /
        # lora_B.grad (added to when a gradient is already there, i.e. under gradient accumulation) -- one flat buffer for the bucketed all-reduce
        # and the fused clip + AdamW, and nothing for autograd to copy.  A stand-alone block hands fresh tensors to autograd instead.
        own = lora_a is not None and blk._grad_a_view is not None
        if own:
            ga, gb = blk._grad_a_view, blk._grad_b_view
            if blk.lora_A.grad is None or blk.lora_A.grad.data_ptr() != ga.data_ptr():
                ga.zero_()
                gb.zero_()
        else:
            ga = torch.zeros_like(lora_a) if lora_a is not None else None
            gb = torch.zeros_like(lora_b) if lora_b is not None else None
        GA = lambda i: None if ga is None else ga[i]
        GB = lambda i: None if gb is None else gb[i]

        # feed-forward branch
        df = ops.cog_gate_residual(None, dout, gate2, T)                                   # d f = gate * d out
        dact = ops.gemm_nt(df.view(M, D), blk.ff2_w_t, None, epilogue=3, aux=pre)          # (d f W2) * gelu'(pre)
        dn2 = ops.gemm_nt(dact, blk.ff1_w_t, None)
        dh1 = ops.cog_ln_mod_bwd(h1, blk.norm2_w, onep2, dn2.view(B, N, D), T, blk.norm_eps, dres=dout)
        # attention branch
        dao = ops.cog_gate_residual(None, dh1, gate1, T)
        o_2d = o.permute(0, 2, 1, 3).reshape(M, D)
        do, _, _ = ops.linear_lora_bwd(o_2d, dao.view(M, D), xa_o, blk.wo_t, A(3), Bm(3), s, GA(3), GB(3))
        heads = lambda t: t.view(B, N, H, 64).permute(0, 2, 1, 3)
        dqn, dkn, dv = ops.attn_bwd(heads(qn), heads(kn), heads(v), o, lse, heads(do))
        flat = lambda t: t.permute(0, 2, 1, 3).reshape(M, D)
        dq = ops.cog_head_ln_bwd(q, blk.norm_q_w, flat(dqn), blk.qk_eps, rope=ctx.rope, rows_per_batch=N, text_len=T)
        dk = ops.cog_head_ln_bwd(k, blk.norm_k_w, flat(dkn), blk.qk_eps, rope=ctx.rope, rows_per_batch=N, text_len=T)
        n1_2d = n1.view(M, D)
        dn_q, _, _ = ops.linear_lora_bwd(n1_2d, dq, xa_q, blk.wq_t, A(0), Bm(0), s, GA(0), GB(0))
        dn_k, _, _ = ops.linear_lora_bwd(n1_2d, dk, xa_k, blk.wk_t, A(1), Bm(1), s, GA(1), GB(1))
        dn_v, _, _ = ops.linear_lora_bwd(n1_2d, flat(dv), xa_v, blk.wv_t, A(2), Bm(2), s, GA(2), GB(2))
        # the three projections read the same tensor: their input gradients add (bf16 adds, like autograd's accumulation), through the
        # residual kernel with a gate of ones
        ones = blk._ones(B, D, x.device)
        dn1 = ops.cog_gate_residual(dn_v.view(B, N, D), dn_k.view(B, N, D), ones, 0)
        dn1 = ops.cog_gate_residual(dn1, dn_q.view(B, N, D), ones, 0)
        dx = ops.cog_ln_mod_bwd(x, blk.norm1_w, onep1, dn1, T, blk.norm_eps, dres=dh1)
        if own:
            blk.lora_A.grad, blk.lora_B.grad = ga, gb
            if blk._grad_hook is not None:
                blk._grad_hook(ga, gb)  # data parallelism: this block's LoRA gradients are final -- start averaging them while the earlier blocks run
            return None, dx, None, None, None, None, None, None
        return None, dx, None, None, ga, gb, None, None


class MI355XCogVideoXBlock(nn.Module):
    """Frozen bf16 block weights (+ their transposes for the dgrads, made once) and the fp32 LoRA adapters of to_q / to_k / to_v / to_out.0."""

    def __init__(self, dim: int = 1920, heads: int = 30, time_embed_dim: int = 512, ff_mult: int = 4, norm_eps: float = 1e-5,
                 device: Optional[torch.device] = None):
        super().__init__()
        if dim != heads * 64:
            raise ValueError("the gfx950 attention kernels need head_dim 64")
        self.dim, self.heads, self.norm_eps, self.qk_eps = dim, heads, norm_eps, 1e-6
        dev = device or torch.device("cuda", 0)
        z = lambda *shape: torch.zeros(shape, dtype=bf16, device=dev)
        for name, shape in (("norm1_lin_w", (6 * dim, time_embed_dim)), ("norm1_lin_b", (6 * dim,)), ("norm1_w", (dim,)), ("norm1_b", (dim,)),
                            ("norm2_lin_w", (6 * dim, time_embed_dim)), ("norm2_lin_b", (6 * dim,)), ("norm2_w", (dim,)), ("norm2_b", (dim,)),
                            ("wq", (dim, dim)), ("bq", (dim,)), ("wk", (dim, dim)), ("bk", (dim,)), ("wv", (dim, dim)), ("bv", (dim,)),
                            ("wo", (dim, dim)), ("bo", (dim,)), ("norm_q_w", (64,)), ("norm_q_b", (64,)), ("norm_k_w", (64,)), ("norm_k_b", (64,)),
                            ("ff1_w", (ff_mult * dim, dim)), ("ff1_b", (ff_mult * dim,)), ("ff2_w", (dim, ff_mult * dim)), ("ff2_b", (dim,))):
            self.register_buffer(name, z(*shape))
        for name in ("wq_t", "wk_t", "wv_t", "wo_t", "ff1_w_t", "ff2_w_t"):
            self.register_buffer(name, None, persistent=False)
        self.lora_A: Optional[nn.Parameter] = None  # [4, r, D]
        self.lora_B: Optional[nn.Parameter] = None  # [4, D, r]
        self.lora_scale = 0.0
        self._ones_cache: Dict[Tuple[int, int], torch.Tensor] = {}
        self._grad_hook = None  # callable(grad_a, grad_b), set by the data-parallel step
        self._grad_a_view = self._grad_b_view = None  # this block's slices of the model's flat gradient buffer

    _KEYS = {  # diffusers CogVideoXBlock parameter name -> buffer
        "norm1.linear.weight": "norm1_lin_w", "norm1.linear.bias": "norm1_lin_b", "norm1.norm.weight": "norm1_w", "norm1.norm.bias": "norm1_b",
        "norm2.linear.weight": "norm2_lin_w", "norm2.linear.bias": "norm2_lin_b", "norm2.norm.weight": "norm2_w", "norm2.norm.bias": "norm2_b",
        "attn1.to_q.weight": "wq", "attn1.to_q.bias": "bq", "attn1.to_k.weight": "wk", "attn1.to_k.bias": "bk",
        "attn1.to_v.weight": "wv", "attn1.to_v.bias": "bv", "attn1.to_out.0.weight": "wo", "attn1.to_out.0.bias": "bo",
        "attn1.norm_q.weight": "norm_q_w", "attn1.norm_q.bias": "norm_q_b", "attn1.norm_k.weight": "norm_k_w", "attn1.norm_k.bias": "norm_k_b",
        "ff.net.0.proj.weight": "ff1_w", "ff.net.0.proj.bias": "ff1_b", "ff.net.2.weight": "ff2_w", "ff.net.2.bias": "ff2_b",
    }

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """``sd``: a diffusers ``CogVideoXBlock`` state dict (peft's ``.base_layer.`` infix accepted)."""
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items()}
        missing = [k for k in self._KEYS if k not in sd]
        if missing:
            raise KeyError(f"CogVideoX block state dict lacks {missing[:4]}")
        for k, name in self._KEYS.items():
            getattr(self, name).copy_(sd[k].to(bf16))
        for name in ("wq", "wk", "wv", "wo", "ff1_w", "ff2_w"):
            setattr(self, name + "_t", ops.transpose_bf16(getattr(self, name)))

    def add_adapter(self, r: int = 64, lora_alpha: float = 64.0, storage_a: Optional[torch.Tensor] = None, storage_b: Optional[torch.Tensor] = None) -> None:
        """``storage_a`` [4, rp, D] / ``storage_b`` [4, D, rp] fp32 (rp = r rounded up to a multiple of 64): views of a model-wide flat buffer (so one
        fused clip + AdamW launch covers every block); allocated here when absent.
        Ranks that are not multiples of 64 (one MFMA K-extension step) are stored zero-padded to the next multiple: the extra rows of A and
        columns of B are zero and STAY zero (their gradients are exact zeros -- x A_pad^T = 0 and dY B_pad = 0 -- and AdamW / weight decay leave a zero
        parameter with zero gradient at zero), so the padded adapter IS the rank-r adapter; the LoRA scale is alpha / r of the USER's rank, and
        ``lora_state_dict`` / ``lora_grad_state_dict`` / the saved file carry the user's rank."""
        if r <= 0:
            raise ValueError(f"LoRA rank must be positive, got {r}")
        rp = -(-int(r) // 64) * 64
        dev, D = self.wq.device, self.dim
        a = torch.empty(4, rp, D, dtype=torch.float32, device=dev) if storage_a is None else storage_a
        b = torch.empty(4, D, rp, dtype=torch.float32, device=dev) if storage_b is None else storage_b
        if a.shape != (4, rp, D) or b.shape != (4, D, rp) or not a.is_contiguous() or not b.is_contiguous():
            raise ValueError("adapter storage must be contiguous [4, rp, D] / [4, D, rp] fp32 (rp = rank rounded up to a multiple of 64)")
        with torch.no_grad():
            a.zero_()
            a[:, :r].uniform_(-(1.0 / D) ** 0.5, (1.0 / D) ** 0.5)  # kaiming_uniform_(a = sqrt(5)) on [r, D]
            b.zero_()
        self.lora_A = nn.Parameter(a)
        self.lora_B = nn.Parameter(b)
        self.lora_rank_user = int(r)
        self.lora_scale = float(lora_alpha) / r

    def _ones(self, B: int, D: int, dev) -> torch.Tensor:
        key = (B, D)
        if key not in self._ones_cache:
            self._ones_cache[key] = torch.ones(B, D, dtype=bf16, device=dev)
        return self._ones_cache[key]

    def forward(self, tokens: torch.Tensor, temb: torch.Tensor, text_len: int, image_rotary_emb=None) -> torch.Tensor:
        """``tokens`` [B, T + S, D] bf16 (text first), ``temb`` [B, time_embed_dim] bf16 -> the block's output tokens in the same layout
        (``[:, :T]`` = encoder_hidden_states, ``[:, T:]`` = hidden_states of the reference block)."""
        if self.wq_t is None:
            raise RuntimeError("load_diffusers_state_dict first (it also builds the transposed weights the dgrads use)")
        temb_silu = torch.nn.functional.silu(temb.to(bf16)).contiguous()
        cos, sin = (None, None) if image_rotary_emb is None else image_rotary_emb
        return _BlockFunction.apply(self, tokens.contiguous(), temb_silu, int(text_len), self.lora_A, self.lora_B, cos, sin)
