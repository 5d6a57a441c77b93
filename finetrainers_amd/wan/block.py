"""Wan-T2V DiT block for FULL fine-tuning on the MI355X (BASELINE config 4): every parameter of the block trains.

Reference: [upstream] diffusers ``WanTransformerBlock`` / ``WanAttnProcessor2_0`` as driven by finetrainers/models/wan/base_specification.py:433-493,
restated in oracle/wan.py (whose rounding points the kernels follow).  One ``torch.autograd.Function`` per block; inside it everything is a call through
the C ABI (include/ftmi355.h): the MFMA GEMM (``ftmi_gemm_nt``; q|k|v of the self-attention and k|v of the cross-attention are one GEMM each), the
head_dim-128 flash attention, the Wan row-wise kernels (``ftmi_wan_*``: FP32LayerNorm + modulation, RMSNorm across heads + rotary embedding, gated
residual, each backward also producing the column sums of its parameter gradients) and the token-reduction GEMM for the weight gradients
(``ftmi_gemm_tn``: dW += dY^T X, accumulated in fp32 straight into the block's flat gradient buffer).

Parameters: ONE flat bf16 buffer per block (``flat``; the unit FSDP-2 shards, all-gathers and reduce-scatters -- parallel/ptd.py:466-499 wraps each block
with ``fully_shard``), gradients one flat fp32 buffer of the same layout (``grad_flat``; the reference reduces gradients in fp32, trainer.py:176-180).
``WanBlockLayout`` names the slices with the diffusers parameter names."""

from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

import ctypes
import os

from .. import _lib, ops
from .._lib import WanBlockConfig, check, ptr, stream_ptr

bf16 = torch.bfloat16
_NATIVE_SCRATCH: Dict[int, torch.Tensor] = {}  # device index -> byte buffer shared by every natively run block on that device (one stream at a time)


def _native_scratch(device: torch.device, nbytes: int) -> torch.Tensor:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    buf = _NATIVE_SCRATCH.get(idx)
    if buf is None or buf.numel() < nbytes:
        _NATIVE_SCRATCH.pop(idx, None)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _NATIVE_SCRATCH[idx] = buf
    return buf


class WanBlockLayout:
    """Order of the parameters inside a block's flat buffer.  Projections that read the same input are adjacent, so that q|k|v (self-attention) and
    k|v (cross-attention) are each ONE [3D, D] / [2D, D] matrix for the GEMM without any copy."""

    def __init__(self, dim: int, ffn_dim: int):
        if dim % 128 != 0 or ffn_dim % 64 != 0:
            raise ValueError("Wan block: dim must be a multiple of 128 (heads of 128), ffn_dim of 64")
        D, F = dim, ffn_dim
        self.dim, self.ffn_dim = D, F
        entries: List[Tuple[str, Tuple[int, ...]]] = [
            ("attn1.to_q.weight", (D, D)), ("attn1.to_k.weight", (D, D)), ("attn1.to_v.weight", (D, D)),
            ("attn1.to_q.bias", (D,)), ("attn1.to_k.bias", (D,)), ("attn1.to_v.bias", (D,)),
            ("attn1.to_out.0.weight", (D, D)), ("attn1.to_out.0.bias", (D,)),
            ("attn1.norm_q.weight", (D,)), ("attn1.norm_k.weight", (D,)),
            ("attn2.to_q.weight", (D, D)), ("attn2.to_q.bias", (D,)),
            ("attn2.to_k.weight", (D, D)), ("attn2.to_v.weight", (D, D)), ("attn2.to_k.bias", (D,)), ("attn2.to_v.bias", (D,)),
            ("attn2.to_out.0.weight", (D, D)), ("attn2.to_out.0.bias", (D,)),
            ("attn2.norm_q.weight", (D,)), ("attn2.norm_k.weight", (D,)),
            ("norm2.weight", (D,)), ("norm2.bias", (D,)),
            ("ffn.net.0.proj.weight", (F, D)), ("ffn.net.0.proj.bias", (F,)),
            ("ffn.net.2.weight", (D, F)), ("ffn.net.2.bias", (D,)),
            ("scale_shift_table", (1, 6, D)),
        ]
        self.entries = entries
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape in entries:
            n = 1
            for s in shape:
                n *= s
            self.offsets[name] = (off, shape)
            off += n  # every size is a multiple of 64 elements: all slices stay 16-byte aligned
        self.total = off
        # fused views: (first member, rows, cols)
        self.fused = {
            "w_qkv1": ("attn1.to_q.weight", 3 * D, D), "b_qkv1": ("attn1.to_q.bias", 3 * D, None),
            "w_kv2": ("attn2.to_k.weight", 2 * D, D), "b_kv2": ("attn2.to_k.bias", 2 * D, None),
        }

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        if name in self.fused:
            first, rows, cols = self.fused[name]
            off = self.offsets[first][0]
            return flat[off:off + rows * (cols or 1)].view((rows, cols) if cols else (rows,))
        off, shape = self.offsets[name]
        n = 1
        for s in shape:
            n *= s
        return flat[off:off + n].view(shape)

    def named_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {name: self.view(flat, name) for name, _ in self.entries}


class _WanBlockFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, blk: "MI355XWanBlock", x, enc, temb, rope_cos, rope_sin):
        B, S, D = x.shape
        # scale_shift_table (bf16 parameter [1, 6, D]) + temb.float(): fp32 [B, 6, D] = (shift, scale, gate) of the attention, then of the feed-forward
        mod = (blk.param("scale_shift_table").float() + temb.float()).contiguous()
        T = enc.shape[1]
        M, H, hd = B * S, blk.heads, blk.head_dim
        P = blk.param
        eps = blk.eps
        rope = (rope_cos, rope_sin)
        x2d, enc2d = x.view(M, D), enc.view(B * T, D)
        heads = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)  # [rows, D] view (any row stride) -> [B, H, n, hd]
        tok = lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0] * t.shape[2], D)  # attention output [B, H, n, hd] laid out [B, n, H, hd] -> [rows, D]

        # self-attention
        n1 = ops.wan_ln(x2d, S, shift=mod[:, 0], scale=mod[:, 1], eps=eps)
        qkv = ops.gemm_nt(n1, P("w_qkv1"), P("b_qkv1"))
        qn = ops.wan_rms_rope(qkv[:, :D], P("attn1.norm_q.weight"), S, rope=rope, head_dim=hd, eps=eps)
        kn = ops.wan_rms_rope(qkv[:, D:2 * D], P("attn1.norm_k.weight"), S, rope=rope, head_dim=hd, eps=eps)
        o1, lse1 = ops.attn_fwd(heads(qn, S), heads(kn, S), heads(qkv[:, 2 * D:], S))
        a1 = ops.gemm_nt(tok(o1), P("attn1.to_out.0.weight"), P("attn1.to_out.0.bias"))
        x1 = ops.wan_gate_res(x2d, a1, S, gate=mod[:, 2])
        # cross-attention to the text tokens (no rotary embedding, no gate)
        n2 = ops.wan_ln(x1, S, w=P("norm2.weight"), b=P("norm2.bias"), eps=eps)
        q2 = ops.gemm_nt(n2, P("attn2.to_q.weight"), P("attn2.to_q.bias"))
        kv2 = ops.gemm_nt(enc2d, P("w_kv2"), P("b_kv2"))
        q2n = ops.wan_rms_rope(q2, P("attn2.norm_q.weight"), S, eps=eps)
        k2n = ops.wan_rms_rope(kv2[:, :D], P("attn2.norm_k.weight"), T, eps=eps)
        o2, lse2 = ops.attn_fwd(heads(q2n, S), heads(k2n, T), heads(kv2[:, D:], T))
        a2 = ops.gemm_nt(tok(o2), P("attn2.to_out.0.weight"), P("attn2.to_out.0.bias"))
        x2 = ops.wan_gate_res(x1, a2, S)
        # feed-forward
        n3 = ops.wan_ln(x2, S, shift=mod[:, 3], scale=mod[:, 4], eps=eps)
        act, pre = ops.gemm_nt(n3, P("ffn.net.0.proj.weight"), P("ffn.net.0.proj.bias"), epilogue=1, want_out2=True)  # GELU-tanh, pre-activation kept
        f = ops.gemm_nt(act, P("ffn.net.2.weight"), P("ffn.net.2.bias"))
        out = ops.wan_gate_res(x2, f, S, gate=mod[:, 5])

        ctx.blk, ctx.dims, ctx.rope = blk, (B, S, T, D), rope
        ctx.temb_dtype = temb.dtype
        ctx.save_for_backward(x, enc, mod)
        ctx.acts = (n1, qkv, qn, kn, o1, lse1, a1, x1, n2, q2, kv2, q2n, k2n, o2, lse2, x2, n3, act, pre, f)
        return out.view(B, S, D)

    @staticmethod
    def backward(ctx, dout):
        blk = ctx.blk
        if blk._pre_backward is not None:
            blk._pre_backward(blk)  # sharded training: gather this block's parameters (prefetch the previous block's), take a gradient buffer
        B, S, T, D = ctx.dims
        x, enc, mod = ctx.saved_tensors
        n1, qkv, qn, kn, o1, lse1, a1, x1, n2, q2, kv2, q2n, k2n, o2, lse2, x2, n3, act, pre, f = ctx.acts
        ctx.acts = None
        M, H, hd, eps, rope = B * S, blk.heads, blk.head_dim, blk.eps, ctx.rope
        P, G, Wt = blk.param, blk.grad, blk.transposed()
        x2d, enc2d = x.view(M, D), enc.view(B * T, D)
        dout = dout.contiguous().view(M, D)
        heads = lambda t, n: t.view(B, n, H, hd).permute(0, 2, 1, 3)
        tok = lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0] * t.shape[2], D)
        dmod = torch.zeros((6, B, D), dtype=torch.float32, device=x.device)  # (shift, scale, gate) x (attention, feed-forward), summed over the tokens

        def linear_grads(name_w, name_b, dy, inp):  # dW += dY^T X (fp32, token-reduction GEMM), db += column sums of dY
            ops.gemm_tn(dy, inp, out=G(name_w))
            ops.wan_colsum(dy, G(name_b))

        # feed-forward branch: out = x2 + f * gate_ff
        df = ops.wan_gate_res_bwd(dout, f, mod[:, 5], S, dgate=dmod[5])
        linear_grads("ffn.net.2.weight", "ffn.net.2.bias", df, act)
        dpre = ops.gemm_nt(df, Wt["ffn.net.2.weight"], None, epilogue=3, aux=pre)  # (d f W2) * gelu'(pre)
        linear_grads("ffn.net.0.proj.weight", "ffn.net.0.proj.bias", dpre, n3)
        dn3 = ops.gemm_nt(dpre, Wt["ffn.net.0.proj.weight"], None)
        dx2 = ops.wan_ln_bwd(x2, dn3, S, scale=mod[:, 4], eps=eps, dres=dout, red1=dmod[3], red2=dmod[4], red_per_batch=True)
        # cross-attention branch: x2 = x1 + a2
        linear_grads("attn2.to_out.0.weight", "attn2.to_out.0.bias", dx2, tok(o2))
        do2 = ops.gemm_nt(dx2, Wt["attn2.to_out.0.weight"], None)
        dkv2 = torch.empty_like(kv2)
        dq2n, dk2n, _ = ops.attn_bwd(heads(q2n, S), heads(k2n, T), heads(kv2[:, D:], T), o2, lse2, heads(do2, S), dv_out=heads(dkv2[:, D:], T))
        dq2 = ops.wan_rms_rope_bwd(q2, P("attn2.norm_q.weight"), tok(dq2n), S, eps=eps, dweight=G("attn2.norm_q.weight"))
        ops.wan_rms_rope_bwd(kv2[:, :D], P("attn2.norm_k.weight"), tok(dk2n), T, eps=eps, dweight=G("attn2.norm_k.weight"), out=dkv2[:, :D])
        linear_grads("attn2.to_q.weight", "attn2.to_q.bias", dq2, n2)
        linear_grads("w_kv2", "b_kv2", dkv2, enc2d)
        denc = ops.gemm_nt(dkv2, Wt["w_kv2"], None).view(B, T, D)
        dn2 = ops.gemm_nt(dq2, Wt["attn2.to_q.weight"], None)
        dx1 = ops.wan_ln_bwd(x1, dn2, S, w=P("norm2.weight"), eps=eps, dres=dx2, red1=G("norm2.bias"), red2=G("norm2.weight"))
        # self-attention branch: x1 = x + a1 * gate_msa
        da1 = ops.wan_gate_res_bwd(dx1, a1, mod[:, 2], S, dgate=dmod[2])
        linear_grads("attn1.to_out.0.weight", "attn1.to_out.0.bias", da1, tok(o1))
        do1 = ops.gemm_nt(da1, Wt["attn1.to_out.0.weight"], None)
        dqkv = torch.empty_like(qkv)
        dqn, dkn, _ = ops.attn_bwd(heads(qn, S), heads(kn, S), heads(qkv[:, 2 * D:], S), o1, lse1, heads(do1, S), dv_out=heads(dqkv[:, 2 * D:], S))
        ops.wan_rms_rope_bwd(qkv[:, :D], P("attn1.norm_q.weight"), tok(dqn), S, rope=rope, head_dim=hd, eps=eps, dweight=G("attn1.norm_q.weight"), out=dqkv[:, :D])
        ops.wan_rms_rope_bwd(qkv[:, D:2 * D], P("attn1.norm_k.weight"), tok(dkn), S, rope=rope, head_dim=hd, eps=eps, dweight=G("attn1.norm_k.weight"),
                             out=dqkv[:, D:2 * D])
        linear_grads("w_qkv1", "b_qkv1", dqkv, n1)
        dn1 = ops.gemm_nt(dqkv, Wt["w_qkv1"], None)  # the three projections' input gradients summed in the GEMM's fp32 accumulator
        dx = ops.wan_ln_bwd(x2d, dn1, S, scale=mod[:, 1], eps=eps, dres=dx1, red1=dmod[0], red2=dmod[1], red_per_batch=True)
        dmod = dmod.permute(1, 0, 2)  # [B, 6, D]
        G("scale_shift_table").add_(dmod.sum(0, keepdim=True))
        if blk._grad_hook is not None:
            blk._grad_hook(blk)  # sharded training: this block's gradients are final -- start their reduce-scatter while the earlier blocks run
        return None, dx.view(B, S, D), denc, dmod.to(ctx.temb_dtype), None, None


class _WanBlockNativeFunction(torch.autograd.Function):
    """``_WanBlockFunction`` with ONE C call per direction (``ftmi_wan_block_forward / _backward``, csrc/wan_dit.hip: the same kernels in the same order,
    activations in one planned buffer per block, transients in a buffer shared by all blocks).  The sharding hooks run around the calls exactly as in the
    Python composition: parameters are read through ``blk._params()`` (the local or the all-gathered buffer), gradients go to ``blk.grad_flat``."""

    @staticmethod
    def _cfg(blk: "MI355XWanBlock", B: int, S: int, T: int) -> WanBlockConfig:
        return WanBlockConfig(B=B, S=S, T=T, D=blk.dim, H=blk.heads, F=blk.ffn_dim, eps=float(blk.eps), gemm_variant=8)

    @staticmethod
    def forward(ctx, blk: "MI355XWanBlock", x, enc, temb, rope_cos, rope_sin):
        B, S, D = x.shape
        T = enc.shape[1]
        mod = (blk.param("scale_shift_table").float() + temb.float()).contiguous()
        cfg = _WanBlockNativeFunction._cfg(blk, B, S, T)
        lib = _lib.load()
        params = blk._params()
        if params.numel() != lib.ftmi_wan_block_param_elements(ctypes.byref(cfg)) or not params.is_contiguous():
            raise RuntimeError("Wan block: the flat parameter buffer does not have the layout the C orchestrator expects")
        saved = torch.empty(lib.ftmi_wan_block_saved_bytes(ctypes.byref(cfg)), dtype=torch.uint8, device=x.device)
        out = torch.empty_like(x)
        check(lib.ftmi_wan_block_forward(ctypes.byref(cfg), ptr(params), ptr(x), ptr(enc), ptr(mod), ptr(rope_cos), ptr(rope_sin), ptr(out), ptr(saved), saved.numel(),
                                         stream_ptr()), "ftmi_wan_block_forward")
        ctx.blk, ctx.dims, ctx.rope, ctx.temb_dtype = blk, (B, S, T, D), (rope_cos, rope_sin), temb.dtype
        ctx.save_for_backward(x, enc, mod, saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        blk = ctx.blk
        if blk._pre_backward is not None:
            blk._pre_backward(blk)  # sharded training: gather this block's parameters (prefetch the previous block's), take a gradient buffer
        B, S, T, D = ctx.dims
        x, enc, mod, saved = ctx.saved_tensors
        dout = dout.contiguous()
        dmod = torch.zeros((6, B, D), dtype=torch.float32, device=x.device)
        if blk.grad_flat is None:
            blk.zero_grad_flat()
        cfg = _WanBlockNativeFunction._cfg(blk, B, S, T)
        lib = _lib.load()
        scratch = _native_scratch(x.device, lib.ftmi_wan_block_scratch_bytes(ctypes.byref(cfg)))
        dx, denc = torch.empty_like(x), torch.empty_like(enc)
        check(lib.ftmi_wan_block_backward(ctypes.byref(cfg), ptr(blk._params()), ptr(blk.grad_flat), ptr(x), ptr(enc), ptr(mod), ptr(ctx.rope[0]), ptr(ctx.rope[1]),
                                          ptr(dout), ptr(dx), ptr(denc), ptr(dmod), ptr(saved), saved.numel(), ptr(scratch), scratch.numel(), stream_ptr()),
              "ftmi_wan_block_backward")
        dmod = dmod.permute(1, 0, 2)  # [B, 6, D]
        blk.grad("scale_shift_table").add_(dmod.sum(0, keepdim=True))
        if blk._grad_hook is not None:
            blk._grad_hook(blk)  # sharded training: this block's gradients are final -- start their reduce-scatter while the earlier blocks run
        return None, dx, denc, dmod.to(ctx.temb_dtype), None, None


class MI355XWanBlock(nn.Module):
    """Holds the block's flat bf16 parameters and flat fp32 gradients; ``forward(hidden_states, encoder_hidden_states, temb, rotary)`` like the
    reference block, ``temb`` = the [B, 6, D] time projection, ``rotary`` = (cos, sin) fp32 [S, head_dim / 2]."""

    # one C call per direction (csrc/wan_dit.hip); False (or FTMI_NATIVE_BLOCKS=0 in the environment): the per-kernel composition from Python -- the tests compare the two
    native = os.environ.get("FTMI_NATIVE_BLOCKS", "1") != "0"

    def __init__(self, dim: int = 1536, heads: int = 12, ffn_dim: int = 8960, eps: float = 1e-6, device: Optional[torch.device] = None):
        super().__init__()
        if dim % heads != 0 or dim // heads != 128:
            raise ValueError("the Wan path uses the head_dim-128 attention kernels")
        self.dim, self.heads, self.head_dim, self.ffn_dim, self.eps = dim, heads, dim // heads, ffn_dim, eps
        self.layout = WanBlockLayout(dim, ffn_dim)
        dev = device or torch.device("cuda", 0)
        # gradients do not go through ``.grad`` (they are fp32, the parameters bf16): the block's backward writes ``grad_flat``
        self.flat = nn.Parameter(torch.zeros(self.layout.total, dtype=bf16, device=dev), requires_grad=False)
        self.grad_flat: Optional[torch.Tensor] = None  # fp32, allocated by ``zero_grad_flat`` (the sharded trainer hands in its own buffer)
        self._transposed: Optional[Dict[str, torch.Tensor]] = None
        self._transposed_version = None
        self._grad_hook = None     # callable(block) at the end of the block's backward (its gradients are final)
        self._pre_forward = None   # callable(block) before the block's forward / backward: sharded training gathers the parameters there
        self._pre_backward = None
        self._param_src: Optional[torch.Tensor] = None  # sharded training: the all-gathered parameters to compute with instead of ``flat``

    # -- parameter / gradient views -----------------------------------------------------------------------------------------------------------
    def _params(self) -> torch.Tensor:
        return self.flat.data if self._param_src is None else self._param_src

    def param(self, name: str) -> torch.Tensor:
        return self.layout.view(self._params(), name)

    def grad(self, name: str) -> torch.Tensor:
        if self.grad_flat is None:
            self.zero_grad_flat()
        return self.layout.view(self.grad_flat, name)

    def zero_grad_flat(self, buffer: Optional[torch.Tensor] = None) -> None:
        if buffer is not None:
            if buffer.shape != (self.layout.total,) or buffer.dtype != torch.float32:
                raise ValueError("gradient buffer must be fp32 [layout.total]")
            self.grad_flat = buffer
        if self.grad_flat is None:
            self.grad_flat = torch.zeros(self.layout.total, dtype=torch.float32, device=self.flat.device)
        else:
            self.grad_flat.zero_()

    def named_grads(self) -> Dict[str, torch.Tensor]:
        return self.layout.named_views(self.grad_flat)

    def transposed(self) -> Dict[str, torch.Tensor]:
        """K-contiguous copies of the weights for the input-gradient GEMMs (dX = dY W as an NT GEMM against W^T), rebuilt when the parameters changed."""
        src = self._params()
        ver = (src.data_ptr(), src._version, self._epoch)
        if self._transposed is None or ver != self._transposed_version:
            names = ("w_qkv1", "attn1.to_out.0.weight", "attn2.to_q.weight", "w_kv2", "attn2.to_out.0.weight", "ffn.net.0.proj.weight", "ffn.net.2.weight")
            self._transposed = {n: ops.transpose_bf16(self.param(n)) for n in names}
            self._transposed_version = ver
        return self._transposed

    _epoch = 0

    def mark_updated(self) -> None:
        """The parameters were changed in place by the library (optimiser kernel, all-gather into the same buffer): drop the cached transposes."""
        self._epoch += 1

    # -- loading --------------------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """``sd``: a diffusers ``WanTransformerBlock`` state dict."""
        missing = [n for n, _ in self.layout.entries if n not in sd]
        if missing:
            raise KeyError(f"Wan block state dict lacks {missing[:4]}")
        for name, view in self.layout.named_views(self.flat.data).items():
            view.copy_(sd[name].to(bf16).reshape(view.shape))
        self.mark_updated()

    def state_dict_views(self) -> Dict[str, torch.Tensor]:
        if self.flat.numel() < self.layout.total:
            raise RuntimeError("this block's parameters are sharded over the ranks: use the step object's gathered_state_dict()")
        return self.layout.named_views(self.flat.data)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor, rotary) -> torch.Tensor:
        if self._pre_forward is not None:
            self._pre_forward(self)
        fn = _WanBlockNativeFunction if self.native else _WanBlockFunction
        return fn.apply(self, hidden_states.contiguous(), encoder_hidden_states.contiguous(), temb.contiguous(), rotary[0], rotary[1])
