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

import ctypes
import os

from .. import _lib, ops
from .._lib import HY_DUAL_WEIGHT_FIELDS, HyDualConfig, HyDualWeights, HySingleConfig, HySingleWeights, check, ptr, stream_ptr

bf16 = torch.bfloat16
_NATIVE_SCRATCH: Dict[int, torch.Tensor] = {}  # device index -> byte buffer shared by every natively run block on that device (one stream at a time)


def _native_scratch(device: torch.device, nbytes: int) -> torch.Tensor:
    idx = device.index if device.index is not None else torch.cuda.current_device()
    buf = _NATIVE_SCRATCH.get(idx)
    if buf is None or buf.numel() < nbytes:
        buf = None
        _NATIVE_SCRATCH.pop(idx, None)
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _NATIVE_SCRATCH[idx] = buf
    return buf


class _SingleBlockFunction(torch.autograd.Function):
    @staticmethod
    def _run(blk: "MI355XHunyuanSingleBlock", x, temb_silu, key_bias, rope_cos, rope_sin, text_len, lora_a, lora_b, need_out: bool = True):
        """The block's forward kernels: (output, the tensors its backward reads).  Called by ``forward`` and, under gradient checkpointing, a second time
        by ``backward`` -- the kernels are deterministic, so the recomputed activations are bit-identical to the ones that were dropped."""
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
        out = None
        if need_out:  # (the recomputation inside the backward stops here: nothing downstream of the attention is read again)
            cat[:, :D].copy_(o.permute(0, 2, 1, 3).reshape(M, D))
            y = ops.gemm_nt(cat, blk.proj_out_w, blk.proj_out_b)
            out = ops.cog_gate_residual(x, y.view(B, N, D), gate, 0)
        # proj_out and the MLP carry no adapter: their input (``cat``, 5 x the token width) is not needed again and is NOT kept -- 40 GB over the 40 blocks
        # at 32 896 tokens
        return out, (n, q, k, qn, kn, v, o, lse, pre, onep, gate, xa_q, xa_k, xa_v)

    @staticmethod
    def forward(ctx, blk: "MI355XHunyuanSingleBlock", x, temb_silu, key_bias, rope_cos, rope_sin, text_len, lora_a, lora_b):
        out, acts = _SingleBlockFunction._run(blk, x, temb_silu, key_bias, rope_cos, rope_sin, text_len, lora_a, lora_b)
        ctx.blk, ctx.T, ctx.rope, ctx.key_bias, ctx.has_lora = blk, int(text_len), (rope_cos, rope_sin), key_bias, lora_a is not None
        ctx.recompute = bool(blk.gradient_checkpointing)
        la, lb = (lora_a, lora_b) if lora_a is not None else (x.new_empty(0), x.new_empty(0))
        if ctx.recompute:  # --gradient_checkpointing: keep the block's INPUT only (1 token-width tensor instead of 12)
            ctx.save_for_backward(x, temb_silu, la, lb)
        else:
            ctx.save_for_backward(x, *acts, la, lb)
        return out

    @staticmethod
    def backward(ctx, dout):
        blk, T, rope = ctx.blk, ctx.T, ctx.rope
        if ctx.recompute:
            x, temb_silu, lora_a, lora_b = ctx.saved_tensors
            if not ctx.has_lora:
                lora_a = lora_b = None
            if blk._bwd_seen == 0:
                blk._materialize_fwd()  # fp8 storage: the arena holds another block's weights by now
            _, acts = _SingleBlockFunction._run(blk, x, temb_silu, ctx.key_bias, rope[0], rope[1], T, lora_a, lora_b, need_out=False)
            n, q, k, qn, kn, v, o, lse, pre, onep, gate, xa_q, xa_k, xa_v = acts
        else:
            x, n, q, k, qn, kn, v, o, lse, pre, onep, gate, xa_q, xa_k, xa_v, lora_a, lora_b = ctx.saved_tensors
            if not ctx.has_lora:
                lora_a = lora_b = None
        if blk._bwd_seen == 0:
            blk._materialize_bwd()  # fp8 storage: the transposed bf16 weights of THIS block into the shared arena (no-op otherwise)
        B, N, D = x.shape
        M, H, hd, s = B * N, blk.heads, 128, blk.lora_scale
        dout = dout.contiguous()
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        # the step object may own the gradient storage (views of ONE flat fp32 buffer it zeroes before the backward): the kernels then add into it in
        # place, .grad is set to the views and autograd gets None -- nothing is concatenated afterwards, and the block's slice can be exchanged while the
        # earlier blocks still compute
        own = lora_a is not None and blk._grad_a_view is not None
        ga = blk._grad_a_view if own else (torch.zeros_like(lora_a) if lora_a is not None else None)
        gb = blk._grad_b_view if own else (torch.zeros_like(lora_b) if lora_b is not None else None)
        GA = lambda i: None if ga is None else ga[i]
        GB = lambda i: None if gb is None else gb[i]
        dy = ops.cog_gate_residual(None, dout, gate, 0).view(M, D)  # d proj_out output = gate * d out
        wt = blk.proj_out_w_t  # [D + mlp, D]
        do = ops.gemm_nt(dy, wt[:D], None)                             # gradient of the attention features
        dpre = ops.gemm_nt(dy, wt[D:], None, epilogue=3, aux=pre)      # (gradient of the MLP features) * gelu'(pre)
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
        if own:
            blk._backward_done(ga, gb)
            return None, dx, None, None, None, None, None, None, None
        return None, dx, None, None, None, None, None, ga, gb


class _SingleBlockNativeFunction(torch.autograd.Function):
    """The same block as ``_SingleBlockFunction`` with ONE C call per direction (``ftmi_hy_single_forward / _backward``, csrc/hy_dit.hip: the identical
    kernel sequence, issued from C out of a planned ``saved`` buffer per block and a ``scratch`` buffer shared by all blocks).  Gradient checkpointing
    keeps the block's input only and rebuilds ``saved`` inside the backward (the forward call with ``out = NULL``)."""

    @staticmethod
    def _args(blk: "MI355XHunyuanSingleBlock", B: int, N: int, T: int, lora_a, lora_b, backward: bool):
        D = blk.dim
        cfg = HySingleConfig(B=B, T=T, S=N - T, D=D, H=blk.heads, mlp=blk.mlp_dim, r=0 if lora_a is None else int(lora_a.shape[1]),
                             lora_scale=float(blk.lora_scale), eps=1e-6, gemm_variant=8)
        w = HySingleWeights()
        names = ["norm_lin_w", "norm_lin_b", "proj_mlp_w", "proj_mlp_b", "wq", "bq", "wk", "bk", "wv", "bv", "norm_q_w", "norm_k_w", "proj_out_w", "proj_out_b", "ones", "zeros"]
        if backward:
            names += ["wq_t", "wk_t", "wv_t", "proj_mlp_w_t", "proj_out_w_t"]
        keep = []
        for n in names:
            t = getattr(blk, n)
            if t is None or not t.is_contiguous():
                raise RuntimeError(f"HunyuanVideo single-stream block: weight '{n}' is not materialised")
            keep.append(t)
            setattr(w, n, ptr(t))
        if lora_a is not None:
            la, lb = lora_a.contiguous(), lora_b.contiguous()
            keep += [la, lb]
            w.lora_a, w.lora_b = ptr(la), ptr(lb)
        return cfg, w, keep

    @staticmethod
    def _forward_call(blk, x, temb_silu, key_bias, rope_cos, rope_sin, T, lora_a, lora_b, out):
        B, N, _ = x.shape
        cfg, w, keep = _SingleBlockNativeFunction._args(blk, B, N, T, lora_a, lora_b, backward=False)
        lib = _lib.load()
        saved = torch.empty(lib.ftmi_hy_single_saved_bytes(ctypes.byref(cfg)), dtype=torch.uint8, device=x.device)
        scratch = _native_scratch(x.device, lib.ftmi_hy_single_scratch_bytes(ctypes.byref(cfg)))
        check(lib.ftmi_hy_single_forward(ctypes.byref(cfg), ctypes.byref(w), ptr(x), ptr(temb_silu), ptr(key_bias), ptr(rope_cos), ptr(rope_sin), ptr(out),
                                         ptr(saved), saved.numel(), ptr(scratch), scratch.numel(), stream_ptr()), "ftmi_hy_single_forward")
        return saved

    @staticmethod
    def forward(ctx, blk: "MI355XHunyuanSingleBlock", x, temb_silu, key_bias, rope_cos, rope_sin, text_len, lora_a, lora_b):
        out = torch.empty_like(x)
        saved = _SingleBlockNativeFunction._forward_call(blk, x, temb_silu, key_bias, rope_cos, rope_sin, int(text_len), lora_a, lora_b, out)
        ctx.blk, ctx.T, ctx.rope, ctx.key_bias, ctx.has_lora = blk, int(text_len), (rope_cos, rope_sin), key_bias, lora_a is not None
        ctx.recompute = bool(blk.gradient_checkpointing)
        la, lb = (lora_a, lora_b) if lora_a is not None else (x.new_empty(0), x.new_empty(0))
        if ctx.recompute:
            ctx.save_for_backward(x, temb_silu, la, lb)
        else:
            ctx.save_for_backward(x, temb_silu, la, lb, saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        blk, T, rope = ctx.blk, ctx.T, ctx.rope
        if ctx.recompute:
            x, temb_silu, lora_a, lora_b = ctx.saved_tensors
        else:
            x, temb_silu, lora_a, lora_b, saved = ctx.saved_tensors
        if not ctx.has_lora:
            lora_a = lora_b = None
        if ctx.recompute:
            if blk._bwd_seen == 0:
                blk._materialize_fwd()  # fp8 storage: the arena holds another block's weights by now
            saved = _SingleBlockNativeFunction._forward_call(blk, x, temb_silu, ctx.key_bias, rope[0], rope[1], T, lora_a, lora_b, None)
        if blk._bwd_seen == 0:
            blk._materialize_bwd()
        B, N, _ = x.shape
        dout = dout.contiguous()
        own = lora_a is not None and blk._grad_a_view is not None
        ga = blk._grad_a_view if own else (torch.zeros_like(lora_a) if lora_a is not None else None)
        gb = blk._grad_b_view if own else (torch.zeros_like(lora_b) if lora_b is not None else None)
        cfg, w, keep = _SingleBlockNativeFunction._args(blk, B, N, T, lora_a, lora_b, backward=True)
        lib = _lib.load()
        scratch = _native_scratch(x.device, lib.ftmi_hy_single_scratch_bytes(ctypes.byref(cfg)))
        dx = torch.empty_like(x)
        check(lib.ftmi_hy_single_backward(ctypes.byref(cfg), ctypes.byref(w), ptr(x), ptr(dout), ptr(ctx.key_bias), ptr(rope[0]), ptr(rope[1]),
                                          ptr(blk.ones_rows(B, x.device)), ptr(dx), ptr(ga), ptr(gb), ptr(saved), saved.numel(), ptr(scratch), scratch.numel(),
                                          stream_ptr()), "ftmi_hy_single_backward")
        if own:
            blk._backward_done(ga, gb)
            return None, dx, None, None, None, None, None, None, None
        return None, dx, None, None, None, None, None, ga, gb


class _FlatGradMixin:
    """Gradient storage handed in by the step object (hunyuan_video/trainer.py): ``_grad_a_view`` / ``_grad_b_view`` are views of its flat buffer laid out
    like the parameters; ``_grad_hook(block)`` is called once per step, when the block's LAST backward call (one per forward call) has added its part."""
    _grad_a_view = None
    _grad_b_view = None
    _grad_hook = None
    gradient_checkpointing = False  # True: the block keeps only its input and runs its forward kernels again inside the backward
    _fwd_calls = 0
    _bwd_seen = 0

    def _backward_done(self, ga, gb) -> None:
        self._bwd_seen += 1
        if self._bwd_seen >= self._fwd_calls:
            self.lora_A.grad, self.lora_B.grad = ga, gb
            self._bwd_seen = 0
            if self._grad_hook is not None:
                self._grad_hook(self)


class _Fp8StorageMixin:
    """Real fp8 weight storage (the reference's layerwise up-casting, trainer/sft_trainer/trainer.py:111-118: storage float8_e4m3fn, compute bf16).
    ``store_weights_fp8`` moves every 2-D frozen weight of the block into e4m3fn bytes and drops its bf16 buffer and its transposed bf16 copy;
    ``_materialize_fwd`` / ``_materialize_bwd`` cast the block's weights up into a bf16 arena the model shares between ALL blocks (forward layout /
    transposed layout for the input-gradient GEMMs) right before the block's forward / backward kernels are queued -- same stream, so the arena is
    reused safely block after block.  1 byte per frozen parameter in HBM instead of 4 (bf16 + transposed bf16)."""
    _w8 = None          # name -> float8_e4m3fn tensor
    _arena_fwd = None   # bf16 scratch shared by all blocks (set by the model)
    _arena_bwd = None

    def _weight_names_2d(self):
        # (the reference's skip patterns keep every module whose name contains "norm" in bf16 -- args.py:395 --: the AdaLN Linear layers stay as they are)
        return [n for n, b in self.named_buffers() if b is not None and b.dim() == 2 and not n.endswith("_t") and not n.startswith("norm")]

    def _transposed_names(self):
        return list(getattr(self, "_TRANSPOSED", ("wq", "wk", "wv", "proj_mlp_w", "proj_out_w")))

    def fp8_elements(self):
        """(elements of all 2-D weights, elements of the transposed subset): what the two arenas must hold."""
        src = self._w8 if self._w8 is not None else {n: getattr(self, n) for n in self._weight_names_2d()}
        return sum(v.numel() for v in src.values()), sum(src[n].numel() for n in self._transposed_names())

    @torch.no_grad()
    def store_weights_fp8(self) -> int:
        if self._w8 is not None:
            return 0  # already stored (a second apply_layerwise_casting is a no-op)
        names = self._weight_names_2d()
        self._w8 = {n: getattr(self, n).to(torch.float8_e4m3fn).contiguous() for n in names}
        self._shapes = {n: tuple(getattr(self, n).shape) for n in names}
        # The bf16 copies a block computes with are views of ONE arena shared by all blocks, valid only between this block's up-cast and the next
        # block's.  They must not be module buffers: state_dict() / named_buffers() / .to() would then report whichever block was materialised
        # last as the weights of every block.  The names leave ``_buffers`` and become plain attributes; state_dict() exports the exact up-cast of
        # the stored bytes instead (hook below).
        for n in names + [t + "_t" for t in self._transposed_names()]:
            self._buffers.pop(n, None)
            object.__setattr__(self, n, None)
        self._register_state_dict_hook(_fp8_state_dict_hook)
        return len(names)

    def _materialize_fwd(self) -> None:
        if self._w8 is None:
            return
        off = 0
        for n, w8 in self._w8.items():
            k = w8.numel()
            object.__setattr__(self, n, ops.fp8_upcast(w8, out=self._arena_fwd[off:off + k]))
            off += k

    def _materialize_bwd(self) -> None:
        if self._w8 is None:
            return
        off = 0
        for n in self._transposed_names():
            w8 = self._w8[n]
            k = w8.numel()
            object.__setattr__(self, n + "_t", ops.fp8_upcast(w8, out=self._arena_bwd[off:off + k], transpose=True))
            off += k


def _fp8_state_dict_hook(module, state_dict, prefix, local_metadata):
    """state_dict() of a block whose weights live as e4m3fn bytes: the exact bf16 up-cast of the stored values under the buffers' own names."""
    for n, w8 in (module._w8 or {}).items():
        state_dict[prefix + n] = w8.to(bf16)
    return state_dict


class MI355XHunyuanSingleBlock(_FlatGradMixin, _Fp8StorageMixin, nn.Module):
    """Frozen bf16 weights (+ the transposes the input-gradient GEMMs use, made once) and the fp32 LoRA adapters of to_q / to_k / to_v."""

    # one C call per direction (csrc/hy_dit.hip); False (or FTMI_NATIVE_BLOCKS=0 in the environment): the per-kernel composition from Python -- the tests compare the two
    native = os.environ.get("FTMI_NATIVE_BLOCKS", "1") != "0"

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
        if self._w8 is not None:
            raise RuntimeError("the block's weights are stored in fp8: load the bf16 weights first, cast afterwards")
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items()}
        missing = [k for k in self._KEYS if k not in sd]
        if missing:
            raise KeyError(f"HunyuanVideo single-stream block state dict lacks {missing[:4]}")
        for k, name in self._KEYS.items():
            getattr(self, name).copy_(sd[k].to(bf16))
        for name in ("wq", "wk", "wv", "proj_mlp_w", "proj_out_w"):
            setattr(self, name + "_t", ops.transpose_bf16(getattr(self, name)))

    def add_adapter(self, r: int = 64, lora_alpha: float = 64.0) -> None:
        # ranks that are not multiples of 64 are stored zero-padded (the padding provably stays zero: see the CogVideoX block's add_adapter)
        if r <= 0:
            raise ValueError(f"LoRA rank must be positive, got {r}")
        rp = -(-int(r) // 64) * 64
        dev, D = self.ones.device, self.dim  # (self.wq is gone once the weights are stored in fp8)
        a = torch.zeros(3, rp, D, dtype=torch.float32, device=dev)
        a[:, :r].uniform_(-(1.0 / D) ** 0.5, (1.0 / D) ** 0.5)  # kaiming_uniform_(a = sqrt(5)) on [r, D]
        self.lora_A, self.lora_B = nn.Parameter(a), nn.Parameter(torch.zeros(3, D, rp, dtype=torch.float32, device=dev))
        self.lora_rank_user = int(r)
        self.lora_scale = float(lora_alpha) / r

    def forward(self, tokens: torch.Tensor, temb: torch.Tensor, text_len: int, image_rotary_emb, text_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``tokens`` [B, T + S, D] bf16 (text first), ``temb`` [B, D] the conditioning vector, ``image_rotary_emb`` = (cos, sin) fp32 [S, 128],
        ``text_mask`` [B, T] (1 = real token; None: all real) -> the block's output tokens in the same layout."""
        if self.wq_t is None and self._w8 is None:
            raise RuntimeError("load_diffusers_state_dict first (it also builds the transposed weights the input-gradient GEMMs use)")
        self._materialize_fwd()
        B, N, _ = tokens.shape
        key_bias = None
        if text_mask is not None:  # padded text tokens are never attended to (the reference masks every key beyond a sample's real text length)
            key_bias = torch.zeros((B, N), dtype=torch.float32, device=tokens.device)
            key_bias[:, :text_len].masked_fill_(~text_mask.to(tokens.device).bool(), float("-inf"))
        temb_silu = torch.nn.functional.silu(temb.to(bf16)).contiguous()
        cos, sin = image_rotary_emb
        self._fwd_calls, self._bwd_seen = 1, 0
        fn = _SingleBlockNativeFunction if self.native else _SingleBlockFunction
        return fn.apply(self, tokens.contiguous(), temb_silu, key_bias, cos.contiguous(), sin.contiguous(), int(text_len), self.lora_A, self.lora_B)


class _DualBlockFunction(torch.autograd.Function):
    """One sample (the modulation, the attention and its mask are per sample anyway; the module loops over the batch).  The two streams stay in their own
    buffers [T, D] / [S, D] as in the reference; only q, k, v are laid out as one joint [T + S] sequence (text first) for the attention."""

    @staticmethod
    def _run(blk: "MI355XHunyuanDualBlock", x_v, x_t, temb_silu, key_bias, rope_cos, rope_sin, lora_a, lora_b, need_out: bool = True):
        """Forward kernels of one sample: ((video out, text out), the tensors the backward reads); run again by ``backward`` under gradient checkpointing."""
        S, D = x_v.shape
        T = x_t.shape[0]
        N, H, hd, s = T + S, blk.heads, 128, blk.lora_scale
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        rope = (rope_cos, rope_sin)

        def modulation(w, b):  # AdaLayerNormZero: (shift, scale, gate) of the attention, then of the feed-forward, each [1, D]
            m = ops.gemm_nt(temb_silu, w, b).view(1, 6, D)
            return [m[:, i].contiguous() for i in (0, 2, 3, 5)] + [(1 + m[:, 1]).contiguous(), (1 + m[:, 4]).contiguous()]

        sh_v, g_v, shm_v, gm_v, op_v, opm_v = modulation(blk.norm1_lin_w, blk.norm1_lin_b)
        sh_t, g_t, shm_t, gm_t, op_t, opm_t = modulation(blk.norm1c_lin_w, blk.norm1c_lin_b)
        n_v = ops.cog_ln_mod(x_v[None], blk.ones, blk.zeros, sh_v, op_v, 0, 1e-6)[0]
        n_t = ops.cog_ln_mod(x_t[None], blk.ones, blk.zeros, sh_t, op_t, 0, 1e-6)[0]
        qj, kj, vj = (torch.empty((N, D), dtype=bf16, device=x_v.device) for _ in range(3))
        q_v, xa_q = ops.linear_lora_fwd(n_v, blk.wq, blk.bq, A(0), Bm(0), s)
        k_v, xa_k = ops.linear_lora_fwd(n_v, blk.wk, blk.bk, A(1), Bm(1), s)
        v_v, xa_v = ops.linear_lora_fwd(n_v, blk.wv, blk.bv, A(2), Bm(2), s)
        ops.head_rms_rope(q_v, blk.norm_q_w, hd, 1e-6, rope=rope, rows_per_batch=S, rope_from=0, out=qj[T:])
        ops.head_rms_rope(k_v, blk.norm_k_w, hd, 1e-6, rope=rope, rows_per_batch=S, rope_from=0, out=kj[T:])
        vj[T:].copy_(v_v)
        q_t = ops.gemm_nt(n_t, blk.add_q_w, blk.add_q_b)
        k_t = ops.gemm_nt(n_t, blk.add_k_w, blk.add_k_b)
        ops.gemm_nt(n_t, blk.add_v_w, blk.add_v_b, out=vj[:T])
        ops.head_rms_rope(q_t, blk.norm_added_q_w, hd, 1e-6, out=qj[:T])
        ops.head_rms_rope(k_t, blk.norm_added_k_w, hd, 1e-6, out=kj[:T])
        heads = lambda t: t.view(1, N, H, hd).permute(0, 2, 1, 3)
        o, lse = ops.attn_fwd(heads(qj), heads(kj), heads(vj), key_bias)
        o2d = o.permute(0, 2, 1, 3).reshape(N, D)
        a_v, xa_o = ops.linear_lora_fwd(o2d[T:], blk.wo, blk.bo, A(3), Bm(3), s)
        a_t = ops.gemm_nt(o2d[:T], blk.add_out_w, blk.add_out_b)
        h_v = ops.cog_gate_residual(x_v[None], a_v[None], g_v, 0)[0]
        h_t = ops.cog_gate_residual(x_t[None], a_t[None], g_t, 0)[0]
        n2_v = ops.cog_ln_mod(h_v[None], blk.ones, blk.zeros, shm_v, opm_v, 0, 1e-6)[0]
        n2_t = ops.cog_ln_mod(h_t[None], blk.ones, blk.zeros, shm_t, opm_t, 0, 1e-6)[0]
        act_v, pre_v = ops.gemm_nt(n2_v, blk.ff1_w, blk.ff1_b, epilogue=1, want_out2=True)
        act_t, pre_t = ops.gemm_nt(n2_t, blk.ffc1_w, blk.ffc1_b, epilogue=1, want_out2=True)
        out_v = out_t = None
        if need_out:  # (the recomputation inside the backward stops before the second feed-forward GEMMs)
            out_v = ops.cog_gate_residual(h_v[None], ops.gemm_nt(act_v, blk.ff2_w, blk.ff2_b)[None], gm_v, 0)[0]
            out_t = ops.cog_gate_residual(h_t[None], ops.gemm_nt(act_t, blk.ffc2_w, blk.ffc2_b)[None], gm_t, 0)[0]
        return (out_v, out_t), (n_v, n_t, q_v, k_v, q_t, k_t, qj, kj, vj, o, lse, h_v, h_t, pre_v, pre_t, g_v, g_t, gm_v, gm_t, op_v, op_t, opm_v, opm_t,
                                xa_q, xa_k, xa_v, xa_o)

    @staticmethod
    def forward(ctx, blk: "MI355XHunyuanDualBlock", x_v, x_t, temb_silu, key_bias, rope_cos, rope_sin, lora_a, lora_b):
        (out_v, out_t), acts = _DualBlockFunction._run(blk, x_v, x_t, temb_silu, key_bias, rope_cos, rope_sin, lora_a, lora_b)
        ctx.blk, ctx.rope, ctx.key_bias, ctx.has_lora = blk, (rope_cos, rope_sin), key_bias, lora_a is not None
        ctx.recompute = bool(blk.gradient_checkpointing)
        la, lb = (lora_a, lora_b) if lora_a is not None else (x_v.new_empty(0), x_v.new_empty(0))
        if ctx.recompute:  # --gradient_checkpointing: keep the two input streams only
            ctx.save_for_backward(x_v, x_t, temb_silu, la, lb)
        else:
            ctx.save_for_backward(x_v, x_t, *acts, la, lb)
        return out_v, out_t

    @staticmethod
    def backward(ctx, dout_v, dout_t):
        blk, rope = ctx.blk, ctx.rope
        if ctx.recompute:
            x_v, x_t, temb_silu, lora_a, lora_b = ctx.saved_tensors
            if not ctx.has_lora:
                lora_a = lora_b = None
            if blk._bwd_seen == 0:
                blk._materialize_fwd()  # fp8 storage: the arena holds another block's weights by now
            _, acts = _DualBlockFunction._run(blk, x_v, x_t, temb_silu, ctx.key_bias, rope[0], rope[1], lora_a, lora_b, need_out=False)
        else:
            x_v, x_t, *acts, lora_a, lora_b = ctx.saved_tensors
            if not ctx.has_lora:
                lora_a = lora_b = None
        (n_v, n_t, q_v, k_v, q_t, k_t, qj, kj, vj, o, lse, h_v, h_t, pre_v, pre_t, g_v, g_t, gm_v, gm_t, op_v, op_t, opm_v, opm_t,
         xa_q, xa_k, xa_v, xa_o) = acts
        if blk._bwd_seen == 0:
            blk._materialize_bwd()  # fp8 storage: the transposed bf16 weights of THIS block into the shared arena (no-op otherwise)
        S, D = x_v.shape
        T = x_t.shape[0]
        N, H, hd, s = T + S, blk.heads, 128, blk.lora_scale
        A = lambda i: None if lora_a is None else lora_a[i]
        Bm = lambda i: None if lora_b is None else lora_b[i]
        # the step object may own the gradient storage (views of ONE flat fp32 buffer it zeroes before the backward): the kernels then add into it in
        # place, .grad is set to the views and autograd gets None -- nothing is concatenated afterwards, and the block's slice can be exchanged while the
        # earlier blocks still compute
        own = lora_a is not None and blk._grad_a_view is not None
        ga = blk._grad_a_view if own else (torch.zeros_like(lora_a) if lora_a is not None else None)
        gb = blk._grad_b_view if own else (torch.zeros_like(lora_b) if lora_b is not None else None)
        GA = lambda i: None if ga is None else ga[i]
        GB = lambda i: None if gb is None else gb[i]
        dout_v, dout_t = dout_v.contiguous(), dout_t.contiguous()
        ones = blk.ones_rows(1, x_v.device)
        add = lambda a, b: ops.cog_gate_residual(a[None], b[None], ones, 0)[0]  # bf16 accumulation of two gradient tensors

        def ff_bwd(dout, gm, pre, w2_t, w1_t, h, opm):  # out = h + gate * FF(LN(h) * (1 + scale) + shift)
            df = ops.cog_gate_residual(None, dout[None], gm, 0)[0]
            dn2 = ops.gemm_nt(ops.gemm_nt(df, w2_t, None, epilogue=3, aux=pre), w1_t, None)
            return ops.cog_ln_mod_bwd(h[None], blk.ones, opm, dn2[None], 0, 1e-6, dres=dout[None])[0]

        dh_v = ff_bwd(dout_v, gm_v, pre_v, blk.ff2_w_t, blk.ff1_w_t, h_v, opm_v)
        dh_t = ff_bwd(dout_t, gm_t, pre_t, blk.ffc2_w_t, blk.ffc1_w_t, h_t, opm_t)
        # attention outputs: h = x + gate_msa * (o W_o^T + b)
        o2d = o.permute(0, 2, 1, 3).reshape(N, D)
        doj = torch.empty((N, D), dtype=bf16, device=x_v.device)
        da_v = ops.cog_gate_residual(None, dh_v[None], g_v, 0)[0]
        do_v, _, _ = ops.linear_lora_bwd(o2d[T:], da_v, xa_o, blk.wo_t, A(3), Bm(3), s, GA(3), GB(3))
        doj[T:].copy_(do_v)
        ops.gemm_nt(ops.cog_gate_residual(None, dh_t[None], g_t, 0)[0], blk.add_out_w_t, None, out=doj[:T])
        heads = lambda t: t.view(1, N, H, hd).permute(0, 2, 1, 3)
        flat = lambda t: t.permute(0, 2, 1, 3).reshape(N, D)
        dqj, dkj, dvj = (flat(t) for t in ops.attn_bwd(heads(qj), heads(kj), heads(vj), o, lse, heads(doj), ctx.key_bias))
        # video stream: RMSNorm + rotary backward, the three LoRA projections
        dq_v = ops.head_rms_rope_bwd(q_v, blk.norm_q_w, dqj[T:], hd, 1e-6, rope=rope, rows_per_batch=S, rope_from=0)
        dk_v = ops.head_rms_rope_bwd(k_v, blk.norm_k_w, dkj[T:], hd, 1e-6, rope=rope, rows_per_batch=S, rope_from=0)
        dn_q, _, _ = ops.linear_lora_bwd(n_v, dq_v, xa_q, blk.wq_t, A(0), Bm(0), s, GA(0), GB(0))
        dn_k, _, _ = ops.linear_lora_bwd(n_v, dk_v, xa_k, blk.wk_t, A(1), Bm(1), s, GA(1), GB(1))
        dn_vv, _, _ = ops.linear_lora_bwd(n_v, dvj[T:], xa_v, blk.wv_t, A(2), Bm(2), s, GA(2), GB(2))
        dn_v = add(add(dn_vv, dn_k), dn_q)
        dx_v = ops.cog_ln_mod_bwd(x_v[None], blk.ones, op_v, dn_v[None], 0, 1e-6, dres=dh_v[None])[0]
        # text stream
        dq_t = ops.head_rms_rope_bwd(q_t, blk.norm_added_q_w, dqj[:T], hd, 1e-6)
        dk_t = ops.head_rms_rope_bwd(k_t, blk.norm_added_k_w, dkj[:T], hd, 1e-6)
        dn_t = add(add(ops.gemm_nt(dvj[:T], blk.add_v_w_t, None), ops.gemm_nt(dk_t, blk.add_k_w_t, None)), ops.gemm_nt(dq_t, blk.add_q_w_t, None))
        dx_t = ops.cog_ln_mod_bwd(x_t[None], blk.ones, op_t, dn_t[None], 0, 1e-6, dres=dh_t[None])[0]
        if own:
            blk._backward_done(ga, gb)
            return None, dx_v, dx_t, None, None, None, None, None, None
        return None, dx_v, dx_t, None, None, None, None, ga, gb


class _DualBlockNativeFunction(torch.autograd.Function):
    """``_DualBlockFunction`` with ONE C call per sample and direction (``ftmi_hy_dual_forward / _backward``, csrc/hy_dit.hip)."""

    @staticmethod
    def _args(blk: "MI355XHunyuanDualBlock", S: int, T: int, lora_a, lora_b, backward: bool):
        cfg = HyDualConfig(T=T, S=S, D=blk.dim, H=blk.heads, mlp=int(blk.ff1_w.shape[0]), r=0 if lora_a is None else int(lora_a.shape[1]),
                           lora_scale=float(blk.lora_scale), eps=1e-6, gemm_variant=8)
        w = HyDualWeights()
        keep = []
        for n in HY_DUAL_WEIGHT_FIELDS:
            if n in ("lora_a", "lora_b") or (n.endswith("_t") and not backward):
                continue
            t = getattr(blk, n)
            if t is None or not t.is_contiguous():
                raise RuntimeError(f"HunyuanVideo dual-stream block: weight '{n}' is not materialised")
            keep.append(t)
            setattr(w, n, ptr(t))
        if lora_a is not None:
            la, lb = lora_a.contiguous(), lora_b.contiguous()
            keep += [la, lb]
            w.lora_a, w.lora_b = ptr(la), ptr(lb)
        return cfg, w, keep

    @staticmethod
    def _forward_call(blk, x_v, x_t, temb_silu, key_bias, rope_cos, rope_sin, lora_a, lora_b, out_v, out_t):
        cfg, w, keep = _DualBlockNativeFunction._args(blk, x_v.shape[0], x_t.shape[0], lora_a, lora_b, backward=False)
        lib = _lib.load()
        saved = torch.empty(lib.ftmi_hy_dual_saved_bytes(ctypes.byref(cfg)), dtype=torch.uint8, device=x_v.device)
        scratch = _native_scratch(x_v.device, lib.ftmi_hy_dual_scratch_bytes(ctypes.byref(cfg)))
        check(lib.ftmi_hy_dual_forward(ctypes.byref(cfg), ctypes.byref(w), ptr(x_v), ptr(x_t), ptr(temb_silu), ptr(key_bias), ptr(rope_cos), ptr(rope_sin), ptr(out_v),
                                       ptr(out_t), ptr(saved), saved.numel(), ptr(scratch), scratch.numel(), stream_ptr()), "ftmi_hy_dual_forward")
        return saved

    @staticmethod
    def forward(ctx, blk: "MI355XHunyuanDualBlock", x_v, x_t, temb_silu, key_bias, rope_cos, rope_sin, lora_a, lora_b):
        out_v, out_t = torch.empty_like(x_v), torch.empty_like(x_t)
        saved = _DualBlockNativeFunction._forward_call(blk, x_v, x_t, temb_silu, key_bias, rope_cos, rope_sin, lora_a, lora_b, out_v, out_t)
        ctx.blk, ctx.rope, ctx.key_bias, ctx.has_lora = blk, (rope_cos, rope_sin), key_bias, lora_a is not None
        ctx.recompute = bool(blk.gradient_checkpointing)
        la, lb = (lora_a, lora_b) if lora_a is not None else (x_v.new_empty(0), x_v.new_empty(0))
        if ctx.recompute:
            ctx.save_for_backward(x_v, x_t, temb_silu, la, lb)
        else:
            ctx.save_for_backward(x_v, x_t, temb_silu, la, lb, saved)
        return out_v, out_t

    @staticmethod
    def backward(ctx, dout_v, dout_t):
        blk, rope = ctx.blk, ctx.rope
        if ctx.recompute:
            x_v, x_t, temb_silu, lora_a, lora_b = ctx.saved_tensors
        else:
            x_v, x_t, temb_silu, lora_a, lora_b, saved = ctx.saved_tensors
        if not ctx.has_lora:
            lora_a = lora_b = None
        if ctx.recompute:
            if blk._bwd_seen == 0:
                blk._materialize_fwd()  # fp8 storage: the arena holds another block's weights by now
            saved = _DualBlockNativeFunction._forward_call(blk, x_v, x_t, temb_silu, ctx.key_bias, rope[0], rope[1], lora_a, lora_b, None, None)
        if blk._bwd_seen == 0:
            blk._materialize_bwd()
        dout_v, dout_t = dout_v.contiguous(), dout_t.contiguous()
        own = lora_a is not None and blk._grad_a_view is not None
        ga = blk._grad_a_view if own else (torch.zeros_like(lora_a) if lora_a is not None else None)
        gb = blk._grad_b_view if own else (torch.zeros_like(lora_b) if lora_b is not None else None)
        cfg, w, keep = _DualBlockNativeFunction._args(blk, x_v.shape[0], x_t.shape[0], lora_a, lora_b, backward=True)
        lib = _lib.load()
        scratch = _native_scratch(x_v.device, lib.ftmi_hy_dual_scratch_bytes(ctypes.byref(cfg)))
        dx_v, dx_t = torch.empty_like(x_v), torch.empty_like(x_t)
        check(lib.ftmi_hy_dual_backward(ctypes.byref(cfg), ctypes.byref(w), ptr(x_v), ptr(x_t), ptr(dout_v), ptr(dout_t), ptr(ctx.key_bias), ptr(rope[0]), ptr(rope[1]),
                                        ptr(blk.ones_rows(1, x_v.device)), ptr(dx_v), ptr(dx_t), ptr(ga), ptr(gb), ptr(saved), saved.numel(), ptr(scratch),
                                        scratch.numel(), stream_ptr()), "ftmi_hy_dual_backward")
        if own:
            blk._backward_done(ga, gb)
            return None, dx_v, dx_t, None, None, None, None, None, None
        return None, dx_v, dx_t, None, None, None, None, ga, gb


class MI355XHunyuanDualBlock(_FlatGradMixin, _Fp8StorageMixin, nn.Module):
    """HunyuanVideo dual-stream block (20 of the 60 blocks; [upstream] ``HunyuanVideoTransformerBlock``, oracle/hunyuan.py ``DualStreamBlock``): the video and
    the text tokens have their own modulation, projections, q / k norms and feed-forward and meet in ONE joint attention.  LoRA on the video stream's to_q /
    to_k / to_v / to_out.0 (what the default target regex matches; the text stream's ``add_*_proj`` / ``to_add_out`` stay frozen)."""

    _KEYS = {
        "norm1.linear.weight": "norm1_lin_w", "norm1.linear.bias": "norm1_lin_b", "norm1_context.linear.weight": "norm1c_lin_w", "norm1_context.linear.bias": "norm1c_lin_b",
        "attn.to_q.weight": "wq", "attn.to_q.bias": "bq", "attn.to_k.weight": "wk", "attn.to_k.bias": "bk", "attn.to_v.weight": "wv", "attn.to_v.bias": "bv",
        "attn.to_out.0.weight": "wo", "attn.to_out.0.bias": "bo", "attn.norm_q.weight": "norm_q_w", "attn.norm_k.weight": "norm_k_w",
        "attn.add_q_proj.weight": "add_q_w", "attn.add_q_proj.bias": "add_q_b", "attn.add_k_proj.weight": "add_k_w", "attn.add_k_proj.bias": "add_k_b",
        "attn.add_v_proj.weight": "add_v_w", "attn.add_v_proj.bias": "add_v_b", "attn.to_add_out.weight": "add_out_w", "attn.to_add_out.bias": "add_out_b",
        "attn.norm_added_q.weight": "norm_added_q_w", "attn.norm_added_k.weight": "norm_added_k_w",
        "ff.net.0.proj.weight": "ff1_w", "ff.net.0.proj.bias": "ff1_b", "ff.net.2.weight": "ff2_w", "ff.net.2.bias": "ff2_b",
        "ff_context.net.0.proj.weight": "ffc1_w", "ff_context.net.0.proj.bias": "ffc1_b", "ff_context.net.2.weight": "ffc2_w", "ff_context.net.2.bias": "ffc2_b",
    }
    _TRANSPOSED = ("wq", "wk", "wv", "wo", "add_q_w", "add_k_w", "add_v_w", "add_out_w", "ff1_w", "ff2_w", "ffc1_w", "ffc2_w")
    native = os.environ.get("FTMI_NATIVE_BLOCKS", "1") != "0"  # one C call per sample and direction (csrc/hy_dit.hip); False: the Python composition

    def __init__(self, dim: int = 3072, heads: int = 24, mlp_ratio: float = 4.0, device: Optional[torch.device] = None):
        super().__init__()
        if dim != heads * 128:
            raise ValueError("HunyuanVideo blocks have heads of 128 channels")
        self.dim, self.heads = dim, heads
        mlp = int(dim * mlp_ratio)
        dev = device or torch.device("cuda", 0)
        shapes = {"norm1_lin_w": (6 * dim, dim), "norm1_lin_b": (6 * dim,), "norm1c_lin_w": (6 * dim, dim), "norm1c_lin_b": (6 * dim,),
                  "ff1_w": (mlp, dim), "ff1_b": (mlp,), "ff2_w": (dim, mlp), "ff2_b": (dim,), "ffc1_w": (mlp, dim), "ffc1_b": (mlp,), "ffc2_w": (dim, mlp), "ffc2_b": (dim,)}
        for name in self._KEYS.values():
            is_weight = name in ("wq", "wk", "wv", "wo") or name.endswith("_w")
            shape = shapes.get(name) or ((128,) if name.startswith("norm_") else ((dim, dim) if is_weight else (dim,)))
            self.register_buffer(name, torch.zeros(shape, dtype=bf16, device=dev))
        for name in self._TRANSPOSED:
            self.register_buffer(name + "_t", None, persistent=False)
        self.register_buffer("ones", torch.ones(dim, dtype=bf16, device=dev), persistent=False)
        self.register_buffer("zeros", torch.zeros(dim, dtype=bf16, device=dev), persistent=False)
        self.lora_A: Optional[nn.Parameter] = None  # [4, r, D]: to_q, to_k, to_v, to_out.0
        self.lora_B: Optional[nn.Parameter] = None  # [4, D, r]
        self.lora_scale = 0.0
        self._ones_rows: Dict[int, torch.Tensor] = {}

    ones_rows = MI355XHunyuanSingleBlock.ones_rows

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        if self._w8 is not None:
            raise RuntimeError("the block's weights are stored in fp8: load the bf16 weights first, cast afterwards")
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items()}
        missing = [k for k in self._KEYS if k not in sd]
        if missing:
            raise KeyError(f"HunyuanVideo dual-stream block state dict lacks {missing[:4]}")
        for k, name in self._KEYS.items():
            getattr(self, name).copy_(sd[k].to(bf16))
        for name in self._TRANSPOSED:
            setattr(self, name + "_t", ops.transpose_bf16(getattr(self, name)))

    def add_adapter(self, r: int = 64, lora_alpha: float = 64.0) -> None:
        # ranks that are not multiples of 64 are stored zero-padded (the padding provably stays zero: see the CogVideoX block's add_adapter)
        if r <= 0:
            raise ValueError(f"LoRA rank must be positive, got {r}")
        rp = -(-int(r) // 64) * 64
        dev, D = self.ones.device, self.dim  # (self.wq is gone once the weights are stored in fp8)
        a = torch.zeros(4, rp, D, dtype=torch.float32, device=dev)
        a[:, :r].uniform_(-(1.0 / D) ** 0.5, (1.0 / D) ** 0.5)  # kaiming_uniform_(a = sqrt(5)) on [r, D]
        self.lora_A, self.lora_B = nn.Parameter(a), nn.Parameter(torch.zeros(4, D, rp, dtype=torch.float32, device=dev))
        self.lora_rank_user = int(r)
        self.lora_scale = float(lora_alpha) / r

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, temb: torch.Tensor, image_rotary_emb,
                text_mask: Optional[torch.Tensor] = None):
        """``hidden_states`` [B, S, D] (video), ``encoder_hidden_states`` [B, T, D] (text), ``temb`` [B, D], ``image_rotary_emb`` = (cos, sin) fp32 [S, 128],
        ``text_mask`` [B, T] (1 = real token) -> (video, text) like the reference block."""
        if self.wq_t is None and self._w8 is None:
            raise RuntimeError("load_diffusers_state_dict first (it also builds the transposed weights the input-gradient GEMMs use)")
        self._materialize_fwd()
        B, S, _ = hidden_states.shape
        T = encoder_hidden_states.shape[1]
        cos, sin = (t.contiguous() for t in image_rotary_emb)
        temb_silu = torch.nn.functional.silu(temb.to(bf16)).contiguous()
        outs_v, outs_t = [], []
        self._fwd_calls, self._bwd_seen = B, 0  # one autograd node per sample: the block's gradient is complete after B backward calls
        for b in range(B):
            key_bias = None
            if text_mask is not None:
                key_bias = torch.zeros((1, T + S), dtype=torch.float32, device=hidden_states.device)
                key_bias[0, :T].masked_fill_(~text_mask[b].to(hidden_states.device).bool(), float("-inf"))
            fn = _DualBlockNativeFunction if self.native else _DualBlockFunction
            ov, ot = fn.apply(self, hidden_states[b].contiguous(), encoder_hidden_states[b].contiguous(), temb_silu[b:b + 1], key_bias, cos, sin,
                                              self.lora_A, self.lora_B)
            outs_v.append(ov)
            outs_t.append(ot)
        return torch.stack(outs_v), torch.stack(outs_t)
