"""MI355X-native LTX-Video DiT behind the interface finetrainers' trainer expects of a transformer.

``MI355XLTXVideoTransformer3DModel`` is an ``nn.Module`` whose ``forward`` has the signature of the
reference's patched ``LTXVideoTransformer3DModel.forward`` (finetrainers/patches/models/ltx_video/
patch.py:38-50) and whose backward yields LoRA gradients -- but the whole 28-block forward and backward
run as hand-written gfx950 kernels launched by two C calls (``ftmi_ltx_forward`` / ``ftmi_ltx_backward``).

HBM layout (designed for 288 GB, not ported from the reference's per-module parameters):
  * frozen bf16 base weights are stacked per kind along a leading layer axis ([L,3D,D] fused q|k|v, ...)
    and every weight that needs a dgrad also keeps a transposed copy, so forward and dgrad are the same
    K-contiguous GEMM kernel (costs 1x extra weight memory, 3.6 GB);
  * the 224 LoRA adapters live in two flat fp32 parameters ``lora_A`` [L,8,r,D] and ``lora_B`` [L,8,D,r]
    (adapter order q,k,v,out of attn1 then attn2) -> one fused clip+AdamW launch and one contiguous
    all-reduce; the LoRA branch runs at fp32-equivalent precision like the reference's (fp32 adapters,
    trainer.py:132-136): bf16 (hi, lo) working copies of A / B are refreshed after each optimiser step;
  * all activations of a step live in one caller-owned workspace (about 0.37 GB per block at B=2).
``state_dict()`` / ``load_state_dict()`` speak the diffusers + peft key layout (views, no copies); ``lora_state_dict()`` is the
``get_peft_model_state_dict`` form.
"""

from __future__ import annotations

import ctypes
import math
import re
from dataclasses import dataclass
from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import LtxConfig, LtxWeights, check, ptr, stream_ptr

bf16 = torch.bfloat16

LORA_ORDER = ("attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0")
# finetrainers/trainer/sft_trainer/config.py:24-26
DEFAULT_TARGET_MODULES = "(transformer_blocks|single_transformer_blocks).*(to_q|to_k|to_v|to_out.0)"


@dataclass
class LTXTransformerConfig:
    """Hyper-parameters of LTXVideoTransformer3DModel (reference: tests/models/ltx_video/_test_tp.py:29-59)."""

    in_channels: int = 128
    out_channels: int = 128
    patch_size: int = 1
    patch_size_t: int = 1
    num_attention_heads: int = 32
    attention_head_dim: int = 64
    cross_attention_dim: int = 2048
    num_layers: int = 28
    caption_channels: int = 4096
    norm_eps: float = 1e-6
    qk_norm_eps: float = 1e-5
    ff_mult: int = 4

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def ltx_rope_tables(num_frames: int, height: int, width: int, rope_interpolation_scale, dim: int = 2048,
                    base_num_frames: int = 20, base_height: int = 2048, base_width: int = 2048, patch_size: int = 1,
                    patch_size_t: int = 1, theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """(cos, sin) per rotated pair, fp32 [S, dim/2], for one sample (identical across the batch).

    Same arithmetic as upstream ``LTXVideoRotaryPosEmbed`` (always fp32).  The upstream tables are
    [B,S,dim] with every frequency repeated for the two members of a pair and ``dim % 6`` leading
    (cos=1, sin=0) pad columns; since dim % 6 == 2 the pad is exactly pair 0, so one value per pair
    carries the same information.  Computed once per clip shape on the host and kept resident."""
    if dim % 6 != 2:
        raise ValueError("compact RoPE table assumes dim % 6 == 2 (LTX: 2048)")
    grid_f = torch.arange(num_frames, dtype=torch.float32)
    grid_h = torch.arange(height, dtype=torch.float32)
    grid_w = torch.arange(width, dtype=torch.float32)
    grid = torch.stack(torch.meshgrid(grid_f, grid_h, grid_w, indexing="ij"), dim=0).unsqueeze(0)
    if rope_interpolation_scale is not None:
        grid[:, 0:1] = grid[:, 0:1] * rope_interpolation_scale[0] * patch_size_t / base_num_frames
        grid[:, 1:2] = grid[:, 1:2] * rope_interpolation_scale[1] * patch_size / base_height
        grid[:, 2:3] = grid[:, 2:3] * rope_interpolation_scale[2] * patch_size / base_width
    grid = grid.flatten(2, 4).transpose(1, 2)  # [1, S, 3]
    freqs = theta ** torch.linspace(math.log(1.0, theta), math.log(theta, theta), dim // 6, dtype=torch.float32)
    freqs = freqs * math.pi / 2.0
    freqs = freqs * (grid.unsqueeze(-1) * 2 - 1)
    freqs = freqs.transpose(-1, -2).flatten(2)[0]  # [S, 3 * (dim // 6)]
    pad_c = torch.ones(freqs.shape[0], 1)
    pad_s = torch.zeros(freqs.shape[0], 1)
    cos = torch.cat([pad_c, freqs.cos()], dim=-1).contiguous()
    sin = torch.cat([pad_s, freqs.sin()], dim=-1).contiguous()
    return cos, sin


# finetrainers/args.py:395 (layerwise_upcasting_skip_modules_pattern default)
LAYERWISE_UPCASTING_SKIP_PATTERNS = ("patch_embed", "pos_embed", "x_embedder", "context_embedder", "time_embed", "^proj_in$", "^proj_out$", "norm")


class _LTXDiTFunction(torch.autograd.Function):
    """One autograd node for the whole DiT: Python overhead is O(1) per step."""

    @staticmethod
    def forward(ctx, module, x_t, text, key_bias, tvals, cos, sin, lora_a, lora_b):
        B, S, _ = x_t.shape
        T = text.shape[1]
        # gradient checkpointing only where a backward will follow (a forward under no_grad keeps every activation readable for the tests / tools)
        cfg = module._c_config(B, S, T, checkpoint=module.gradient_checkpointing and any(ctx.needs_input_grad) and module.lora_A is not None)
        weights = module._c_weights(cos, sin)
        lib = _lib.load()
        ws_bytes = lib.ftmi_ltx_workspace_bytes(ctypes.byref(cfg))
        # the activation workspace (10.4 GB at cfg 2) is recycled across steps instead of going back to the caching allocator:
        # a multi-GB block that is freed and re-requested every step occasionally costs a device-synchronising hipMalloc
        ws = module._acquire_workspace(ws_bytes, x_t.device)
        pred = torch.empty((B, S, module.config.out_channels), dtype=bf16, device=x_t.device)
        check(lib.ftmi_ltx_forward(ctypes.byref(cfg), ctypes.byref(weights), ptr(x_t), ptr(text), ptr(key_bias), ptr(tvals), ptr(pred),
                                   ptr(ws), ws_bytes, stream_ptr()), "ftmi_ltx_forward")
        ctx.module, ctx.cfg, ctx.weights, ctx.ws, ctx.ws_bytes = module, cfg, weights, ws, ws_bytes
        ctx.keep = (x_t, text, key_bias, tvals, cos, sin)
        module._last_workspace = (cfg, ws)
        if not any(ctx.needs_input_grad) or module.lora_A is None:
            module._release_workspace(ws)  # no backward will come for this call (its content stays readable until the next forward)
            ctx.ws = None
        return pred

    @staticmethod
    def backward(ctx, dpred):
        module = ctx.module
        if module.lora_A is None:
            return (None,) * 9
        dpred = dpred.contiguous()
        # ONE flat fp32 gradient buffer [A | B] -- a contiguous all-reduce per block range and a single clip+AdamW launch downstream.
        # The buffer IS lora_A.grad / lora_B.grad: the kernels write (or, under gradient accumulation, add) straight into it and
        # autograd gets None for the two parameters, so nothing is copied or re-added on the way.
        n_a, n_b = module._lora_A_full.numel(), module._lora_B_full.numel()  # padded rank (== the parameters' own size unless r % 64 != 0)
        ga_live, gb_live = module.lora_A.grad, module.lora_B.grad
        gflat = module._grad_flat_buf
        own = gflat is not None and gflat.numel() == n_a + n_b and gflat.device == dpred.device
        foreign = None
        if ga_live is None and gb_live is None:  # zero_grad(set_to_none=True) happened (trainer.py:503): start from zero
            if not own:
                gflat = module._grad_flat_buf = torch.empty(n_a + n_b, dtype=torch.float32, device=dpred.device)
            accumulate = 0
        elif (own and ga_live is not None and gb_live is not None and ga_live.data_ptr() == gflat.data_ptr()
              and gb_live.data_ptr() == gflat.data_ptr() + 4 * n_a):
            accumulate = 1  # gradient accumulation: add into the live buffer
        else:  # somebody installed their own .grad tensors: compute into a scratch buffer and let autograd accumulate
            if module._grad_bucket_hook is not None or module._grad_bucket_finish is not None:
                raise RuntimeError("the gradient exchange is installed on this model (apply_ddp / a data-parallel step) but lora_A.grad / lora_B.grad are not the "
                                   "backend's flat gradient buffer: foreign .grad tensors would never be all-reduced and the replicas would diverge silently -- "
                                   "use optimizer.zero_grad(set_to_none=True) (the reference trainer's default) instead of installing .grad tensors")
            foreign = gflat = torch.empty(n_a + n_b, dtype=torch.float32, device=dpred.device)
            accumulate = 0
        ga = gflat[:n_a].view_as(module._lora_A_full)
        gb = gflat[n_a:].view_as(module._lora_B_full)
        r_user = module.lora_rank  # the parameters (and their .grad) are the first r_user rows / columns of the padded storage
        _, text, key_bias, _, _, _ = ctx.keep
        lib = _lib.load()
        hook = module._grad_bucket_hook if foreign is None else None
        L = module.config.num_layers
        step = module.grad_bucket_blocks if (hook is not None and module.grad_bucket_blocks > 0) else L
        hi = L
        while hi > 0:  # block ranges in backward order; each range's gradients are final when its call returns (stream order)
            lo = max(0, hi - step)
            check(lib.ftmi_ltx_backward_range(ctypes.byref(ctx.cfg), ctypes.byref(ctx.weights), ptr(text), ptr(key_bias), ptr(dpred), ptr(ga), ptr(gb),
                                              ptr(ctx.ws), ctx.ws_bytes, hi, lo, accumulate, stream_ptr()), "ftmi_ltx_backward")
            if hook is not None:
                hook(lo, hi, ga[lo:hi], gb[lo:hi])
            hi = lo
        # MI355XParallelBackend.apply_ddp installs the exchange permanently, the way DDP's reducer sits on the module: the backward then also
        # ends it (the compute stream waits for the last bucket), so an unmodified loop may clip and step right after loss.backward()
        fin = module._grad_bucket_finish
        if hook is not None and fin is not None:
            fin()
        module._release_workspace(ctx.ws)
        ctx.ws = None
        if foreign is not None:
            return None, None, None, None, None, None, None, ga[:, :, :r_user, :], gb[:, :, :, :r_user]
        module._grad_flat = gflat
        if ga_live is None:
            module.lora_A.grad, module.lora_B.grad = ga[:, :, :r_user, :], gb[:, :, :, :r_user]
        return (None,) * 9


class MI355XLTXVideoTransformer3DModel(nn.Module):
    _FROZEN = {
        # name: shape builder (cfg dims) ; filled in __init__
    }

    def __init__(self, config: Optional[LTXTransformerConfig] = None, device: Optional[torch.device] = None, gemm_variant: int = 8):
        super().__init__()
        self.config = config or LTXTransformerConfig()
        c = self.config
        if c.inner_dim != 2048 or c.attention_head_dim != 64 or c.patch_size != 1 or c.patch_size_t != 1:
            raise ValueError("the MI355X kernels are built for LTX-Video's production geometry: width 2048 = 32 heads x 64, patch size 1")
        self.gemm_variant = gemm_variant
        # --gradient_checkpointing (trainer/sft_trainer/trainer.py:155-157; the reference's own LTX example sets it): off by default -- the activations of
        # BASELINE config 2 are 10.4 GB of 288 -- but implemented: one block slot instead of 28, every block's forward re-run inside its backward (bit-identical
        # gradients; see enable_gradient_checkpointing)
        self.gradient_checkpointing = False
        dev = device if device is not None else torch.device("cuda", torch.cuda.current_device() if torch.cuda.is_available() else 0)
        D, L, Dff, Dcap, Cin, Cout = c.inner_dim, c.num_layers, c.inner_dim * c.ff_mult, c.caption_channels, c.in_channels, c.out_channels
        shapes = {
            "proj_in_w": (D, Cin), "proj_in_b": (D,),
            "time_l1_w": (D, 256), "time_l1_b": (D,), "time_l2_w": (D, D), "time_l2_b": (D,),
            "time_lin_w": (6 * D, D), "time_lin_b": (6 * D,),
            "cap_l1_w": (D, Dcap), "cap_l1_b": (D,), "cap_l2_w": (D, D), "cap_l2_b": (D,),
            "tables": (L, 6, D), "table_out": (2, D),
            "proj_out_w": (Cout, D), "proj_out_b": (Cout,), "proj_out_w_t": (D, Cout),
            "w_qkv": (L, 3 * D, D), "b_qkv": (L, 3 * D), "w_qkv_t": (L, D, 3 * D),
            "norm_q": (L, D), "norm_k": (L, D),
            "w_o": (L, D, D), "b_o": (L, D), "w_o_t": (L, D, D),
            "w_q2": (L, D, D), "b_q2": (L, D), "w_q2_t": (L, D, D),
            "w_kv2": (L, 2 * D, D), "b_kv2": (L, 2 * D),
            "norm_q2": (L, D), "norm_k2": (L, D),
            "w_o2": (L, D, D), "b_o2": (L, D), "w_o2_t": (L, D, D),
            "w_ff1": (L, Dff, D), "b_ff1": (L, Dff), "w_ff1_t": (L, D, Dff),
            "w_ff2": (L, D, Dff), "b_ff2": (L, D), "w_ff2_t": (L, Dff, D),
        }
        self._frozen_names = list(shapes.keys())
        for n, shp in shapes.items():
            self.register_buffer(n, torch.zeros(shp, dtype=bf16, device=dev), persistent=False)
        self.lora_A: Optional[nn.Parameter] = None
        self.lora_B: Optional[nn.Parameter] = None
        self.lora_rank = 0
        self.lora_rank_padded = 0
        self._lora_A_full = self._lora_B_full = None
        self.lora_alpha = 0.0
        self._lora_versions = None
        self._rope_cache: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}
        self._last_workspace = None
        self._grad_flat = None
        self._grad_flat_buf = None
        # data-parallel gradient exchange (finetrainers_amd.trainer.MI355XSFTStep installs these): when set, the backward runs in ranges
        # of `grad_bucket_blocks` blocks and calls hook(l_lo, l_hi, grad_A[l_lo:l_hi], grad_B[l_lo:l_hi]) as soon as a range is final
        self._grad_bucket_hook = None
        self._grad_bucket_finish = None
        self.grad_bucket_blocks = 7
        self._ws_pool = []  # idle activation workspaces (uint8 tensors), see _acquire_workspace
        self.lora_flat = None

    # ------------------------------------------------------------------ weights
    @property
    def device(self) -> torch.device:
        return self.proj_in_w.device

    @property
    def dtype(self) -> torch.dtype:
        return bf16

    @torch.no_grad()
    def init_random_(self, seed: int = 0) -> "MI355XLTXVideoTransformer3DModel":
        """Random-init weights of the production architecture (nn.Linear default init; tables randn/sqrt(D);
        norm weights 1).  Used by bench.py / smoke (no checkpoints are reachable offline)."""
        g = torch.Generator(device=self.device).manual_seed(seed)
        D = self.config.inner_dim
        for n in self._frozen_names:
            t = getattr(self, n)
            if n.endswith("_t"):
                continue
            if n in ("tables", "table_out"):
                t.copy_((torch.randn(t.shape, generator=g, device=self.device) / D**0.5).to(bf16))
            elif n.startswith("norm_"):
                t.fill_(1.0)
            else:
                fan_in = {"b_qkv": D, "b_o": D, "b_q2": D, "b_kv2": D, "b_o2": D, "b_ff1": D, "b_ff2": 4 * D, "proj_in_b": self.config.in_channels,
                          "time_l1_b": 256, "time_l2_b": D, "time_lin_b": D, "cap_l1_b": self.config.caption_channels, "cap_l2_b": D,
                          "proj_out_b": D}.get(n, t.shape[-1])
                bound = 1.0 / math.sqrt(fan_in)
                # chunked to bound temporary fp32 memory
                flat = t.view(-1)
                step = 1 << 26
                for i in range(0, flat.numel(), step):
                    m = min(step, flat.numel() - i)
                    flat[i:i + m] = ((torch.rand(m, generator=g, device=self.device) * 2 - 1) * bound).to(bf16)
        self._make_transposes()
        return self

    @torch.no_grad()
    def _make_transposes(self) -> None:
        L = self.config.num_layers
        self.proj_out_w_t.copy_(ops.transpose_bf16(self.proj_out_w))
        for name in ("w_qkv", "w_o", "w_q2", "w_o2", "w_ff1", "w_ff2"):
            src, dst = getattr(self, name), getattr(self, name + "_t")
            for l in range(L):
                dst[l].copy_(ops.transpose_bf16(src[l]))

    @torch.no_grad()
    def apply_layerwise_casting(self, storage_dtype: torch.dtype = torch.float8_e4m3fn, compute_dtype: torch.dtype = bf16,
                                skip_modules_pattern=LAYERWISE_UPCASTING_SKIP_PATTERNS, non_blocking: bool = False) -> List[str]:
        """The reference's ``--layerwise_upcasting_modules transformer`` (trainer/sft_trainer/trainer.py:111-118 -> [upstream] diffusers
        ``apply_layerwise_casting``): every Linear whose module name matches none of ``skip_modules_pattern`` (regex search; default = args.py:395)
        keeps weight AND bias in ``storage_dtype`` (float8_e4m3fn) and casts them up to ``compute_dtype`` for each forward.  The up-cast is exact,
        so the arithmetic is that of bf16 weights that hold fp8-representable values: here the frozen weights are rounded to the storage dtype once
        and kept in bf16 (the 1.9 GB the reference saves on a 24-80 GB card are not what limits a 288 GB one).  Call it where the reference does:
        after loading, before ``add_adapter``.  Returns the diffusers module names that were cast."""
        if compute_dtype != bf16:
            raise ValueError("the MI355X backend computes in bf16")
        cast = []
        for key, view in self._base_views().items():
            mod, _, leaf = key.replace(".base_layer.", ".").rpartition(".")
            if leaf not in ("weight", "bias") or view.dim() == 0:
                continue
            if mod.endswith(("norm_q", "norm_k")) or key.endswith("scale_shift_table"):  # RMSNorm / block Parameters: not a layer type the hook supports
                continue
            if any(re.search(p, mod) for p in skip_modules_pattern):
                continue
            view.copy_(view.to(storage_dtype).to(bf16))
            if leaf == "weight":
                cast.append(mod)
        self._make_transposes()
        return cast

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """Load base weights from a diffusers-style ``LTXVideoTransformer3DModel`` state dict (peft-wrapped
        ``.base_layer.`` names accepted).  LoRA tensors in ``sd`` are loaded too if an adapter exists."""
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items()}
        dev = self.device

        def cp(dst, key):
            dst.copy_(sd[key].to(device=dev, dtype=bf16))

        cp(self.proj_in_w, "proj_in.weight"); cp(self.proj_in_b, "proj_in.bias")
        cp(self.time_l1_w, "time_embed.emb.timestep_embedder.linear_1.weight"); cp(self.time_l1_b, "time_embed.emb.timestep_embedder.linear_1.bias")
        cp(self.time_l2_w, "time_embed.emb.timestep_embedder.linear_2.weight"); cp(self.time_l2_b, "time_embed.emb.timestep_embedder.linear_2.bias")
        cp(self.time_lin_w, "time_embed.linear.weight"); cp(self.time_lin_b, "time_embed.linear.bias")
        cp(self.cap_l1_w, "caption_projection.linear_1.weight"); cp(self.cap_l1_b, "caption_projection.linear_1.bias")
        cp(self.cap_l2_w, "caption_projection.linear_2.weight"); cp(self.cap_l2_b, "caption_projection.linear_2.bias")
        cp(self.table_out, "scale_shift_table")
        cp(self.proj_out_w, "proj_out.weight"); cp(self.proj_out_b, "proj_out.bias")
        D = self.config.inner_dim
        for l in range(self.config.num_layers):
            p = f"transformer_blocks.{l}."
            cp(self.tables[l], p + "scale_shift_table")
            for i, t in enumerate(("to_q", "to_k", "to_v")):
                cp(self.w_qkv[l, i * D:(i + 1) * D], p + f"attn1.{t}.weight")
                cp(self.b_qkv[l, i * D:(i + 1) * D], p + f"attn1.{t}.bias")
            cp(self.norm_q[l], p + "attn1.norm_q.weight"); cp(self.norm_k[l], p + "attn1.norm_k.weight")
            cp(self.w_o[l], p + "attn1.to_out.0.weight"); cp(self.b_o[l], p + "attn1.to_out.0.bias")
            cp(self.w_q2[l], p + "attn2.to_q.weight"); cp(self.b_q2[l], p + "attn2.to_q.bias")
            for i, t in enumerate(("to_k", "to_v")):
                cp(self.w_kv2[l, i * D:(i + 1) * D], p + f"attn2.{t}.weight")
                cp(self.b_kv2[l, i * D:(i + 1) * D], p + f"attn2.{t}.bias")
            cp(self.norm_q2[l], p + "attn2.norm_q.weight"); cp(self.norm_k2[l], p + "attn2.norm_k.weight")
            cp(self.w_o2[l], p + "attn2.to_out.0.weight"); cp(self.b_o2[l], p + "attn2.to_out.0.bias")
            cp(self.w_ff1[l], p + "ff.net.0.proj.weight"); cp(self.b_ff1[l], p + "ff.net.0.proj.bias")
            cp(self.w_ff2[l], p + "ff.net.2.weight"); cp(self.b_ff2[l], p + "ff.net.2.bias")
        self._make_transposes()
        if self.lora_A is not None and any("lora_A" in k for k in sd):
            self.load_lora_state_dict({k: v for k, v in sd.items() if "lora_" in k})

    # ------------------------------------------------------------------ LoRA (peft-compatible surface)
    # ---- --gradient_checkpointing (trainer/sft_trainer/trainer.py:155-157 -> utils/activation_checkpoint.py:24-49) ---------------------------------------
    def apply_activation_checkpointing(self, checkpointing_type: str = "full", n_layer: int = 1) -> "MI355XLTXVideoTransformer3DModel":
        """The reference wraps every transformer block ("full"; the trainer passes nothing else).  Here: the activation workspace holds one block slot
        instead of ``num_layers`` (10.4 GB -> 1.3 GB at BASELINE config 2), the forward keeps only the residual stream and ``ftmi_ltx_backward_range`` runs a
        block's forward kernels again right before its gradient kernels.  Gradients are bit-identical to the un-checkpointed step (deterministic kernels);
        the price is one extra forward per step (``bench.py --gradient-checkpointing``).  "block_skip" would mix kept and recomputed blocks in one
        workspace: not offered for this model (HunyuanVideo, where memory matters, has it)."""
        if checkpointing_type != "full":
            raise ValueError(f"LTX-Video: checkpointing_type {checkpointing_type!r} is not supported (only 'full', what the reference trainer uses)")
        self.gradient_checkpointing = True
        return self

    def enable_gradient_checkpointing(self) -> None:  # diffusers ModelMixin spelling
        self.apply_activation_checkpointing("full")

    def disable_gradient_checkpointing(self) -> None:
        self.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self) -> bool:
        return bool(self.gradient_checkpointing)

    def add_adapter(self, adapter_config=None, adapter_name: str = "default", *, r: Optional[int] = None, lora_alpha: Optional[float] = None,
                    target_modules=None) -> None:
        """Mirror of diffusers ``PeftAdapterMixin.add_adapter(LoraConfig(r, lora_alpha, init_lora_weights=True,
        target_modules))`` as called at finetrainers/trainer/sft_trainer/trainer.py:121-128: A ~ kaiming-uniform(a=sqrt(5)),
        B = 0, fp32 parameters."""
        if adapter_config is not None:
            r = getattr(adapter_config, "r", r)
            lora_alpha = getattr(adapter_config, "lora_alpha", lora_alpha)
            target_modules = getattr(adapter_config, "target_modules", target_modules)
        if target_modules is not None:
            pats = [target_modules] if isinstance(target_modules, str) else list(target_modules)
            names = [f"transformer_blocks.0.{n}" for n in LORA_ORDER]
            hit = [any(re.fullmatch(p, n) or re.search(p, n) for p in pats) for n in names]
            if not all(hit):
                raise ValueError("the MI355X LTX backend fuses LoRA into to_q/to_k/to_v/to_out.0 of attn1+attn2; "
                                 f"target_modules={target_modules!r} does not cover exactly that set")
        if r is None or r <= 0:
            raise ValueError(f"LoRA rank must be positive, got {r}")
        if self.lora_A is not None:
            raise ValueError(f"adapter {adapter_name!r}: an adapter is already attached")
        L, D = self.config.num_layers, self.config.inner_dim
        # The gfx950 kernels work on ranks that are multiples of 64 (one MFMA K-extension step).  Any other rank -- the reference's LTX
        # example trains with --rank 32 (examples/training/sft/ltx_video/crush_smol_lora/train.sh:75) -- is stored padded: the extra rows
        # of A and columns of B are zero and STAY zero (their gradients are exact zeros: x A_pad^T = 0 and dY B_pad = 0; AdamW and weight
        # decay leave a zero parameter with zero gradient at zero), so the padded model is the rank-r model bit for bit.  The Parameters
        # the trainer sees are the [.., :r, :] / [.., :r] views.
        rp = -(-int(r) // 64) * 64
        n = L * 8 * rp * D
        # one flat fp32 buffer [A | B]; the two Parameters are views into it
        self.lora_flat = torch.zeros(2 * n, dtype=torch.float32, device=self.device)
        self._lora_A_full = self.lora_flat[:n].view(L, 8, rp, D)
        self._lora_B_full = self.lora_flat[n:].view(L, 8, D, rp)
        bound = math.sqrt(6.0 / ((1 + 5.0) * D))  # kaiming_uniform_(a=sqrt(5)) on [r, D]: bound = sqrt(6/((1+a^2) fan_in))
        self._lora_A_full[:, :, :r, :].uniform_(-bound, bound)
        self.lora_A = nn.Parameter(self._lora_A_full[:, :, :r, :])
        self.lora_B = nn.Parameter(self._lora_B_full[:, :, :, :r])
        self.lora_rank, self.lora_alpha = int(r), float(lora_alpha if lora_alpha is not None else r)
        self.lora_rank_padded = rp
        r = rp  # working copies and kernels: padded rank
        # bf16 (hi, lo) working copies for the fp32-equivalent LoRA branch (include/ftmi355.h, ftmi_ltx_weights)
        for n, shp in (("lora_a_sp", (L, 8, 2 * r, D)), ("lora_bt_sp", (L, 8, 2 * r, D)), ("lora_b_ext", (L, 8, D, 3 * r)),
                       ("lora_at_ext", (L, 8, D, 3 * r)), ("lora_at_qkv_ext", (L, D, 9 * r))):
            self.register_buffer(n, torch.zeros(shp, dtype=bf16, device=self.device), persistent=False)
        self._lora_versions = None

    def lora_views(self) -> Iterable[Tuple[str, torch.Tensor, torch.Tensor]]:
        """(module path, A view [r,D], B view [D,r]) for each of the L*8 adapters (views into the flat parameters)."""
        for l in range(self.config.num_layers):
            for i, n in enumerate(LORA_ORDER):
                yield f"transformer_blocks.{l}.{n}", self.lora_A[l, i], self.lora_B[l, i]

    def lora_state_dict(self, adapter_name: Optional[str] = None) -> Dict[str, torch.Tensor]:
        """peft-format keys (what ``get_peft_model_state_dict`` returns: adapter name stripped)."""
        out = {}
        mid = f".{adapter_name}" if adapter_name else ""
        for path, a, b in self.lora_views():
            out[f"{path}.lora_A{mid}.weight"] = a
            out[f"{path}.lora_B{mid}.weight"] = b
        return out

    @torch.no_grad()
    def load_lora_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k.replace(".default.", "."): v for k, v in sd.items()}
        for path, a, b in self.lora_views():
            a.copy_(sd[f"{path}.lora_A.weight"].to(a))
            b.copy_(sd[f"{path}.lora_B.weight"].to(b))
        self._lora_versions = None

    # ---- diffusers / peft shaped state dict (what the reference's save and checkpoint paths read, trainer.py:279-306) ----
    def _base_views(self) -> Dict[str, torch.Tensor]:
        """name -> view of the frozen bf16 base weights under diffusers' ``LTXVideoTransformer3DModel`` parameter names (no copies)."""
        D = self.config.inner_dim
        peft = ".base_layer" if self.lora_A is not None else ""
        o = {
            "proj_in.weight": self.proj_in_w, "proj_in.bias": self.proj_in_b, "scale_shift_table": self.table_out,
            "time_embed.emb.timestep_embedder.linear_1.weight": self.time_l1_w, "time_embed.emb.timestep_embedder.linear_1.bias": self.time_l1_b,
            "time_embed.emb.timestep_embedder.linear_2.weight": self.time_l2_w, "time_embed.emb.timestep_embedder.linear_2.bias": self.time_l2_b,
            "time_embed.linear.weight": self.time_lin_w, "time_embed.linear.bias": self.time_lin_b,
            "caption_projection.linear_1.weight": self.cap_l1_w, "caption_projection.linear_1.bias": self.cap_l1_b,
            "caption_projection.linear_2.weight": self.cap_l2_w, "caption_projection.linear_2.bias": self.cap_l2_b,
            "proj_out.weight": self.proj_out_w, "proj_out.bias": self.proj_out_b,
        }
        for l in range(self.config.num_layers):
            p = f"transformer_blocks.{l}."
            o[p + "scale_shift_table"] = self.tables[l]
            for i, t in enumerate(("to_q", "to_k", "to_v")):
                o[p + f"attn1.{t}{peft}.weight"] = self.w_qkv[l, i * D:(i + 1) * D]
                o[p + f"attn1.{t}{peft}.bias"] = self.b_qkv[l, i * D:(i + 1) * D]
            o[p + "attn1.norm_q.weight"], o[p + "attn1.norm_k.weight"] = self.norm_q[l], self.norm_k[l]
            o[p + f"attn1.to_out.0{peft}.weight"], o[p + f"attn1.to_out.0{peft}.bias"] = self.w_o[l], self.b_o[l]
            o[p + f"attn2.to_q{peft}.weight"], o[p + f"attn2.to_q{peft}.bias"] = self.w_q2[l], self.b_q2[l]
            for i, t in enumerate(("to_k", "to_v")):
                o[p + f"attn2.{t}{peft}.weight"] = self.w_kv2[l, i * D:(i + 1) * D]
                o[p + f"attn2.{t}{peft}.bias"] = self.b_kv2[l, i * D:(i + 1) * D]
            o[p + "attn2.norm_q.weight"], o[p + "attn2.norm_k.weight"] = self.norm_q2[l], self.norm_k2[l]
            o[p + f"attn2.to_out.0{peft}.weight"], o[p + f"attn2.to_out.0{peft}.bias"] = self.w_o2[l], self.b_o2[l]
            o[p + "ff.net.0.proj.weight"], o[p + "ff.net.0.proj.bias"] = self.w_ff1[l], self.b_ff1[l]
            o[p + "ff.net.2.weight"], o[p + "ff.net.2.bias"] = self.w_ff2[l], self.b_ff2[l]
        return o

    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False, **kwargs) -> Dict[str, torch.Tensor]:
        """Keys as the reference's peft-wrapped diffusers transformer would emit them: base weights under their diffusers names
        (``.base_layer.`` once an adapter is attached), LoRA tensors as ``...lora_A.default.weight`` -- so ``get_peft_model_state_dict``
        (trainer.py:283) and the DCP ``ModelWrapper`` (parallel/ptd.py:313-321) see the layout they expect.  All values are views."""
        out = destination if destination is not None else {}
        for k, v in self._base_views().items():
            out[prefix + k] = v if keep_vars else v.detach()
        if self.lora_A is not None:
            for k, v in self.lora_state_dict(adapter_name="default").items():
                out[prefix + k] = v if keep_vars else v.detach()
        return out

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        """Accepts what ``state_dict()`` emits (and plain diffusers / peft key variants).  Values are COPIED into the stacked buffers and
        the flat LoRA buffer (``assign=True`` is refused: it would break the aliasing the fused optimiser step relies on)."""
        if assign:
            raise ValueError("MI355XLTXVideoTransformer3DModel.load_state_dict(assign=True) is not supported: parameters are views of one flat buffer")
        sd = {k.replace(".base_layer.", "."): v for k, v in state_dict.items()}
        base = {k: v for k, v in sd.items() if "lora_" not in k}
        lora = {k: v for k, v in sd.items() if "lora_" in k}
        missing = []
        if base:
            want = {k.replace(".base_layer.", ".") for k in self._base_views()}
            missing = sorted(want - set(base))
            if strict and (missing or set(base) - want):
                raise RuntimeError(f"load_state_dict: missing {missing[:5]}{'...' if len(missing) > 5 else ''}, unexpected {sorted(set(base) - want)[:5]}")
            if not missing:
                self.load_diffusers_state_dict(base)
        if lora:
            if self.lora_A is None:
                raise RuntimeError("load_state_dict: LoRA tensors given but no adapter attached (call add_adapter first)")
            self.load_lora_state_dict(lora)
        return nn.modules.module._IncompatibleKeys(missing, [])

    def _assert_flat_aliasing(self) -> None:
        """The fused clip+AdamW updates ``lora_flat`` in place; lora_A / lora_B must still be views of it (``model.to(...)`` or a foreign
        ``load_state_dict(assign=True)`` would silently detach them)."""
        n = self._lora_A_full.numel()
        if self.lora_A.data_ptr() != self.lora_flat.data_ptr() or self.lora_B.data_ptr() != self.lora_flat.data_ptr() + 4 * n:
            raise RuntimeError("lora_A / lora_B no longer alias transformer.lora_flat (the module was moved or re-assigned after add_adapter); "
                               "re-attach the adapter on the target device")

    def lora_grad_views(self) -> Dict[str, torch.Tensor]:
        out = {}
        if self.lora_A is None or self.lora_A.grad is None:
            return out
        for l in range(self.config.num_layers):
            for i, n in enumerate(LORA_ORDER):
                out[f"transformer_blocks.{l}.{n}.lora_A.weight"] = self.lora_A.grad[l, i]
                out[f"transformer_blocks.{l}.{n}.lora_B.weight"] = self.lora_B.grad[l, i]
        return out

    @torch.no_grad()
    def refresh_lora_copies(self, force: bool = False) -> None:
        if self.lora_A is None:
            return
        ver = (self.lora_A._version, self.lora_B._version, self.lora_A.data_ptr(), self.lora_B.data_ptr())
        if not force and ver == self._lora_versions:
            return
        c = self.config
        check(_lib.load().ftmi_lora_refresh(ptr(self._lora_A_full), ptr(self._lora_B_full), ptr(self.lora_a_sp), ptr(self.lora_bt_sp), ptr(self.lora_b_ext),
                                             ptr(self.lora_at_ext), ptr(self.lora_at_qkv_ext), c.num_layers, self.lora_rank_padded, c.inner_dim, stream_ptr()),
              "ftmi_lora_refresh")
        self._lora_versions = ver

    # ------------------------------------------------------------------ C structs
    def _c_config(self, B: int, S: int, T: int, checkpoint: bool = False) -> LtxConfig:
        c = self.config
        return LtxConfig(B=B, S=S, T=T, D=c.inner_dim, H=c.num_attention_heads, L=c.num_layers, C_in=c.in_channels, C_out=c.out_channels,
                         D_ff=c.inner_dim * c.ff_mult, D_cap=c.caption_channels, r=self.lora_rank_padded,
                         lora_scale=(self.lora_alpha / self.lora_rank) if self.lora_rank else 0.0, eps_norm=c.norm_eps, eps_qk=c.qk_norm_eps,
                         gemm_variant=self.gemm_variant, checkpoint=int(bool(checkpoint)),
                         # (a narrow model embedded by zero padding -- ltx_video/narrow.py -- tells the kernels its true width and head width; 0 = this geometry)
                         d_valid=getattr(self, "_narrow", (0, 0))[0], head_dim_valid=getattr(self, "_narrow", (0, 0))[1])

    def _c_weights(self, cos: torch.Tensor, sin: torch.Tensor) -> LtxWeights:
        w = LtxWeights()
        for n in self._frozen_names:
            setattr(w, n, getattr(self, n).data_ptr())
        if self.lora_A is not None:
            w.lora_a_sp, w.lora_bt_sp = self.lora_a_sp.data_ptr(), self.lora_bt_sp.data_ptr()
            w.lora_b_ext, w.lora_at_ext = self.lora_b_ext.data_ptr(), self.lora_at_ext.data_ptr()
            w.lora_at_qkv_ext = self.lora_at_qkv_ext.data_ptr()
        w.rope_cos, w.rope_sin = cos.data_ptr(), sin.data_ptr()
        return w

    def rope_tables(self, num_frames: int, height: int, width: int, rope_interpolation_scale) -> Tuple[torch.Tensor, torch.Tensor]:
        key = (num_frames, height, width, None if rope_interpolation_scale is None else tuple(float(x) for x in rope_interpolation_scale))
        if key not in self._rope_cache:
            cos, sin = ltx_rope_tables(num_frames, height, width, rope_interpolation_scale, dim=self.config.inner_dim)
            self._rope_cache[key] = (cos.to(self.device), sin.to(self.device))
        return self._rope_cache[key]

    # ------------------------------------------------------------------ forward (patch.py:38-50 signature)
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, timestep: torch.Tensor,
                encoder_attention_mask: Optional[torch.Tensor], num_frames: int, height: int, width: int,
                rope_interpolation_scale=None, return_dict: bool = True, *args, **kwargs):
        if not hidden_states.is_cuda:
            raise RuntimeError("MI355XLTXVideoTransformer3DModel runs on the GPU only (no CPU path)")
        B, S, _ = hidden_states.shape
        if S != num_frames * height * width:
            raise ValueError(f"sequence length {S} != num_frames*height*width = {num_frames * height * width}")
        x_t = hidden_states.to(bf16).contiguous()
        text = encoder_hidden_states.to(bf16).contiguous()
        T = text.shape[1]
        # patch.py:55-57: mask -> additive bias, computed in the activations' dtype
        if encoder_attention_mask is None:
            key_bias = None
        elif encoder_attention_mask.ndim == 2:
            key_bias = ((1 - encoder_attention_mask.to(bf16)) * -10000.0).float().contiguous()
        else:
            key_bias = encoder_attention_mask.reshape(B, T).float().contiguous()
        # every token of a sample shares its timestep in SFT (base_specification.py:319-320): one row per sample
        if timestep.ndim == 1:
            tvals = timestep.float().contiguous()
        else:
            tvals = timestep.reshape(B, -1)[:, 0].float().contiguous()
        cos, sin = self.rope_tables(num_frames, height, width, rope_interpolation_scale)
        self.refresh_lora_copies()
        la = self.lora_A if self.lora_A is not None else None
        lb = self.lora_B if self.lora_B is not None else None
        out = _LTXDiTFunction.apply(self, x_t, text, key_bias, tvals, cos, sin, la, lb)
        if not return_dict:
            return (out,)
        return {"sample": out}

    # ------------------------------------------------------------------ workspace pool
    def _acquire_workspace(self, nbytes: int, device: torch.device) -> torch.Tensor:
        for i, t in enumerate(self._ws_pool):
            if t.numel() == nbytes and t.device == device:
                return self._ws_pool.pop(i)
        self._ws_pool.clear()  # a different problem size: do not hold on to the old blocks
        return torch.empty((nbytes,), dtype=torch.uint8, device=device)

    def _release_workspace(self, ws: Optional[torch.Tensor]) -> None:
        if ws is not None and len(self._ws_pool) < 2:
            self._ws_pool.append(ws)

    # ------------------------------------------------------------------ debugging / tests
    def workspace_tensor(self, name: str, layer: int, shape, dtype=bf16) -> torch.Tensor:
        """View of a stashed activation of the most recent forward (tests / debugging only)."""
        cfg, ws = self._last_workspace
        off = ctypes.c_size_t(0)
        check(_lib.load().ftmi_ltx_workspace_offset(ctypes.byref(cfg), name.encode(), layer, ctypes.byref(off)), "ftmi_ltx_workspace_offset")
        n = 1
        for s in shape:
            n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        return ws[off.value:off.value + nbytes].view(dtype).view(*shape)
