"""CPU restatement (oracle) of the reference LTX-Video LoRA SFT training step.

TEST INFRASTRUCTURE -- see ``oracle/__init__.py``.  Pure PyTorch on CPU; it executes
the same torch op sequence the reference executes (so bf16 rounding points are the
ones the reference PyTorch-CPU path has), with the LoRA branch in fp32 as the
reference's PTD backend runs it.

What each piece follows (paths relative to /root/reference):

* model-level forward ........ finetrainers/patches/models/ltx_video/patch.py:38-127
* RoPE apply ................. finetrainers/patches/models/ltx_video/patch.py:23-33
* RMSNorm .................... finetrainers/patches/dependencies/diffusers/rms_norm.py:17-29
* noising / pack / target .... finetrainers/models/ltx_video/base_specification.py:271-345,427-459
                               finetrainers/functional/diffusion.py:4-11
* sigma sampling ............. finetrainers/utils/diffusion.py:38-63,84-114
* loss ....................... finetrainers/trainer/sft_trainer/trainer.py:463-481
* clip ....................... finetrainers/utils/torch.py:99-161,299-374
* optimiser .................. finetrainers/optimizer.py:117-125 (torch.optim.AdamW)
* LoRA config / dtypes ....... finetrainers/trainer/sft_trainer/trainer.py:121-136,
                               finetrainers/trainer/sft_trainer/config.py:24-26
* module tree / hyper-params . tests/models/ltx_video/_test_tp.py:29-59,186-245

[upstream] parts (NOT in /root/reference; restated from the published algorithm of
diffusers 0.32/0.33 ``models/transformers/transformer_ltx.py``, ``models/embeddings.py``,
``models/normalization.py``, ``models/attention.py``, ``models/activations.py`` and
peft 0.14 ``tuners/lora/layer.py``): ``LTXVideoTransformerBlock``, the attention
processor, ``LTXVideoRotaryPosEmbed``, ``AdaLayerNormSingle``, ``PixArtAlphaTextProjection``,
``FeedForward``, ``FlowMatchEulerDiscreteScheduler.sigmas``, ``compute_loss_weighting_for_sd3``,
LoRA ``Linear``.  Parity for those is therefore *unpinned* (no upstream values
available offline) and is anchored on structure: parameter names/shapes/counts
(tests/test_oracle.py) and the reference's own call sites.
"""

from __future__ import annotations

import math
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------


@dataclass
class LTXConfig:
    """Hyper-parameters of ``LTXVideoTransformer3DModel`` (tests/models/ltx_video/_test_tp.py:29-59)."""

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
    qk_norm_eps: float = 1e-5  # [upstream] diffusers Attention(eps=1e-5) default
    ff_mult: int = 4
    text_seq_len: int = 128  # finetrainers/models/ltx_video/base_specification.py:233

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def production(num_layers: int = 28) -> "LTXConfig":
        return LTXConfig(num_layers=num_layers)

    @staticmethod
    def dummy() -> "LTXConfig":
        """The reference's tiny fixture (tests/models/ltx_video/base_specification.py:48-58)."""
        return LTXConfig(
            in_channels=8,
            out_channels=8,
            num_attention_heads=4,
            attention_head_dim=8,
            cross_attention_dim=32,
            num_layers=1,
            caption_channels=32,
        )


# --------------------------------------------------------------------------------------
# [upstream] building blocks
# --------------------------------------------------------------------------------------


class RMSNorm(nn.Module):
    """diffusers RMSNorm with the reference's patched forward (rms_norm.py:17-29, torch>=2.4 branch)."""

    def __init__(self, dim: int, eps: float, elementwise_affine: bool = True):
        super().__init__()
        self.eps = eps
        self.dim = dim
        self.weight = nn.Parameter(torch.ones(dim)) if elementwise_affine else None
        self.bias = None

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        input_dtype = hidden_states.dtype
        if self.weight is not None and self.weight.dtype in (torch.float16, torch.bfloat16):
            hidden_states = hidden_states.to(self.weight.dtype)
        hidden_states = F.rms_norm(hidden_states, (hidden_states.shape[-1],), weight=self.weight, eps=self.eps)
        return hidden_states.to(input_dtype)


class LoraLinear(nn.Module):
    """[upstream] peft ``lora.Linear`` around a frozen ``nn.Linear`` (adapter name "default").

    y = base(x) + lora_B(lora_A(x.to(A.dtype))) * (alpha / r);  result cast back to base dtype.
    """

    def __init__(self, base: nn.Linear, r: int, alpha: float):
        super().__init__()
        self.base_layer = base
        self.r = r
        self.scaling = alpha / r
        self.lora_A = nn.ModuleDict({"default": nn.Linear(base.in_features, r, bias=False)})
        self.lora_B = nn.ModuleDict({"default": nn.Linear(r, base.out_features, bias=False)})
        # peft init_lora_weights=True: A kaiming-uniform(a=sqrt(5)), B zeros
        nn.init.kaiming_uniform_(self.lora_A["default"].weight, a=math.sqrt(5))
        nn.init.zeros_(self.lora_B["default"].weight)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        result = self.base_layer(x)
        torch_result_dtype = result.dtype
        a = self.lora_A["default"]
        b = self.lora_B["default"]
        xx = x.to(a.weight.dtype)
        result = result + b(a(xx)) * self.scaling
        return result.to(torch_result_dtype)


def apply_rotary_emb(x: torch.Tensor, freqs: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """patch.py:23-33 (same math as upstream's)."""
    cos, sin = freqs
    x_real, x_imag = x.unflatten(2, (-1, 2)).unbind(-1)  # [B, S, D // 2]
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(2)
    out = (x.float() * cos + x_rotated.float() * sin).to(x.dtype)
    return out


def native_sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """What the reference executes: its NATIVE provider (attention_dispatch.py:938-962) is a pass-through to
    ``torch.nn.functional.scaled_dot_product_attention`` with dropout 0, non-causal, default scale.  On the CPU
    torch dispatches bf16 inputs to ``aten::_scaled_dot_product_flash_attention_for_cpu`` (+ ``..._backward``;
    recorded by tests/test_oracle.py::test_native_sdpa_dispatch): scores and softmax statistics in fp32, the
    probabilities rounded to bf16 before P.V, dS rounded to bf16 before the dQ / dK products -- the same rounding
    points as a fused GPU kernel.  Inputs [B, H, S, d]."""
    return F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=False)


def sdpa_math(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, attn_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """NOT on the oracle's path: the fp32-probability "math" form softmax(q k^T / sqrt(d) + mask) v on the
    bf16-rounded inputs, kept as a comparison point for the attention-kernel tests (the reference's own provider
    tests compare against ``_NATIVE_MATH``, tests/models/attention_dispatch.py:48-78)."""
    scale = 1.0 / math.sqrt(q.shape[-1])
    s = torch.matmul(q.float(), k.float().transpose(-1, -2)) * scale
    if attn_mask is not None:
        s = s + attn_mask.float()
    p = torch.softmax(s, dim=-1)
    return torch.matmul(p, v.float()).to(q.dtype)


class Attention(nn.Module):
    """[upstream] diffusers ``Attention`` as configured by LTX (qk_norm="rms_norm_across_heads",
    bias=True, out_bias=True) + ``LTXVideoAttentionProcessor2_0``."""

    def __init__(self, cfg: LTXConfig, cross: bool):
        super().__init__()
        d = cfg.inner_dim
        kv_in = cfg.cross_attention_dim if cross else d
        self.heads = cfg.num_attention_heads
        self.norm_q = RMSNorm(d, eps=cfg.qk_norm_eps, elementwise_affine=True)
        self.norm_k = RMSNorm(d, eps=cfg.qk_norm_eps, elementwise_affine=True)
        self.to_q = nn.Linear(d, d, bias=True)
        self.to_k = nn.Linear(kv_in, d, bias=True)
        self.to_v = nn.Linear(kv_in, d, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(d, d, bias=True), nn.Dropout(0.0)])

    def forward(
        self,
        hidden_states: torch.Tensor,
        encoder_hidden_states: Optional[torch.Tensor] = None,
        attention_mask: Optional[torch.Tensor] = None,
        image_rotary_emb: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
    ) -> torch.Tensor:
        batch_size, sequence_length, _ = (
            hidden_states.shape if encoder_hidden_states is None else encoder_hidden_states.shape
        )
        if attention_mask is not None:
            # prepare_attention_mask: [B,1,T] -> repeat over heads -> view [B, H, 1, T]
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
            attention_mask = attention_mask.view(batch_size, self.heads, -1, attention_mask.shape[-1])
        if encoder_hidden_states is None:
            encoder_hidden_states = hidden_states

        query = self.to_q(hidden_states)
        key = self.to_k(encoder_hidden_states)
        value = self.to_v(encoder_hidden_states)

        query = self.norm_q(query)
        key = self.norm_k(key)

        if image_rotary_emb is not None:
            query = apply_rotary_emb(query, image_rotary_emb)
            key = apply_rotary_emb(key, image_rotary_emb)

        query = query.unflatten(2, (self.heads, -1)).transpose(1, 2)
        key = key.unflatten(2, (self.heads, -1)).transpose(1, 2)
        value = value.unflatten(2, (self.heads, -1)).transpose(1, 2)

        hidden_states = native_sdpa(query, key, value, attention_mask)
        hidden_states = hidden_states.transpose(1, 2).flatten(2, 3)
        hidden_states = hidden_states.to(query.dtype)

        hidden_states = self.to_out[0](hidden_states)
        hidden_states = self.to_out[1](hidden_states)
        return hidden_states


class GELUProj(nn.Module):
    """[upstream] diffusers ``GELU(dim_in, dim_out, approximate="tanh")``."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out, bias=True)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return F.gelu(self.proj(x), approximate="tanh")


class FeedForward(nn.Module):
    """[upstream] diffusers ``FeedForward(activation_fn="gelu-approximate")``: net = [GELU, Dropout, Linear]."""

    def __init__(self, dim: int, mult: int):
        super().__init__()
        self.net = nn.ModuleList([GELUProj(dim, dim * mult), nn.Dropout(0.0), nn.Linear(dim * mult, dim, bias=True)])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        for m in self.net:
            x = m(x)
        return x


class LTXVideoTransformerBlock(nn.Module):
    """[upstream] ``LTXVideoTransformerBlock`` (module tree: _test_tp.py:207-241)."""

    def __init__(self, cfg: LTXConfig):
        super().__init__()
        d = cfg.inner_dim
        self.norm1 = RMSNorm(d, eps=cfg.norm_eps, elementwise_affine=False)
        self.attn1 = Attention(cfg, cross=False)
        self.norm2 = RMSNorm(d, eps=cfg.norm_eps, elementwise_affine=False)
        self.attn2 = Attention(cfg, cross=True)
        self.ff = FeedForward(d, cfg.ff_mult)
        self.scale_shift_table = nn.Parameter(torch.randn(6, d) / d**0.5)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb=None, encoder_attention_mask=None):
        batch_size = hidden_states.size(0)
        norm_hidden_states = self.norm1(hidden_states)

        num_ada_params = self.scale_shift_table.shape[0]
        ada_values = self.scale_shift_table[None, None] + temb.reshape(batch_size, temb.size(1), num_ada_params, -1)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = ada_values.unbind(dim=2)
        norm_hidden_states = norm_hidden_states * (1 + scale_msa) + shift_msa

        attn_hidden_states = self.attn1(
            hidden_states=norm_hidden_states, encoder_hidden_states=None, image_rotary_emb=image_rotary_emb
        )
        hidden_states = hidden_states + attn_hidden_states * gate_msa

        attn_hidden_states = self.attn2(
            hidden_states,
            encoder_hidden_states=encoder_hidden_states,
            image_rotary_emb=None,
            attention_mask=encoder_attention_mask,
        )
        hidden_states = hidden_states + attn_hidden_states
        norm_hidden_states = self.norm2(hidden_states) * (1 + scale_mlp) + shift_mlp

        ff_output = self.ff(norm_hidden_states)
        hidden_states = hidden_states + ff_output * gate_mlp
        return hidden_states


class LTXVideoRotaryPosEmbed(nn.Module):
    """[upstream] ``LTXVideoRotaryPosEmbed`` -- always computed in fp32; returns (cos, sin) [B,S,dim]."""

    def __init__(self, dim, base_num_frames=20, base_height=2048, base_width=2048, patch_size=1, patch_size_t=1, theta=10000.0):
        super().__init__()
        self.dim = dim
        self.base_num_frames = base_num_frames
        self.base_height = base_height
        self.base_width = base_width
        self.patch_size = patch_size
        self.patch_size_t = patch_size_t
        self.theta = theta

    def forward(self, hidden_states, num_frames, height, width, rope_interpolation_scale=None):
        batch_size = hidden_states.size(0)
        dev = hidden_states.device
        grid_h = torch.arange(height, dtype=torch.float32, device=dev)
        grid_w = torch.arange(width, dtype=torch.float32, device=dev)
        grid_f = torch.arange(num_frames, dtype=torch.float32, device=dev)
        grid = torch.meshgrid(grid_f, grid_h, grid_w, indexing="ij")
        grid = torch.stack(grid, dim=0)
        grid = grid.unsqueeze(0).repeat(batch_size, 1, 1, 1, 1)

        if rope_interpolation_scale is not None:
            grid[:, 0:1] = grid[:, 0:1] * rope_interpolation_scale[0] * self.patch_size_t / self.base_num_frames
            grid[:, 1:2] = grid[:, 1:2] * rope_interpolation_scale[1] * self.patch_size / self.base_height
            grid[:, 2:3] = grid[:, 2:3] * rope_interpolation_scale[2] * self.patch_size / self.base_width

        grid = grid.flatten(2, 4).transpose(1, 2)

        start = 1.0
        end = self.theta
        freqs = self.theta ** torch.linspace(
            math.log(start, self.theta), math.log(end, self.theta), self.dim // 6, device=dev, dtype=torch.float32
        )
        freqs = freqs * math.pi / 2.0
        freqs = freqs * (grid.unsqueeze(-1) * 2 - 1)
        freqs = freqs.transpose(-1, -2).flatten(2)

        cos_freqs = freqs.cos().repeat_interleave(2, dim=-1)
        sin_freqs = freqs.sin().repeat_interleave(2, dim=-1)

        if self.dim % 6 != 0:
            cos_padding = torch.ones_like(cos_freqs[:, :, : self.dim % 6])
            sin_padding = torch.zeros_like(cos_freqs[:, :, : self.dim % 6])
            cos_freqs = torch.cat([cos_padding, cos_freqs], dim=-1)
            sin_freqs = torch.cat([sin_padding, sin_freqs], dim=-1)
        return cos_freqs, sin_freqs


def get_timestep_embedding(timesteps: torch.Tensor, embedding_dim: int) -> torch.Tensor:
    """[upstream] diffusers ``get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)``."""
    half_dim = embedding_dim // 2
    exponent = -math.log(10000) * torch.arange(start=0, end=half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - 0.0)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)  # flip_sin_to_cos
    return emb


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels: int, time_embed_dim: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim, bias=True)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim, bias=True)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


class Timesteps(nn.Module):
    def __init__(self, num_channels: int):
        super().__init__()
        self.num_channels = num_channels

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels)


class PixArtAlphaCombinedTimestepSizeEmbeddings(nn.Module):
    def __init__(self, embedding_dim: int):
        super().__init__()
        self.time_proj = Timesteps(256)
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)

    def forward(self, timestep, batch_size, hidden_dtype):
        timesteps_proj = self.time_proj(timestep)
        return self.timestep_embedder(timesteps_proj.to(dtype=hidden_dtype))


class AdaLayerNormSingle(nn.Module):
    """[upstream] ``AdaLayerNormSingle(use_additional_conditions=False)``."""

    def __init__(self, embedding_dim: int):
        super().__init__()
        self.emb = PixArtAlphaCombinedTimestepSizeEmbeddings(embedding_dim)
        self.silu = nn.SiLU()
        self.linear = nn.Linear(embedding_dim, 6 * embedding_dim, bias=True)

    def forward(self, timestep, batch_size=None, hidden_dtype=None):
        embedded_timestep = self.emb(timestep, batch_size=batch_size, hidden_dtype=hidden_dtype)
        return self.linear(self.silu(embedded_timestep)), embedded_timestep


class PixArtAlphaTextProjection(nn.Module):
    def __init__(self, in_features: int, hidden_size: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden_size, bias=True)
        self.act_1 = nn.GELU(approximate="tanh")
        self.linear_2 = nn.Linear(hidden_size, hidden_size, bias=True)

    def forward(self, caption):
        return self.linear_2(self.act_1(self.linear_1(caption)))


class LTXVideoTransformer3DModel(nn.Module):
    """Module tree per _test_tp.py:186-245; forward restates patch.py:38-127 (gradient
    checkpointing branch omitted -- it changes memory, not values)."""

    def __init__(self, cfg: LTXConfig):
        super().__init__()
        self.cfg = cfg
        d = cfg.inner_dim
        self.proj_in = nn.Linear(cfg.in_channels, d)
        self.scale_shift_table = nn.Parameter(torch.randn(2, d) / d**0.5)
        self.time_embed = AdaLayerNormSingle(d)
        self.caption_projection = PixArtAlphaTextProjection(cfg.caption_channels, d)
        self.rope = LTXVideoRotaryPosEmbed(
            dim=d, patch_size=cfg.patch_size, patch_size_t=cfg.patch_size_t, theta=10000.0
        )
        self.transformer_blocks = nn.ModuleList([LTXVideoTransformerBlock(cfg) for _ in range(cfg.num_layers)])
        self.norm_out = nn.LayerNorm(d, eps=1e-6, elementwise_affine=False)
        self.proj_out = nn.Linear(d, cfg.out_channels)

    def forward(
        self,
        hidden_states: torch.Tensor,
        encoder_hidden_states: torch.Tensor,
        timestep: torch.Tensor,
        encoder_attention_mask: torch.Tensor,
        num_frames: int,
        height: int,
        width: int,
        rope_interpolation_scale=None,
        return_dict: bool = True,
    ):
        image_rotary_emb = self.rope(hidden_states, num_frames, height, width, rope_interpolation_scale)

        if encoder_attention_mask is not None and encoder_attention_mask.ndim == 2:
            encoder_attention_mask = (1 - encoder_attention_mask.to(hidden_states.dtype)) * -10000.0
            encoder_attention_mask = encoder_attention_mask.unsqueeze(1)

        batch_size = hidden_states.size(0)
        if timestep.ndim == 1:
            timestep = timestep.view(-1, 1, 1).expand(-1, *hidden_states.shape[1:-1], -1)

        temb, embedded_timestep = self.time_embed(
            timestep.flatten(), batch_size=batch_size, hidden_dtype=hidden_states.dtype
        )
        temb = temb.view(batch_size, *hidden_states.shape[1:-1], temb.size(-1))
        embedded_timestep = embedded_timestep.view(batch_size, *hidden_states.shape[1:-1], embedded_timestep.size(-1))

        hidden_states = self.proj_in(hidden_states)

        encoder_hidden_states = self.caption_projection(encoder_hidden_states)
        encoder_hidden_states = encoder_hidden_states.view(batch_size, -1, hidden_states.size(-1))

        for block in self.transformer_blocks:
            hidden_states = block(
                hidden_states=hidden_states,
                encoder_hidden_states=encoder_hidden_states,
                temb=temb,
                image_rotary_emb=image_rotary_emb,
                encoder_attention_mask=encoder_attention_mask,
            )

        scale_shift_values = self.scale_shift_table[None, None] + embedded_timestep[:, :, None]
        shift, scale = scale_shift_values[:, :, 0], scale_shift_values[:, :, 1]

        hidden_states = self.norm_out(hidden_states)
        hidden_states = hidden_states * (1 + scale) + shift
        output = self.proj_out(hidden_states)
        if not return_dict:
            return (output,)
        return {"sample": output}


# --------------------------------------------------------------------------------------
# accumulation-order variant (for the parity-tolerance "triangle": fp32 oracle <-> bf16 oracle <-> kernel)
# --------------------------------------------------------------------------------------


class _ChunkedLinear(torch.autograd.Function):
    """``F.linear`` for a frozen bf16 weight with the SAME rounding points as torch's (fp32 accumulation, one round
    to bf16 at the output, in forward and in the input gradient) but a different fp32 summation order: partial
    products over chunks of the reduction axis are summed left to right.  Any two correct bf16 implementations of
    the reference graph (CPU vs GPU, two BLAS libraries, two tile shapes) differ at least like this."""

    @staticmethod
    def forward(ctx, x, w, b, chunk):
        ctx.save_for_backward(w)
        ctx.chunk = chunk
        acc = None
        for k0 in range(0, x.shape[-1], chunk):
            t = x[..., k0:k0 + chunk].float() @ w[:, k0:k0 + chunk].float().t()
            acc = t if acc is None else acc + t
        if b is not None:
            acc = acc + b.float()
        return acc.to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        acc = None
        for n0 in range(0, w.shape[0], ctx.chunk):
            t = g[..., n0:n0 + ctx.chunk].float() @ w[n0:n0 + ctx.chunk].float()
            acc = t if acc is None else acc + t
        return acc.to(g.dtype), None, None, None


class accumulation_order_variant:
    """Context manager: every frozen bf16 ``F.linear`` of the oracle sums its reduction in ``chunk``-wide partials.
    ``rel_l2(grads(variant), grads(oracle))`` is the op-order noise floor of the LoRA gradients: the part of a
    kernel-vs-oracle difference that no implementation can remove (tests/test_oracle.py pins its size)."""

    def __init__(self, chunk: int = 512):
        self.chunk = chunk

    def __enter__(self):
        self._orig = F.linear
        orig, chunk = self._orig, self.chunk

        def linear(x, w, b=None):
            if x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and not w.requires_grad and x.shape[-1] > chunk:
                return _ChunkedLinear.apply(x, w, b, chunk)
            return orig(x, w, b)

        F.linear = linear
        torch.nn.functional.linear = linear
        return self

    def __exit__(self, *exc):
        F.linear = self._orig
        torch.nn.functional.linear = self._orig
        return False


class attention_order_variant:
    """Context manager: every scaled-dot-product attention of the oracle sees its keys (and values, and the mask columns) in a fixed
    pseudo-random ORDER.  Mathematically the identity; numerically another tile / online-softmax-rescale / accumulation order with the same
    rounding points -- what any two flash-attention kernels (torch's CPU kernel and a GPU kernel) differ by."""

    def __enter__(self):
        import sys

        mod = sys.modules[__name__]
        self._mod, self._orig = mod, mod.native_sdpa

        def permuted(q, k, v, attn_mask):
            perm = torch.randperm(k.shape[2], generator=torch.Generator().manual_seed(k.shape[2]))
            m = None if attn_mask is None else attn_mask[..., perm]
            return self._orig(q, k[:, :, perm], v[:, :, perm], m)

        mod.native_sdpa = permuted
        return self

    def __exit__(self, *exc):
        self._mod.native_sdpa = self._orig
        return False


class _FusedQKVBase(torch.autograd.Function):
    """The three frozen projections of a self-attention with ONE input gradient: dX = dQ Wq + dK Wk + dV Wv summed in fp32 and rounded to
    bf16 once (eager autograd rounds each of the three products to bf16 and adds them in bf16: two more rounding points)."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv):
        ctx.save_for_backward(wq, wk, wv)
        return F.linear(x, wq, bq), F.linear(x, wk, bk), F.linear(x, wv, bv)

    @staticmethod
    def backward(ctx, gq, gk, gv):
        wq, wk, wv = ctx.saved_tensors
        dx = gq.float() @ wq.float() + gk.float() @ wk.float() + gv.float() @ wv.float()
        return dx.to(gq.dtype), None, None, None, None, None, None


class fused_qkv_dgrad_variant:
    """Context manager: the self-attentions' q | k | v base projections share one fp32 input-gradient accumulator (``_FusedQKVBase``) -- the
    MI355X kernels' deliberate difference (ii) of DESIGN.md section 3 (one fused dgrad GEMM over the concatenated weights)."""

    def __enter__(self):
        self._orig = Attention.forward

        def forward(att, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None):
            if encoder_hidden_states is not None or not isinstance(att.to_q, LoraLinear):
                return self._orig(att, hidden_states, encoder_hidden_states, attention_mask, image_rotary_emb)
            lq, lk, lv = att.to_q, att.to_k, att.to_v
            qb, kb, vb = _FusedQKVBase.apply(hidden_states, lq.base_layer.weight, lq.base_layer.bias, lk.base_layer.weight, lk.base_layer.bias,
                                             lv.base_layer.weight, lv.base_layer.bias)

            def lora(l, base):  # LoraLinear.forward with the base result given
                xx = hidden_states.to(l.lora_A["default"].weight.dtype)
                return (base + l.lora_B["default"](l.lora_A["default"](xx)) * l.scaling).to(base.dtype)

            query, key, value = lora(lq, qb), lora(lk, kb), lora(lv, vb)
            query, key = att.norm_q(query), att.norm_k(key)
            if image_rotary_emb is not None:
                query, key = apply_rotary_emb(query, image_rotary_emb), apply_rotary_emb(key, image_rotary_emb)
            sp = lambda z: z.unflatten(2, (att.heads, -1)).transpose(1, 2)
            import sys
            o = sys.modules[__name__].native_sdpa(sp(query), sp(key), sp(value), None).transpose(1, 2).flatten(2, 3).to(query.dtype)
            return att.to_out[1](att.to_out[0](o))

        Attention.forward = forward
        return self

    def __exit__(self, *exc):
        Attention.forward = self._orig
        return False


def lora_grads(model: nn.Module, inp: "StepInputs", flow_weighting_scheme: str = "none") -> Tuple[Dict[str, torch.Tensor], float]:
    """({peft name without the adapter infix: gradient}, loss) of one forward + backward of the oracle."""
    for p in model.parameters():
        p.grad = None
    loss = forward_loss(model, inp, flow_weighting_scheme, contiguous_hidden_states=True)[0]
    loss.backward()
    return {n.replace(".default", ""): p.grad.detach().clone() for n, p in lora_parameters(model)}, loss.item()


def grads_rel_l2(a: Dict[str, torch.Tensor], b: Dict[str, torch.Tensor]) -> Tuple[float, float]:
    """(all gradients as one vector, worst single adapter tensor) relative L2 error of ``a`` against ``b``."""
    num = den = 0.0
    worst = 0.0
    for k, gb in b.items():
        d = (a[k].float().cpu() - gb.float()).pow(2).sum().item()
        n = gb.float().pow(2).sum().item()
        num, den = num + d, den + n
        worst = max(worst, math.sqrt(d / max(n, 1e-60)))
    return math.sqrt(num / max(den, 1e-60)), worst


# --------------------------------------------------------------------------------------
# LoRA injection (sft_trainer/trainer.py:121-136, sft_trainer/config.py:24-26)
# --------------------------------------------------------------------------------------

LORA_TARGETS = ("to_q", "to_k", "to_v", "to_out.0")


LAYERWISE_UPCASTING_SKIP_PATTERNS = ("patch_embed", "pos_embed", "x_embedder", "context_embedder", "time_embed", "^proj_in$", "^proj_out$", "norm")  # args.py:395


def apply_layerwise_casting(model: nn.Module, storage_dtype: torch.dtype = torch.float8_e4m3fn, skip_modules_pattern=LAYERWISE_UPCASTING_SKIP_PATTERNS) -> List[str]:
    """trainer/sft_trainer/trainer.py:111-118 -> [upstream, unpinned] diffusers ``apply_layerwise_casting``: Linear layers whose module name matches none of
    the skip patterns (regex search) store weight and bias in ``storage_dtype`` and up-cast to the compute dtype for every forward.  The up-cast is exact,
    so the model computes with bf16 tensors holding storage-dtype-representable values: restated as a one-time rounding.  Call before ``add_lora``."""
    import re

    cast = []
    for name, mod in model.named_modules():
        if isinstance(mod, nn.Linear) and not any(re.search(p, name) for p in skip_modules_pattern):
            with torch.no_grad():
                mod.weight.copy_(mod.weight.to(storage_dtype).to(mod.weight.dtype))
                if mod.bias is not None:
                    mod.bias.copy_(mod.bias.to(storage_dtype).to(mod.bias.dtype))
            cast.append(name)
    return cast


def add_lora(model: LTXVideoTransformer3DModel, rank: int = 64, alpha: float = 64.0) -> List[str]:
    """Wrap to_q/to_k/to_v/to_out.0 of attn1+attn2 in every block (the default target regex),
    freeze everything else, LoRA params fp32 (``cast_training_params``)."""
    for p in model.parameters():
        p.requires_grad_(False)
    names = []
    for li, block in enumerate(model.transformer_blocks):
        for an in ("attn1", "attn2"):
            attn = getattr(block, an)
            for t in ("to_q", "to_k", "to_v"):
                setattr(attn, t, LoraLinear(getattr(attn, t), rank, alpha))
                names.append(f"transformer_blocks.{li}.{an}.{t}")
            attn.to_out[0] = LoraLinear(attn.to_out[0], rank, alpha)
            names.append(f"transformer_blocks.{li}.{an}.to_out.0")
    for n, p in model.named_parameters():
        if "lora_" in n:
            p.data = p.data.float()
            p.requires_grad_(True)
    return names


def lora_parameters(model: nn.Module) -> List[Tuple[str, nn.Parameter]]:
    return [(n, p) for n, p in model.named_parameters() if "lora_" in n]


def build_model(cfg: LTXConfig, seed: int = 0, dtype: torch.dtype = torch.bfloat16, rank: int = 64,
                alpha: float = 64.0, lora_b_std: Optional[float] = None) -> LTXVideoTransformer3DModel:
    """``torch.manual_seed(seed)``, default init (as the reference's fixtures do), cast base to
    ``dtype``, inject LoRA.  ``lora_b_std`` != None re-draws B ~ N(0, std) so dA is not
    identically zero (a parity-only variation; the true init is B = 0)."""
    torch.manual_seed(seed)
    model = LTXVideoTransformer3DModel(cfg)
    # norm_q / norm_k weights default to ones; perturb so the affine part is exercised
    g = torch.Generator().manual_seed(seed + 1)
    for n, p in model.named_parameters():
        if n.endswith("norm_q.weight") or n.endswith("norm_k.weight"):
            p.data = 1.0 + 0.1 * torch.randn(p.shape, generator=g)
    model.to(dtype)
    if rank > 0:
        add_lora(model, rank, alpha)
        if lora_b_std is not None:
            for n, p in model.named_parameters():
                if "lora_B" in n:
                    p.data = torch.randn(p.shape, generator=g) * lora_b_std
    return model


# --------------------------------------------------------------------------------------
# spec-level forward (ltx_video/base_specification.py:271-345)
# --------------------------------------------------------------------------------------


def flow_match_xt(x0, n, t):  # functional/diffusion.py:4-6
    return (1.0 - t) * x0 + t * n


def flow_match_target(n, x0):  # functional/diffusion.py:9-11
    return n - x0


def normalize_latents(latents, latents_mean, latents_std, scaling_factor: float = 1.0):
    """base_specification.py:427-436 with the evident per-channel intent for B>1:
    the reference's ``view(batch_size, -1, 1, 1, 1)`` only broadcasts at B == 1, where the two agree."""
    latents_mean = latents_mean.view(1, -1, 1, 1, 1).to(device=latents.device)
    latents_std = latents_std.view(1, -1, 1, 1, 1).to(device=latents.device)
    return ((latents.float() - latents_mean) * scaling_factor / latents_std).to(latents)


def posterior_sample(moments: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """base_specification.py:285-289 (compute_posterior = False, the --enable_precomputation path): [upstream, unpinned] diffusers
    ``DiagonalGaussianDistribution(moments).sample()`` -- mean, logvar = chunk(moments, 2, dim=1); logvar clamped to [-30, 20];
    std = exp(0.5 * logvar); x = mean + std * eps, every op in the moments' dtype (bf16: one rounding per op)."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    logvar = torch.clamp(logvar, -30.0, 20.0)
    std = torch.exp(0.5 * logvar)
    return mean + std * eps.to(moments.dtype)


def pack_latents(latents, patch_size: int = 1, patch_size_t: int = 1):
    """base_specification.py:438-459."""
    b, c, f, h, w = latents.shape
    pf, ph, pw = f // patch_size_t, h // patch_size, w // patch_size
    latents = latents.reshape(b, -1, pf, patch_size_t, ph, patch_size, pw, patch_size)
    return latents.permute(0, 2, 4, 6, 1, 3, 5, 7).flatten(4, 7).flatten(1, 3)


def spec_forward(
    transformer: nn.Module,
    latents: torch.Tensor,  # [B, C, F, H, W] (un-normalised)
    latents_mean: torch.Tensor,
    latents_std: torch.Tensor,
    encoder_hidden_states: torch.Tensor,
    encoder_attention_mask: torch.Tensor,
    sigmas: torch.Tensor,  # [B,1,1,1,1] fp32
    noise: Optional[torch.Tensor] = None,  # injected N(0,1) in latents' dtype/shape
    first_frame_sigma: Optional[torch.Tensor] = None,  # injected => first-frame branch ON
    generator: Optional[torch.Generator] = None,
    patch_size: int = 1,
    patch_size_t: int = 1,
    contiguous_hidden_states: bool = False,
):
    """Returns (pred, target, sigmas[B,S,1]).  Noise and the 10 % first-frame branch are
    *injected* rather than drawn (the branch uses Python's global RNG).

    ``contiguous_hidden_states``: the reference hands the transformer ``noisy_latents.to(latents)``, a
    NON-contiguous view (base_specification.py:322).  On the CPU, torch's bf16 ``F.linear`` takes a slower
    path for non-contiguous inputs that rounds twice (matmul -> bf16, + bias -> bf16) instead of once, which
    perturbs ~28 % of proj_in's outputs by one bf16 ulp.  That is an artefact of running the reference on the
    CPU, not part of its algorithm (on a GPU the bias is fused and rounding happens once).  Default False
    reproduces the reference-on-CPU bit-for-bit (golden fixtures); the GPU parity tests pass True."""
    num_frames, height, width = latents.shape[2:]
    latents = normalize_latents(latents, latents_mean, latents_std)
    if noise is None:
        noise = torch.zeros_like(latents).normal_(generator=generator)

    if first_frame_sigma is not None:
        first_frame_sigma = torch.min(first_frame_sigma, sigmas.new_full(sigmas.shape, 0.25))
        lf, lr = latents[:, :, :1], latents[:, :, 1:]
        nf = flow_match_xt(lf, noise[:, :, :1], first_frame_sigma)
        nr = flow_match_xt(lr, noise[:, :, 1:], sigmas)
        noisy_latents = torch.cat([nf, nr], dim=2)
    else:
        noisy_latents = flow_match_xt(latents, noise, sigmas)

    latents_p = pack_latents(latents, patch_size, patch_size_t)
    noise_p = pack_latents(noise, patch_size, patch_size_t)
    noisy_p = pack_latents(noisy_latents, patch_size, patch_size_t)
    sig = sigmas.view(-1, 1, 1).expand(-1, *noisy_p.shape[1:-1], -1)
    timesteps = (sig * 1000.0).long()

    rope_interpolation_scale = [1 / (25 / 8), 32, 32]
    hidden = noisy_p.to(latents_p)
    if contiguous_hidden_states:
        hidden = hidden.contiguous()
    pred = transformer(
        hidden_states=hidden,
        encoder_hidden_states=encoder_hidden_states,
        encoder_attention_mask=encoder_attention_mask,
        num_frames=num_frames,
        height=height,
        width=width,
        timestep=timesteps,
        rope_interpolation_scale=rope_interpolation_scale,
        return_dict=False,
    )[0]
    target = flow_match_target(noise_p, latents_p)
    return pred, target, sig


# --------------------------------------------------------------------------------------
# sigma sampling, loss weights, loss, clip, AdamW (trainer level)
# --------------------------------------------------------------------------------------


def scheduler_sigmas(num_train_timesteps: int = 1000, shift: float = 1.0) -> torch.Tensor:
    """[upstream] ``FlowMatchEulerDiscreteScheduler().sigmas`` right after construction."""
    import numpy as np

    timesteps = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
    sig = torch.from_numpy(timesteps).to(dtype=torch.float32) / num_train_timesteps
    sig = shift * sig / (1 + (shift - 1) * sig)
    return sig


def compute_density_for_timestep_sampling(weighting_scheme, batch_size, logit_mean=None, logit_std=None,
                                          mode_scale=None, device=torch.device("cpu"), generator=None):
    """utils/diffusion.py:38-63."""
    if weighting_scheme == "logit_normal":
        u = torch.normal(mean=logit_mean, std=logit_std, size=(batch_size,), device=device, generator=generator)
        u = torch.nn.functional.sigmoid(u)
    elif weighting_scheme == "mode":
        u = torch.rand(size=(batch_size,), device=device, generator=generator)
        u = 1 - u - mode_scale * (torch.cos(math.pi * u / 2) ** 2 - 1 + u)
    else:
        u = torch.rand(size=(batch_size,), device=device, generator=generator)
    return u


def prepare_sigmas(sigmas_table, batch_size, num_train_timesteps=1000, flow_weighting_scheme="none",
                   flow_logit_mean=0.0, flow_logit_std=1.0, flow_mode_scale=1.29, generator=None):
    """utils/diffusion.py:84-114 (FlowMatch branch)."""
    weights = compute_density_for_timestep_sampling(
        flow_weighting_scheme, batch_size, flow_logit_mean, flow_logit_std, flow_mode_scale, generator=generator
    )
    indices = (weights * num_train_timesteps).long()
    return sigmas_table[indices]


def compute_loss_weighting_for_sd3(weighting_scheme: str, sigmas: torch.Tensor) -> torch.Tensor:
    """[upstream] diffusers.training_utils."""
    if weighting_scheme == "sigma_sqrt":
        return (sigmas**-2.0).float()
    if weighting_scheme == "cosmap":
        bot = 1 - 2 * sigmas + 2 * sigmas**2
        return 2 / (math.pi * bot)
    return torch.ones_like(sigmas)


def sft_loss(pred, target, sigmas_bs1, flow_weighting_scheme: str = "none", grad_accum: int = 1):
    """trainer.py:463-480."""
    weights = compute_loss_weighting_for_sd3(flow_weighting_scheme, sigmas_bs1)
    while weights.ndim < pred.ndim:
        weights = weights.unsqueeze(-1)
    loss = weights.float() * (pred.float() - target.float()).pow(2)
    loss = loss.mean(list(range(1, loss.ndim)))
    loss = loss.mean()
    if grad_accum > 1:
        loss = loss / grad_accum
    return loss


@torch.no_grad()
def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """utils/torch.py:99-161 + :299-374 (norm_type 2, local tensors)."""
    grads = [p.grad for p in parameters if p.grad is not None]
    if len(grads) == 0:
        return torch.tensor(0.0)
    norms = [torch.linalg.vector_norm(g, 2.0) for g in grads]
    total_norm = torch.linalg.vector_norm(torch.stack(norms), 2.0)
    clip_coef = max_norm / (total_norm + 1e-6)
    clip_coef_clamped = torch.clamp(clip_coef, max=1.0)
    for g in grads:
        g.mul_(clip_coef_clamped)
    return total_norm


@dataclass
class StepInputs:
    latents: torch.Tensor
    latents_mean: torch.Tensor
    latents_std: torch.Tensor
    encoder_hidden_states: torch.Tensor
    encoder_attention_mask: torch.Tensor
    sigmas: torch.Tensor  # [B] fp32
    noise: torch.Tensor
    first_frame_sigma: Optional[torch.Tensor] = None


def synth_inputs(cfg: LTXConfig, batch: int, frames: int, height: int, width: int, seed: int = 0,
                 mask_lens: Optional[List[int]] = None, sigmas: Optional[List[float]] = None,
                 dtype: torch.dtype = torch.bfloat16) -> StepInputs:
    """Synthetic inputs of SURVEY section 8d (latent-space sizes given directly)."""
    g = torch.Generator().manual_seed(seed)
    c = cfg.in_channels
    lat = torch.randn(batch, c, frames, height, width, generator=g).to(dtype)
    noise = torch.randn(batch, c, frames, height, width, generator=g).to(dtype)
    ehs = torch.randn(batch, cfg.text_seq_len, cfg.caption_channels, generator=g).to(dtype)
    if mask_lens is None:
        mask_lens = [32 + 64 * (i % 2) for i in range(batch)]
    mask = torch.zeros(batch, cfg.text_seq_len, dtype=dtype)
    for i, n in enumerate(mask_lens):
        mask[i, :n] = 1
    if sigmas is None:
        sigmas = [0.25 + 0.45 * (i % 2) for i in range(batch)]
    return StepInputs(
        latents=lat,
        latents_mean=torch.zeros(c),
        latents_std=torch.ones(c),
        encoder_hidden_states=ehs,
        encoder_attention_mask=mask,
        sigmas=torch.tensor(sigmas, dtype=torch.float32),
        noise=noise,
    )


def forward_loss(model, inp: StepInputs, flow_weighting_scheme: str = "none", contiguous_hidden_states: bool = False):
    sig5 = inp.sigmas.view(-1, 1, 1, 1, 1)
    ffs = None if inp.first_frame_sigma is None else inp.first_frame_sigma.view(-1, 1, 1, 1, 1)
    pred, target, sig = spec_forward(
        model, inp.latents.clone(), inp.latents_mean, inp.latents_std, inp.encoder_hidden_states,
        inp.encoder_attention_mask, sig5, noise=inp.noise, first_frame_sigma=ffs,
        contiguous_hidden_states=contiguous_hidden_states,
    )
    loss = sft_loss(pred, target, sig, flow_weighting_scheme)
    return loss, pred, target


def sft_step(model, optimizer, inp: StepInputs, max_grad_norm: float = 1.0, flow_weighting_scheme: str = "none",
             contiguous_hidden_states: bool = False):
    """One optimisation step in the reference's order (trainer.py:436-503).  Returns
    (loss, grad_norm, {name: grad-before-clip})."""
    loss, _, _ = forward_loss(model, inp, flow_weighting_scheme, contiguous_hidden_states)
    loss.backward()
    grads = {n: p.grad.detach().clone() for n, p in lora_parameters(model)}
    gn = clip_grad_norm_([p for p in model.parameters()], max_grad_norm)
    optimizer.step()
    optimizer.zero_grad()
    return loss.detach(), gn, grads


def make_optimizer(model, lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4):
    """optimizer.py:117-125 (torch.optim.AdamW, fused=False) over the trainable (LoRA) params."""
    params = [p for p in model.parameters() if p.requires_grad]
    return torch.optim.AdamW(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, fused=False)
