"""CPU oracle for the NEXT row after CogVideoX (BASELINE config 4): Wan-T2V fine-tuning -- TEST INFRASTRUCTURE ONLY.

Nothing under ``finetrainers_amd/`` may import this file; there are no Wan kernels yet.  It exists so that the Wan work starts the way LTX and
CogVideoX did: restatement first, pinned where the reference's own code can be executed.

What it restates, and how each part is pinned:
  * ``spec_forward``  -- ``WanModelSpecification.forward`` (finetrainers/models/wan/base_specification.py:433-493): stored posterior moments are
    normalised (mean AND log-variance, ``_normalize_latents`` :571-577, which MULTIPLIES by ``latents_std`` -- the processors store 1/std), sampled
    (``DiagonalGaussianDistribution``, finetrainers/models/utils.py:8-62), flow-match mixed (functional/diffusion.py:4-11).  PINNED: golden
    fixtures ``wan.spec.*`` are produced by executing the reference's own ``forward`` / ``_normalize_latents`` / ``DiagonalGaussianDistribution``.
  * ``WanTimeTextImageEmbedding.forward`` -- the reference's patch (finetrainers/patches/models/wan/patch.py:17-33).  PINNED: fixture ``wan.embed.*``
    executes the reference's patched function on this file's sub-modules.
  * ``WanTransformer3DModel`` (rotary table, blocks, head) -- [upstream] diffusers 0.33 ``transformer_wan.py``, absent from the reference tree and
    from this image: restated from the published algorithm, UNPINNED (parity unpinned for the block internals, as for LTX and CogVideoX), anchored
    on the reference's call sites and its dummy configuration (tests/models/wan/base_specification.py:38-52).
Full fine-tuning (config 4 trains every parameter under FSDP-2) needs no adapter code here: ``torch.autograd`` gives the reference gradients of all
parameters.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ltx import RMSNorm, TimestepEmbedding, get_timestep_embedding, native_sdpa


@dataclass
class WanConfig:
    """``WanTransformer3DModel`` hyper-parameters; defaults = Wan2.1-T2V-1.3B [upstream config.json]."""

    patch_size: Tuple[int, int, int] = (1, 2, 2)
    num_attention_heads: int = 12
    attention_head_dim: int = 128
    in_channels: int = 16
    out_channels: int = 16
    text_dim: int = 4096
    freq_dim: int = 256
    ffn_dim: int = 8960
    num_layers: int = 30
    cross_attn_norm: bool = True
    eps: float = 1e-6
    rope_max_seq_len: int = 1024

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def dummy() -> "WanConfig":
        """tests/models/wan/base_specification.py:38-52."""
        return WanConfig(num_attention_heads=2, attention_head_dim=12, text_dim=32, ffn_dim=32, num_layers=2, rope_max_seq_len=32)


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] rotary embedding: complex rotation of channel pairs, the head's channels split t : h : w
# --------------------------------------------------------------------------------------------------------------------------
def _complex_freqs(dim: int, length: int, theta: float = 10000.0) -> torch.Tensor:
    """get_1d_rotary_pos_embed(use_real=False, freqs_dtype=float64): exp(i * pos * theta^(-2k/dim)) [length, dim/2] complex128."""
    f = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
    ang = torch.outer(torch.arange(length, dtype=torch.float64), f)
    return torch.polar(torch.ones_like(ang), ang)


class WanRotaryPosEmbed(nn.Module):
    def __init__(self, cfg: WanConfig):
        super().__init__()
        self.cfg = cfg
        d = cfg.attention_head_dim
        h_dim = w_dim = 2 * (d // 6)
        t_dim = d - h_dim - w_dim
        self.freqs = torch.cat([_complex_freqs(n, cfg.rope_max_seq_len) for n in (t_dim, h_dim, w_dim)], dim=1)

    def forward(self, hidden_states: torch.Tensor) -> torch.Tensor:
        _, _, f, h, w = hidden_states.shape
        pt, ph, pw = self.cfg.patch_size
        ppf, pph, ppw = f // pt, h // ph, w // pw
        d = self.cfg.attention_head_dim
        ft, fh, fw = self.freqs.split_with_sizes([d // 2 - 2 * (d // 6), d // 6, d // 6], dim=1)
        ft = ft[:ppf].view(ppf, 1, 1, -1).expand(ppf, pph, ppw, -1)
        fh = fh[:pph].view(1, pph, 1, -1).expand(ppf, pph, ppw, -1)
        fw = fw[:ppw].view(1, 1, ppw, -1).expand(ppf, pph, ppw, -1)
        return torch.cat([ft, fh, fw], dim=-1).reshape(1, 1, ppf * pph * ppw, -1)


def apply_rotary_emb(x: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """x [B, H, S, d]: pairs (2k, 2k+1) as complex numbers (float64) times freqs, back to x's dtype."""
    xc = torch.view_as_complex(x.to(torch.float64).unflatten(3, (-1, 2)))
    return torch.view_as_real(xc * freqs).flatten(3, 4).type_as(x)


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] modules
# --------------------------------------------------------------------------------------------------------------------------
class FP32LayerNorm(nn.LayerNorm):
    def forward(self, x):
        return F.layer_norm(x.float(), self.normalized_shape, None if self.weight is None else self.weight.float(),
                            None if self.bias is None else self.bias.float(), self.eps).to(x.dtype)


class TextProjection(nn.Module):
    """PixArtAlphaTextProjection(text_dim, dim, act_fn="gelu_tanh")."""

    def __init__(self, in_features: int, hidden: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden)
        self.linear_2 = nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(F.gelu(self.linear_1(x), approximate="tanh"))


class _Timesteps(nn.Module):
    def __init__(self, num_channels: int):
        super().__init__()
        self.num_channels = num_channels

    def forward(self, t):
        return get_timestep_embedding(t, self.num_channels)  # flip_sin_to_cos=True, downscale_freq_shift=0


class WanTimeTextImageEmbedding(nn.Module):
    def __init__(self, cfg: WanConfig):
        super().__init__()
        d = cfg.inner_dim
        self.timesteps_proj = _Timesteps(cfg.freq_dim)
        self.time_embedder = TimestepEmbedding(cfg.freq_dim, d)
        self.act_fn = nn.SiLU()
        self.time_proj = nn.Linear(d, d * 6)
        self.text_embedder = TextProjection(cfg.text_dim, d)
        self.image_embedder = None

    def forward(self, timestep, encoder_hidden_states, encoder_hidden_states_image=None):
        """finetrainers/patches/models/wan/patch.py:17-33 (the reference's patched forward: the timestep projection takes the text dtype)."""
        timestep = self.timesteps_proj(timestep).type_as(encoder_hidden_states)
        temb = self.time_embedder(timestep).type_as(encoder_hidden_states)
        timestep_proj = self.time_proj(self.act_fn(temb))
        encoder_hidden_states = self.text_embedder(encoder_hidden_states)
        if encoder_hidden_states_image is not None:
            encoder_hidden_states_image = self.image_embedder(encoder_hidden_states_image)
        return temb, timestep_proj, encoder_hidden_states, encoder_hidden_states_image


class WanAttention(nn.Module):
    """diffusers ``Attention(qk_norm="rms_norm_across_heads", bias=True, out_bias=True)`` + ``WanAttnProcessor2_0`` (T2V: no image branch):
    q / k RMSNorm over the WHOLE inner dimension (affine), rotary embedding on self-attention only."""

    def __init__(self, cfg: WanConfig):
        super().__init__()
        d = cfg.inner_dim
        self.heads, self.dim_head = cfg.num_attention_heads, cfg.attention_head_dim
        self.to_q, self.to_k, self.to_v = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.to_out = nn.ModuleList([nn.Linear(d, d), nn.Dropout(0.0)])
        self.norm_q = RMSNorm(d, cfg.eps, elementwise_affine=True)
        self.norm_k = RMSNorm(d, cfg.eps, elementwise_affine=True)

    def forward(self, hidden_states, encoder_hidden_states=None, rotary_emb=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        q, k, v = self.norm_q(self.to_q(hidden_states)), self.norm_k(self.to_k(ctx)), self.to_v(ctx)
        split = lambda t: t.unflatten(2, (self.heads, -1)).transpose(1, 2)
        q, k, v = split(q), split(k), split(v)
        if rotary_emb is not None:
            q, k = apply_rotary_emb(q, rotary_emb), apply_rotary_emb(k, rotary_emb)
        o = native_sdpa(q, k, v, None).transpose(1, 2).flatten(2, 3).type_as(q)
        return self.to_out[1](self.to_out[0](o))


class WanFeedForward(nn.Module):
    """FeedForward(dim, inner_dim=ffn_dim, activation_fn="gelu-approximate")."""

    def __init__(self, dim: int, ffn_dim: int):
        super().__init__()
        self.proj_in = nn.Linear(dim, ffn_dim)   # net.0.proj
        self.proj_out = nn.Linear(ffn_dim, dim)  # net.2

    def forward(self, x):
        return self.proj_out(F.gelu(self.proj_in(x), approximate="tanh"))


class WanTransformerBlock(nn.Module):
    def __init__(self, cfg: WanConfig):
        super().__init__()
        d = cfg.inner_dim
        self.norm1 = FP32LayerNorm(d, cfg.eps, elementwise_affine=False)
        self.attn1 = WanAttention(cfg)
        self.attn2 = WanAttention(cfg)
        self.norm2 = FP32LayerNorm(d, cfg.eps, elementwise_affine=True) if cfg.cross_attn_norm else nn.Identity()
        self.ffn = WanFeedForward(d, cfg.ffn_dim)
        self.norm3 = FP32LayerNorm(d, cfg.eps, elementwise_affine=False)
        self.scale_shift_table = nn.Parameter(torch.randn(1, 6, d) / d**0.5)

    def forward(self, hidden_states, encoder_hidden_states, temb, rotary_emb):
        shift_msa, scale_msa, gate_msa, c_shift, c_scale, c_gate = (self.scale_shift_table + temb.float()).chunk(6, dim=1)
        n = (self.norm1(hidden_states.float()) * (1 + scale_msa) + shift_msa).type_as(hidden_states)
        a = self.attn1(n, rotary_emb=rotary_emb)
        hidden_states = (hidden_states.float() + a * gate_msa).type_as(hidden_states)
        n = self.norm2(hidden_states.float()).type_as(hidden_states)
        hidden_states = hidden_states + self.attn2(n, encoder_hidden_states=encoder_hidden_states)
        n = (self.norm3(hidden_states.float()) * (1 + c_scale) + c_shift).type_as(hidden_states)
        f = self.ffn(n)
        return (hidden_states.float() + f.float() * c_gate).type_as(hidden_states)


class WanTransformer3DModel(nn.Module):
    def __init__(self, cfg: WanConfig):
        super().__init__()
        self.cfg = self.config = cfg
        d = cfg.inner_dim
        self.rope = WanRotaryPosEmbed(cfg)
        self.patch_embedding = nn.Conv3d(cfg.in_channels, d, kernel_size=cfg.patch_size, stride=cfg.patch_size)
        self.condition_embedder = WanTimeTextImageEmbedding(cfg)
        self.blocks = nn.ModuleList([WanTransformerBlock(cfg) for _ in range(cfg.num_layers)])
        self.norm_out = FP32LayerNorm(d, cfg.eps, elementwise_affine=False)
        self.proj_out = nn.Linear(d, cfg.out_channels * math.prod(cfg.patch_size))
        self.scale_shift_table = nn.Parameter(torch.randn(1, 2, d) / d**0.5)

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None, return_dict: bool = True, **kwargs):
        b, _, f, h, w = hidden_states.shape
        pt, ph, pw = self.cfg.patch_size
        ppf, pph, ppw = f // pt, h // ph, w // pw
        rotary = self.rope(hidden_states)
        x = self.patch_embedding(hidden_states).flatten(2).transpose(1, 2)
        temb, tproj, enc, _ = self.condition_embedder(timestep, encoder_hidden_states, encoder_hidden_states_image)
        tproj = tproj.unflatten(1, (6, -1))
        for blk in self.blocks:
            x = blk(x, enc, tproj, rotary)
        shift, scale = (self.scale_shift_table + temb.unsqueeze(1)).chunk(2, dim=1)
        x = (self.norm_out(x.float()) * (1 + scale) + shift).type_as(x)
        x = self.proj_out(x)
        x = x.reshape(b, ppf, pph, ppw, pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6)
        out = x.flatten(6, 7).flatten(4, 5).flatten(2, 3)
        return (out,) if not return_dict else {"sample": out}


def build_model(cfg: WanConfig, seed: int = 0, dtype: torch.dtype = torch.bfloat16) -> WanTransformer3DModel:
    torch.manual_seed(seed)
    return WanTransformer3DModel(cfg).to(dtype)


# --------------------------------------------------------------------------------------------------------------------------
# specification forward + loss (the reference's own code paths, pinned)
# --------------------------------------------------------------------------------------------------------------------------
def normalize_latents(latents, latents_mean, latents_std):
    """base_specification.py:571-577 (note: MULTIPLIES by latents_std; the Wan processors hand over 1 / std)."""
    mean = latents_mean.view(1, -1, 1, 1, 1).to(latents.device)
    std = latents_std.view(1, -1, 1, 1, 1).to(latents.device)
    return ((latents.float() - mean) * std).to(latents)


def posterior_sample(moments: torch.Tensor, eps: torch.Tensor) -> torch.Tensor:
    """models/utils.py:8-31: mean + exp(0.5 * clamp(logvar, -30, 20)) * eps along dim 1."""
    mean, logvar = torch.chunk(moments, 2, dim=1)
    return mean + torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0)) * eps.to(moments.dtype)


def spec_forward(transformer, moments: torch.Tensor, latents_mean: torch.Tensor, latents_std: torch.Tensor, encoder_hidden_states: torch.Tensor,
                 sigmas: torch.Tensor, eps: torch.Tensor, noise: torch.Tensor):
    """base_specification.py:433-493, T2V (``image_dim`` None).  The reference forces ``compute_posterior = False`` (:446): ``moments`` are the
    stored VAE moments [B, 2C, F, H, W]; mean and log-variance are BOTH normalised, then sampled with ``eps``; ``noise`` is the flow-match noise.
    ``sigmas`` must already be expanded to the latents' rank ([B, 1, 1, 1, 1])."""
    mu, logvar = torch.chunk(moments, 2, dim=1)
    mu, logvar = normalize_latents(mu, latents_mean, latents_std), normalize_latents(logvar, latents_mean, latents_std)
    latents = posterior_sample(torch.cat([mu, logvar], dim=1), eps)
    noisy = (1.0 - sigmas) * latents + sigmas * noise  # functional/diffusion.py:4-6
    timesteps = (sigmas.flatten() * 1000.0).long()
    pred = transformer(hidden_states=noisy.to(latents), encoder_hidden_states=encoder_hidden_states, timestep=timesteps, return_dict=False)[0]
    return pred, noise - latents, sigmas  # functional/diffusion.py:9-11


def sft_loss(pred, target, sigmas):
    """trainer.py:463-481 with flow_weighting_scheme "none" (weights = 1)."""
    loss = (pred.float() - target.float()).pow(2)
    return loss.mean(list(range(1, loss.ndim))).mean()
