"""CPU restatement (oracle) of the reference CogVideoX LoRA SFT step -- SURVEY section 8(f)-1 / BASELINE config 3.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  Pure PyTorch on CPU, the torch op sequence the reference executes.

What each piece follows (paths relative to /root/reference):

* spec-level forward ......... finetrainers/models/cogvideox/base_specification.py:258-333 (scaling, frame padding :403-410, DDIM
                               ``add_noise``, RoPE table choice, transformer call, ``get_velocity`` -> x0 prediction, target = latents)
* RoPE table construction .... finetrainers/models/cogvideox/utils.py:8-51
* sigma sampling ............. finetrainers/utils/diffusion.py:84-114 (CogVideoXDDIMScheduler branch: uniform)
* loss weight 1/(1-alpha) .... finetrainers/utils/diffusion.py:117-130, trainer/sft_trainer/trainer.py:463-481
* attention .................. finetrainers/models/attention_dispatch.py:938-962 (torch SDPA)
* dummy fixture config ....... tests/models/cogvideox/base_specification.py:49-64

[upstream] parts (diffusers 0.32/0.33, NOT in /root/reference; restated from the published algorithm, **parity unpinned**):
``CogVideoXTransformer3DModel`` (``models/transformers/cogvideox_transformer_3d.py``), ``CogVideoXBlock``,
``CogVideoXLayerNormZero`` / ``AdaLayerNorm`` (``models/normalization.py``), ``CogVideoXPatchEmbed`` + sincos / rotary embeddings
(``models/embeddings.py``), ``CogVideoXAttnProcessor2_0`` (``models/attention_processor.py``), ``CogVideoXDDIMScheduler``
(``schedulers/scheduling_ddim_cogvideox.py``), ``get_resize_crop_region_for_grid`` (``pipelines/cogvideo/pipeline_cogvideox.py``),
peft LoRA (shared with oracle/ltx.py).
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ltx import LoraLinear, TimestepEmbedding, get_timestep_embedding, native_sdpa


@dataclass
class CogVideoXConfig:
    """``CogVideoXTransformer3DModel`` hyper-parameters; defaults = CogVideoX-2b (BASELINE config 3) [upstream config.json]."""

    num_attention_heads: int = 30
    attention_head_dim: int = 64
    in_channels: int = 16
    out_channels: int = 16
    time_embed_dim: int = 512
    text_embed_dim: int = 4096
    num_layers: int = 30
    sample_width: int = 90
    sample_height: int = 60
    sample_frames: int = 49
    patch_size: int = 2
    patch_size_t: Optional[int] = None
    temporal_compression_ratio: int = 4
    max_text_seq_length: int = 226
    norm_eps: float = 1e-5
    spatial_interpolation_scale: float = 1.875
    temporal_interpolation_scale: float = 1.0
    use_rotary_positional_embeddings: bool = False  # 2b: sincos table added in the patch embed; 5b: rotary
    ofs_embed_dim: Optional[int] = None      # CogVideoX 1.5: 512 (an extra "offset" embedding added to the time embedding)
    patch_bias: bool = True                  # CogVideoX 1.5: False
    ff_mult: int = 4

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def dummy() -> "CogVideoXConfig":
        """tests/models/cogvideox/base_specification.py:49-64."""
        return CogVideoXConfig(num_attention_heads=4, attention_head_dim=16, in_channels=4, out_channels=4, time_embed_dim=2, text_embed_dim=32,
                               num_layers=2, sample_width=24, sample_height=24, sample_frames=9, patch_size=2, temporal_compression_ratio=4,
                               max_text_seq_length=16, use_rotary_positional_embeddings=True)


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] positional embeddings
# --------------------------------------------------------------------------------------------------------------------------
def _sincos_1d(embed_dim: int, pos: torch.Tensor) -> torch.Tensor:
    """get_1d_sincos_pos_embed_from_grid: omega_i = 1 / 10000^(i / (D/2)); out = [sin(pos * omega) | cos(pos * omega)] (float64 math)."""
    omega = torch.arange(embed_dim // 2, dtype=torch.float64) / (embed_dim / 2.0)
    omega = 1.0 / 10000**omega
    out = torch.outer(pos.reshape(-1).to(torch.float64), omega)
    return torch.cat([torch.sin(out), torch.cos(out)], dim=1)


def get_3d_sincos_pos_embed(embed_dim: int, spatial_size: Tuple[int, int], temporal_size: int, spatial_interpolation_scale: float = 1.0,
                            temporal_interpolation_scale: float = 1.0) -> torch.Tensor:
    """[upstream] -> [T, H*W, D]: first D/4 channels temporal, last 3D/4 spatial (2-D sincos, w-half then h-half)."""
    d_sp, d_t = 3 * embed_dim // 4, embed_dim // 4
    grid_h = torch.arange(spatial_size[1], dtype=torch.float32) / spatial_interpolation_scale
    grid_w = torch.arange(spatial_size[0], dtype=torch.float32) / spatial_interpolation_scale
    gw, gh = torch.meshgrid(grid_w, grid_h, indexing="xy")  # width goes first
    emb_h = _sincos_1d(d_sp // 2, gw)
    emb_w = _sincos_1d(d_sp // 2, gh)
    pos_sp = torch.cat([emb_h, emb_w], dim=1)  # [H*W, d_sp]
    grid_t = torch.arange(temporal_size, dtype=torch.float32) / temporal_interpolation_scale
    pos_t = _sincos_1d(d_t, grid_t)  # [T, d_t]
    pos_sp = pos_sp[None].repeat_interleave(temporal_size, dim=0)
    pos_t = pos_t[:, None].repeat_interleave(spatial_size[0] * spatial_size[1], dim=1)
    return torch.cat([pos_t, pos_sp], dim=-1).float()


def get_resize_crop_region_for_grid(src, tgt_width, tgt_height):
    """[upstream] pipelines/cogvideo/pipeline_cogvideox.py."""
    tw, th = tgt_width, tgt_height
    h, w = src
    r = h / w
    if r > (th / tw):
        resize_height = th
        resize_width = int(round(th / h * w))
    else:
        resize_width = tw
        resize_height = int(round(tw / w * h))
    crop_top = int(round((th - resize_height) / 2.0))
    crop_left = int(round((tw - resize_width) / 2.0))
    return (crop_top, crop_left), (crop_top + resize_height, crop_left + resize_width)


def _rotary_1d(dim: int, pos, theta: float = 10000.0) -> Tuple[torch.Tensor, torch.Tensor]:
    """get_1d_rotary_pos_embed(use_real=True, repeat_interleave_real=True): cos / sin [S, dim], every frequency repeated twice."""
    if isinstance(pos, int):
        pos = torch.arange(pos)
    pos = torch.as_tensor(pos, dtype=torch.float32)
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
    freqs = torch.outer(pos, freqs)
    return freqs.cos().repeat_interleave(2, dim=1).float(), freqs.sin().repeat_interleave(2, dim=1).float()


def get_3d_rotary_pos_embed(embed_dim, crops_coords, grid_size, temporal_size, theta: float = 10000.0, grid_type: str = "linspace",
                            max_size: Optional[Tuple[int, int]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """[upstream] -> (cos, sin) [T*H*W, embed_dim]; channel split t : h : w = D/4 : 3D/8 : 3D/8."""
    if grid_type == "linspace":
        start, stop = crops_coords
        gh, gw = grid_size
        grid_h = torch.linspace(start[0], stop[0] * (gh - 1) / gh, gh, dtype=torch.float32)
        grid_w = torch.linspace(start[1], stop[1] * (gw - 1) / gw, gw, dtype=torch.float32)
        grid_t = torch.linspace(0, temporal_size * (temporal_size - 1) / temporal_size, temporal_size, dtype=torch.float32)
    elif grid_type == "slice":
        mh, mw = max_size
        gh, gw = grid_size
        grid_h, grid_w, grid_t = torch.arange(mh, dtype=torch.float32), torch.arange(mw, dtype=torch.float32), torch.arange(temporal_size, dtype=torch.float32)
    else:
        raise ValueError(grid_type)
    dim_t, dim_h, dim_w = embed_dim // 4, embed_dim // 8 * 3, embed_dim // 8 * 3
    ft, fh, fw = _rotary_1d(dim_t, grid_t, theta), _rotary_1d(dim_h, grid_h, theta), _rotary_1d(dim_w, grid_w, theta)

    def combine(t, h, w):
        t = t[:, None, None, :].expand(-1, gh, gw, -1)
        h = h[None, :, None, :].expand(temporal_size, -1, gw, -1)
        w = w[None, None, :, :].expand(temporal_size, gh, -1, -1)
        return torch.cat([t, h, w], dim=-1)

    if grid_type == "slice":
        fh = (fh[0][:gh], fh[1][:gh])
        fw = (fw[0][:gw], fw[1][:gw])
    cos = combine(ft[0], fh[0], fw[0]).reshape(temporal_size * gh * gw, -1)
    sin = combine(ft[1], fh[1], fw[1]).reshape(temporal_size * gh * gw, -1)
    return cos, sin


def prepare_rotary_positional_embeddings(height, width, num_frames, vae_scale_factor_spatial=8, patch_size=2, patch_size_t=None,
                                         attention_head_dim=64, base_height=480, base_width=720):
    """finetrainers/models/cogvideox/utils.py:8-51 (restated; the golden fixtures execute the reference's own function)."""
    grid_height = height // (vae_scale_factor_spatial * patch_size)
    grid_width = width // (vae_scale_factor_spatial * patch_size)
    base_size_width = base_width // (vae_scale_factor_spatial * patch_size)
    base_size_height = base_height // (vae_scale_factor_spatial * patch_size)
    if patch_size_t is None:
        crops = get_resize_crop_region_for_grid((grid_height, grid_width), base_size_width, base_size_height)
        return get_3d_rotary_pos_embed(attention_head_dim, crops, (grid_height, grid_width), num_frames)
    base_num_frames = (num_frames + patch_size_t - 1) // patch_size_t
    return get_3d_rotary_pos_embed(attention_head_dim, None, (grid_height, grid_width), base_num_frames, grid_type="slice",
                                   max_size=(base_size_height, base_size_width))


def apply_rotary_emb(x: torch.Tensor, freqs: Tuple[torch.Tensor, torch.Tensor]) -> torch.Tensor:
    """[upstream] embeddings.apply_rotary_emb(use_real=True, use_real_unbind_dim=-1); x [B, H, S, d], cos / sin [S, d]."""
    cos, sin = freqs
    cos, sin = cos[None, None], sin[None, None]
    x_real, x_imag = x.reshape(*x.shape[:-1], -1, 2).unbind(-1)
    x_rotated = torch.stack([-x_imag, x_real], dim=-1).flatten(3)
    return (x.float() * cos + x_rotated.float() * sin).to(x.dtype)


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] modules
# --------------------------------------------------------------------------------------------------------------------------
class CogVideoXPatchEmbed(nn.Module):
    def __init__(self, cfg: CogVideoXConfig):
        super().__init__()
        self.cfg = cfg
        d = cfg.inner_dim
        if cfg.patch_size_t is None:  # CogVideoX 1.0: a 2-D patch convolution per frame
            self.proj = nn.Conv2d(cfg.in_channels, d, kernel_size=(cfg.patch_size, cfg.patch_size), stride=cfg.patch_size, bias=cfg.patch_bias)
        else:  # CogVideoX 1.5: patches over (patch_size_t frames) x p x p, a Linear over the flattened patch (channel, frame, row, column order)
            self.proj = nn.Linear(cfg.in_channels * cfg.patch_size * cfg.patch_size * cfg.patch_size_t, d, bias=cfg.patch_bias)
        self.text_proj = nn.Linear(cfg.text_embed_dim, d)
        self.use_positional_embeddings = not cfg.use_rotary_positional_embeddings
        if self.use_positional_embeddings:
            self.register_buffer("pos_embedding", self._pos(cfg.sample_height, cfg.sample_width, cfg.sample_frames), persistent=False)

    def _pos(self, sample_height, sample_width, sample_frames) -> torch.Tensor:
        c = self.cfg
        ph, pw = sample_height // c.patch_size, sample_width // c.patch_size
        pt = (sample_frames - 1) // c.temporal_compression_ratio + 1
        pe = get_3d_sincos_pos_embed(c.inner_dim, (pw, ph), pt, c.spatial_interpolation_scale, c.temporal_interpolation_scale).flatten(0, 1)
        joint = torch.zeros(1, c.max_text_seq_length + ph * pw * pt, c.inner_dim)
        joint[0, c.max_text_seq_length:] = pe
        return joint

    def forward(self, text_embeds: torch.Tensor, image_embeds: torch.Tensor) -> torch.Tensor:
        text_embeds = self.text_proj(text_embeds)
        b, f, ch, h, w = image_embeds.shape
        if self.cfg.patch_size_t is None:
            x = self.proj(image_embeds.reshape(-1, ch, h, w))
            x = x.view(b, f, *x.shape[1:]).flatten(3).transpose(2, 3).flatten(1, 2)  # [B, F*h*w, D]
        else:
            p, pt = self.cfg.patch_size, self.cfg.patch_size_t
            x = image_embeds.permute(0, 1, 3, 4, 2)  # [B, F, H, W, C]
            x = x.reshape(b, f // pt, pt, h // p, p, w // p, p, ch)
            x = x.permute(0, 1, 3, 5, 7, 2, 4, 6).flatten(4, 7).flatten(1, 3)  # [B, (F/pt)(H/p)(W/p), C pt p p]
            x = self.proj(x)
        embeds = torch.cat([text_embeds, x], dim=1).contiguous()
        if self.use_positional_embeddings:
            c = self.cfg
            pre_f = (f - 1) * c.temporal_compression_ratio + 1
            if (h, w, pre_f) != (c.sample_height, c.sample_width, c.sample_frames):
                pos = self._pos(h, w, pre_f)
            else:
                pos = self.pos_embedding
            embeds = embeds + pos.to(device=embeds.device, dtype=embeds.dtype)
        return embeds


class CogVideoXLayerNormZero(nn.Module):
    def __init__(self, conditioning_dim: int, dim: int, eps: float):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 6 * dim, bias=True)
        self.norm = nn.LayerNorm(dim, eps=eps, elementwise_affine=True)

    def forward(self, hidden_states, encoder_hidden_states, temb):
        shift, scale, gate, enc_shift, enc_scale, enc_gate = self.linear(self.silu(temb)).chunk(6, dim=1)
        hidden_states = self.norm(hidden_states) * (1 + scale)[:, None, :] + shift[:, None, :]
        encoder_hidden_states = self.norm(encoder_hidden_states) * (1 + enc_scale)[:, None, :] + enc_shift[:, None, :]
        return hidden_states, encoder_hidden_states, gate[:, None, :], enc_gate[:, None, :]


class AdaLayerNorm(nn.Module):
    """``AdaLayerNorm(embedding_dim=time_embed_dim, output_dim=2 * dim, chunk_dim=1)`` (norm_out)."""

    def __init__(self, conditioning_dim: int, dim: int, eps: float):
        super().__init__()
        self.silu = nn.SiLU()
        self.linear = nn.Linear(conditioning_dim, 2 * dim)
        self.norm = nn.LayerNorm(dim, eps, True)

    def forward(self, x, temb):
        temb = self.linear(self.silu(temb))
        shift, scale = temb.chunk(2, dim=1)
        return self.norm(x) * (1 + scale[:, None, :]) + shift[:, None, :]


class JointAttention(nn.Module):
    """diffusers ``Attention(qk_norm="layer_norm", eps=1e-6, bias=True, out_bias=True)`` + ``CogVideoXAttnProcessor2_0``: text and video
    tokens attend jointly, q / k LayerNorm per head, RoPE on the video part only."""

    def __init__(self, cfg: CogVideoXConfig):
        super().__init__()
        d = cfg.inner_dim
        self.heads, self.dim_head = cfg.num_attention_heads, cfg.attention_head_dim
        self.norm_q = nn.LayerNorm(cfg.attention_head_dim, eps=1e-6, elementwise_affine=True)
        self.norm_k = nn.LayerNorm(cfg.attention_head_dim, eps=1e-6, elementwise_affine=True)
        self.to_q, self.to_k, self.to_v = nn.Linear(d, d, bias=True), nn.Linear(d, d, bias=True), nn.Linear(d, d, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(d, d, bias=True), nn.Dropout(0.0)])

    def forward(self, hidden_states, encoder_hidden_states, image_rotary_emb=None):
        text_len = encoder_hidden_states.size(1)
        x = torch.cat([encoder_hidden_states, hidden_states], dim=1)
        b, s, _ = x.shape
        split = lambda t: t.view(b, s, self.heads, self.dim_head).transpose(1, 2)
        q, k, v = split(self.to_q(x)), split(self.to_k(x)), split(self.to_v(x))
        q, k = self.norm_q(q), self.norm_k(k)
        if image_rotary_emb is not None:
            q = torch.cat([q[:, :, :text_len], apply_rotary_emb(q[:, :, text_len:], image_rotary_emb)], dim=2)
            k = torch.cat([k[:, :, :text_len], apply_rotary_emb(k[:, :, text_len:], image_rotary_emb)], dim=2)
        o = native_sdpa(q, k, v, None)
        o = o.transpose(1, 2).reshape(b, s, self.heads * self.dim_head)
        o = self.to_out[1](self.to_out[0](o))
        return o[:, text_len:], o[:, :text_len]


class FeedForward(nn.Module):
    def __init__(self, dim: int, mult: int):
        super().__init__()
        self.proj_in = nn.Linear(dim, dim * mult, bias=True)   # net.0.proj (GELU tanh)
        self.proj_out = nn.Linear(dim * mult, dim, bias=True)  # net.2  (final_dropout = identity at p = 0)

    def forward(self, x):
        return self.proj_out(F.gelu(self.proj_in(x), approximate="tanh"))


class CogVideoXBlock(nn.Module):
    def __init__(self, cfg: CogVideoXConfig):
        super().__init__()
        d = cfg.inner_dim
        self.norm1 = CogVideoXLayerNormZero(cfg.time_embed_dim, d, cfg.norm_eps)
        self.attn1 = JointAttention(cfg)
        self.norm2 = CogVideoXLayerNormZero(cfg.time_embed_dim, d, cfg.norm_eps)
        self.ff = FeedForward(d, cfg.ff_mult)

    def forward(self, hidden_states, encoder_hidden_states, temb, image_rotary_emb=None):
        text_len = encoder_hidden_states.size(1)
        nh, ne, gate_msa, enc_gate_msa = self.norm1(hidden_states, encoder_hidden_states, temb)
        ah, ae = self.attn1(nh, ne, image_rotary_emb)
        hidden_states = hidden_states + gate_msa * ah
        encoder_hidden_states = encoder_hidden_states + enc_gate_msa * ae
        nh, ne, gate_ff, enc_gate_ff = self.norm2(hidden_states, encoder_hidden_states, temb)
        ff = self.ff(torch.cat([ne, nh], dim=1))
        hidden_states = hidden_states + gate_ff * ff[:, text_len:]
        encoder_hidden_states = encoder_hidden_states + enc_gate_ff * ff[:, :text_len]
        return hidden_states, encoder_hidden_states


class CogVideoXTransformer3DModel(nn.Module):
    def __init__(self, cfg: CogVideoXConfig):
        super().__init__()
        self.cfg = self.config = cfg
        d = cfg.inner_dim
        self.patch_embed = CogVideoXPatchEmbed(cfg)
        self.time_embedding = TimestepEmbedding(d, cfg.time_embed_dim)
        self.transformer_blocks = nn.ModuleList([CogVideoXBlock(cfg) for _ in range(cfg.num_layers)])
        self.norm_final = nn.LayerNorm(d, cfg.norm_eps, True)
        self.norm_out = AdaLayerNorm(cfg.time_embed_dim, d, cfg.norm_eps)
        self.proj_out = nn.Linear(d, cfg.patch_size * cfg.patch_size * cfg.out_channels * (cfg.patch_size_t or 1))
        # CogVideoX 1.5: Timesteps(ofs_embed_dim, flip_sin_to_cos=True, freq_shift=0) -> TimestepEmbedding(ofs_embed_dim, ofs_embed_dim), added to the time embedding
        self.ofs_embedding = TimestepEmbedding(cfg.ofs_embed_dim, cfg.ofs_embed_dim) if cfg.ofs_embed_dim is not None else None

    @property
    def device(self):
        return self.proj_out.weight.device

    def forward(self, hidden_states, encoder_hidden_states, timestep, image_rotary_emb=None, ofs=None, return_dict: bool = True, **kwargs):
        b, f, ch, h, w = hidden_states.shape
        t_emb = get_timestep_embedding(timestep, self.cfg.inner_dim).to(dtype=hidden_states.dtype)  # flip_sin_to_cos=True, freq_shift=0
        emb = self.time_embedding(t_emb)
        if self.ofs_embedding is not None:
            ofs_emb = get_timestep_embedding(ofs, self.cfg.ofs_embed_dim).to(dtype=hidden_states.dtype)
            emb = emb + self.ofs_embedding(ofs_emb)
        x = self.patch_embed(encoder_hidden_states, hidden_states)
        text_len = encoder_hidden_states.shape[1]
        enc, x = x[:, :text_len], x[:, text_len:]
        for blk in self.transformer_blocks:
            x, enc = blk(x, enc, emb, image_rotary_emb)
        if not self.cfg.use_rotary_positional_embeddings:
            x = self.norm_final(x)
        else:
            x = self.norm_final(torch.cat([enc, x], dim=1))[:, text_len:]
        x = self.proj_out(self.norm_out(x, temb=emb))
        p, pt = self.cfg.patch_size, self.cfg.patch_size_t
        if pt is None:
            out = x.reshape(b, f, h // p, w // p, -1, p, p).permute(0, 1, 4, 2, 5, 3, 6).flatten(5, 6).flatten(3, 4)
        else:
            out = x.reshape(b, (f + pt - 1) // pt, h // p, w // p, -1, pt, p, p).permute(0, 1, 5, 4, 2, 6, 3, 7).flatten(6, 7).flatten(4, 5).flatten(1, 2)
        return (out,) if not return_dict else {"sample": out}


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] scheduler
# --------------------------------------------------------------------------------------------------------------------------
def _rescale_zero_terminal_snr(alphas_cumprod: torch.Tensor) -> torch.Tensor:
    a = alphas_cumprod.sqrt()
    a0, aT = a[0].clone(), a[-1].clone()
    a = (a - aT) * (a0 / (a0 - aT))
    return a**2


class CogVideoXDDIMScheduler:
    """Constructor defaults of ``CogVideoXDDIMScheduler()`` (what the reference's dummy spec builds; the 2b checkpoint's scheduler config adds
    ``snr_shift_scale=3.0``, ``rescale_betas_zero_snr=True`` -- pass them for the production oracle)."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.0120, snr_shift_scale: float = 3.0,
                 rescale_betas_zero_snr: bool = False):
        self.config = type("Cfg", (), {"num_train_timesteps": num_train_timesteps})()
        betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float64) ** 2  # "scaled_linear"
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:
            ac = _rescale_zero_terminal_snr(ac)
        self.alphas_cumprod = ac.float()

    def add_noise(self, original, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original.device, dtype=original.dtype)
        sa = (ac[timesteps] ** 0.5).flatten()
        so = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < original.ndim:
            sa, so = sa.unsqueeze(-1), so.unsqueeze(-1)
        return sa * original + so * noise

    def get_velocity(self, sample, noise, timesteps):
        ac = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        sa = (ac[timesteps] ** 0.5).flatten()
        so = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sa.ndim < sample.ndim:
            sa, so = sa.unsqueeze(-1), so.unsqueeze(-1)
        return sa * noise - so * sample


# --------------------------------------------------------------------------------------------------------------------------
# LoRA, spec forward, loss
# --------------------------------------------------------------------------------------------------------------------------
def add_lora(model: CogVideoXTransformer3DModel, rank: int = 64, alpha: float = 64.0) -> List[str]:
    """Default target regex (sft_trainer/config.py:24-26) on CogVideoX: to_q / to_k / to_v / to_out.0 of every block's attn1."""
    for p in model.parameters():
        p.requires_grad_(False)
    names = []
    for li, blk in enumerate(model.transformer_blocks):
        for t in ("to_q", "to_k", "to_v"):
            setattr(blk.attn1, t, LoraLinear(getattr(blk.attn1, t), rank, alpha))
            names.append(f"transformer_blocks.{li}.attn1.{t}")
        blk.attn1.to_out[0] = LoraLinear(blk.attn1.to_out[0], rank, alpha)
        names.append(f"transformer_blocks.{li}.attn1.to_out.0")
    for n, p in model.named_parameters():
        if "lora_" in n:
            p.data = p.data.float()
            p.requires_grad_(True)
    return names


def build_model(cfg: CogVideoXConfig, seed: int = 0, dtype: torch.dtype = torch.bfloat16, rank: int = 64, alpha: float = 64.0,
                lora_b_std: Optional[float] = None) -> CogVideoXTransformer3DModel:
    torch.manual_seed(seed)
    model = CogVideoXTransformer3DModel(cfg)
    model.to(dtype)
    if rank > 0:
        add_lora(model, rank, alpha)
        if lora_b_std is not None:
            g = torch.Generator().manual_seed(seed + 1)
            for n, p in model.named_parameters():
                if "lora_B" in n:
                    p.data = torch.randn(p.shape, generator=g) * lora_b_std
    return model


def pad_frames(latents: torch.Tensor, patch_size_t: int) -> torch.Tensor:
    """base_specification.py:403-410 (note: pads a FULL group when the frame count already divides -- reproduced as written)."""
    additional = patch_size_t - (latents.size(1) % patch_size_t)
    if additional > 0:
        latents = torch.cat([latents, latents[:, -1:].expand(-1, additional, -1, -1, -1)], dim=1)
    return latents


def spec_forward(transformer, scheduler: CogVideoXDDIMScheduler, latents: torch.Tensor, encoder_hidden_states: torch.Tensor, sigmas: torch.Tensor,
                 noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None, scaling_factor: float = 1.15258426,
                 invert_scale_latents: bool = False):
    """base_specification.py:258-333 with ``compute_posterior=True`` (latents [B, F, C, H, W] already sampled); noise injected for parity."""
    cfg = transformer.cfg
    vae_sf = 8
    rope_base_height, rope_base_width = cfg.sample_height * vae_sf, cfg.sample_width * vae_sf
    if not invert_scale_latents:
        latents = latents * scaling_factor
    if cfg.patch_size_t is not None:
        latents = pad_frames(latents, cfg.patch_size_t)
    timesteps = (sigmas.flatten() * 1000.0).long()
    if noise is None:
        noise = torch.zeros_like(latents).normal_(generator=generator)
    noisy = scheduler.add_noise(latents, noise, timesteps)
    b, f, c, h, w = latents.shape
    ofs = None if cfg.ofs_embed_dim is None else latents.new_full((b,), fill_value=2.0)
    rope = None
    if cfg.use_rotary_positional_embeddings:
        rope = prepare_rotary_positional_embeddings(h * vae_sf, w * vae_sf, f, vae_sf, cfg.patch_size, cfg.patch_size_t, cfg.attention_head_dim,
                                                    rope_base_height, rope_base_width)
    velocity = transformer(hidden_states=noisy.to(latents), encoder_hidden_states=encoder_hidden_states, timestep=timesteps, image_rotary_emb=rope,
                           ofs=ofs, return_dict=False)[0]
    pred = scheduler.get_velocity(velocity, noisy, timesteps)
    return pred, latents, sigmas


def sft_loss(pred, target, sigmas, scheduler: CogVideoXDDIMScheduler):
    """trainer.py:463-481 with ``prepare_loss_weights`` = 1 / (1 - alphas_cumprod[t]) (utils/diffusion.py:125-128)."""
    timesteps = (sigmas.flatten() * 1000.0).long()
    weights = 1 / (1 - scheduler.alphas_cumprod[timesteps])
    while weights.ndim < pred.ndim:
        weights = weights.unsqueeze(-1)
    loss = weights.float() * (pred.float() - target.float()).pow(2)
    return loss.mean(list(range(1, loss.ndim))).mean()
