"""CPU oracle for the row after Wan (BASELINE config 5): HunyuanVideo LoRA SFT with fp8 weight storage -- TEST INFRASTRUCTURE ONLY.

Nothing under ``finetrainers_amd/`` may import this file; there is no HunyuanVideo product path yet.  It exists so that the HunyuanVideo work starts the way
LTX, CogVideoX and Wan did: restatement first, pinned where the reference's own code can be executed.

What it restates, and how each part is pinned:
  * ``spec_forward`` -- ``HunyuanVideoModelSpecification.forward`` (finetrainers/models/hunyuan_video/base_specification.py:294-330): posterior sample
    (``compute_posterior = False``) or given latents, ``latents * vae.scaling_factor``, flow-match mix (functional/diffusion.py:4-11), integer timesteps,
    ``guidance * 1000``, DiT call with the condition dict, target ``noise - latents``.  PINNED: golden fixtures ``hunyuan.spec.*`` are produced by executing
    the reference's own ``forward`` (oracle/make_golden.py) around a recording stub transformer.
  * ``HunyuanVideoTransformer3DModel`` (3-axis rotary table, condition embedding, token refiner, dual-stream and single-stream blocks with joint
    [video | text] attention, per-head RMSNorm of q / k, AdaLN-zero modulation, output head) -- [upstream] diffusers 0.33 ``transformer_hunyuan_video.py``,
    absent from the reference tree and from this image: restated from the published algorithm, UNPINNED (parity unpinned for the DiT, as for the other
    models), anchored on the reference's call site and its dummy configuration (tests/models/hunyuan_video/base_specification.py:97-113).
  * fp8 weight storage (config 5's "fake-fp8 weight-cast") is the reference's ``apply_layerwise_casting`` (trainer/sft_trainer/trainer.py:111-118), restated
    in oracle/ltx.py (``apply_layerwise_casting``) for any module tree; it applies to this model unchanged.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .cogvideox import apply_rotary_emb
from .ltx import RMSNorm, TimestepEmbedding, get_timestep_embedding, native_sdpa
from .wan import posterior_sample


@dataclass
class HunyuanVideoConfig:
    """``HunyuanVideoTransformer3DModel`` hyper-parameters; defaults = hunyuanvideo-community/HunyuanVideo [upstream config.json]."""

    in_channels: int = 16
    out_channels: int = 16
    num_attention_heads: int = 24
    attention_head_dim: int = 128
    num_layers: int = 20
    num_single_layers: int = 40
    num_refiner_layers: int = 2
    mlp_ratio: float = 4.0
    patch_size: int = 2
    patch_size_t: int = 1
    qk_norm: str = "rms_norm"
    guidance_embeds: bool = True
    text_embed_dim: int = 4096
    pooled_projection_dim: int = 768
    rope_theta: float = 256.0
    rope_axes_dim: Tuple[int, int, int] = (16, 56, 56)

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @staticmethod
    def dummy() -> "HunyuanVideoConfig":
        """tests/models/hunyuan_video/base_specification.py:97-113."""
        return HunyuanVideoConfig(in_channels=4, out_channels=4, num_attention_heads=2, attention_head_dim=10, num_layers=2, num_single_layers=2,
                                  num_refiner_layers=1, patch_size=1, patch_size_t=1, guidance_embeds=True, text_embed_dim=16, pooled_projection_dim=8,
                                  rope_axes_dim=(2, 4, 4))


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] embeddings
# --------------------------------------------------------------------------------------------------------------------------
def rotary_tables(cfg: HunyuanVideoConfig, frames: int, height: int, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``HunyuanVideoRotaryPosEmbed``: integer (t, h, w) grid positions, one ``get_1d_rotary_pos_embed(use_real=True)`` per axis (fp32, each frequency
    repeated for its channel pair), concatenated along the channel axis -> (cos, sin) [F' H' W', head_dim]."""
    sizes = (frames // cfg.patch_size_t, height // cfg.patch_size, width // cfg.patch_size)
    grid = torch.stack(torch.meshgrid(*[torch.arange(0, n, dtype=torch.float32) for n in sizes], indexing="ij"), dim=0)
    cos, sin = [], []
    for i, dim in enumerate(cfg.rope_axes_dim):
        freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(grid[i].reshape(-1), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, dim=1), torch.cat(sin, dim=1)


class TextProjection(nn.Module):
    """PixArtAlphaTextProjection(in, hidden, act_fn="silu")."""

    def __init__(self, in_features: int, hidden: int):
        super().__init__()
        self.linear_1 = nn.Linear(in_features, hidden)
        self.linear_2 = nn.Linear(hidden, hidden)

    def forward(self, x):
        return self.linear_2(F.silu(self.linear_1(x)))


class CombinedTimestepTextProjEmbeddings(nn.Module):
    def __init__(self, embedding_dim: int, pooled_projection_dim: int):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.text_embedder = TextProjection(pooled_projection_dim, embedding_dim)

    def forward(self, timestep, pooled_projection):
        t = self.timestep_embedder(get_timestep_embedding(timestep, 256).to(dtype=pooled_projection.dtype))
        return t + self.text_embedder(pooled_projection)


class ConditionEmbedding(nn.Module):
    """Timestep + (embedded) guidance scale + pooled CLIP projection -> the conditioning vector of every AdaLN."""

    def __init__(self, embedding_dim: int, pooled_projection_dim: int, guidance_embeds: bool):
        super().__init__()
        self.timestep_embedder = TimestepEmbedding(256, embedding_dim)
        self.guidance_embedder = TimestepEmbedding(256, embedding_dim) if guidance_embeds else None
        self.text_embedder = TextProjection(pooled_projection_dim, embedding_dim)

    def forward(self, timestep, guidance, pooled_projection):
        cond = self.timestep_embedder(get_timestep_embedding(timestep, 256).to(dtype=pooled_projection.dtype))
        if self.guidance_embedder is not None:
            cond = cond + self.guidance_embedder(get_timestep_embedding(guidance, 256).to(dtype=pooled_projection.dtype))
        return cond + self.text_embedder(pooled_projection)


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] attention: joint [video | text] sequence (video FIRST), per-head RMSNorm, rotary embedding on the video tokens only
# --------------------------------------------------------------------------------------------------------------------------
class JointAttention(nn.Module):
    """diffusers ``Attention(qk_norm="rms_norm", bias=True, eps=1e-6)`` + ``HunyuanVideoAttnProcessor2_0``.  ``added_kv``: the dual-stream form (text has
    its own q / k / v / out projections and q / k norms); ``pre_only``: the single-stream form (one projection set over the concatenated tokens, no output
    projection here)."""

    def __init__(self, dim: int, heads: int, dim_head: int, added_kv: bool, pre_only: bool = False, qk_norm: bool = True):
        super().__init__()
        self.heads, self.added_kv = heads, added_kv
        self.to_q, self.to_k, self.to_v = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
        self.norm_q = RMSNorm(dim_head, 1e-6) if qk_norm else None
        self.norm_k = RMSNorm(dim_head, 1e-6) if qk_norm else None
        self.to_out = None if pre_only else nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        if added_kv:
            self.add_q_proj, self.add_k_proj, self.add_v_proj = nn.Linear(dim, dim), nn.Linear(dim, dim), nn.Linear(dim, dim)
            self.norm_added_q, self.norm_added_k = RMSNorm(dim_head, 1e-6), RMSNorm(dim_head, 1e-6)
            self.to_add_out = nn.Linear(dim, dim)

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, image_rotary_emb=None):
        n_text = 0 if encoder_hidden_states is None else encoder_hidden_states.shape[1]
        if not self.added_kv and encoder_hidden_states is not None:
            hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)
        split = lambda t: t.unflatten(2, (self.heads, -1)).transpose(1, 2)
        q, k, v = split(self.to_q(hidden_states)), split(self.to_k(hidden_states)), split(self.to_v(hidden_states))
        if self.norm_q is not None:
            q, k = self.norm_q(q), self.norm_k(k)
        if image_rotary_emb is not None:
            if not self.added_kv and n_text > 0:
                q = torch.cat([apply_rotary_emb(q[:, :, :-n_text], image_rotary_emb), q[:, :, -n_text:]], dim=2)
                k = torch.cat([apply_rotary_emb(k[:, :, :-n_text], image_rotary_emb), k[:, :, -n_text:]], dim=2)
            else:
                q, k = apply_rotary_emb(q, image_rotary_emb), apply_rotary_emb(k, image_rotary_emb)
        if self.added_kv and encoder_hidden_states is not None:
            eq, ek, ev = split(self.add_q_proj(encoder_hidden_states)), split(self.add_k_proj(encoder_hidden_states)), split(self.add_v_proj(encoder_hidden_states))
            eq, ek = self.norm_added_q(eq), self.norm_added_k(ek)
            q, k, v = torch.cat([q, eq], dim=2), torch.cat([k, ek], dim=2), torch.cat([v, ev], dim=2)
        o = native_sdpa(q, k, v, attention_mask).transpose(1, 2).flatten(2, 3).to(q.dtype)
        if encoder_hidden_states is None:
            return self.to_out[1](self.to_out[0](o)), None
        o, eo = o[:, :-n_text], o[:, -n_text:]
        if self.to_out is not None:
            o = self.to_out[1](self.to_out[0](o))
        if self.added_kv:
            eo = self.to_add_out(eo)
        return o, eo


class FeedForward(nn.Module):
    """diffusers ``FeedForward(dim, mult, activation_fn)``: "gelu-approximate" (blocks) or "linear-silu" (token refiner)."""

    def __init__(self, dim: int, mult: float, activation: str):
        super().__init__()
        inner = int(dim * mult)
        self.proj_in = nn.Linear(dim, inner)   # net.0.proj
        self.proj_out = nn.Linear(inner, dim)  # net.2
        self.activation = activation

    def forward(self, x):
        h = self.proj_in(x)
        h = F.gelu(h, approximate="tanh") if self.activation == "gelu-approximate" else F.silu(h)
        return self.proj_out(h)


class AdaLayerNormZero(nn.Module):
    """chunks: 6 (dual-stream block: shift / scale / gate of the attention, then of the feed-forward) or 3 (single-stream block)."""

    def __init__(self, dim: int, chunks: int):
        super().__init__()
        self.linear = nn.Linear(dim, chunks * dim)
        self.norm = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.chunks = chunks

    def forward(self, x, emb):
        parts = self.linear(F.silu(emb)).chunk(self.chunks, dim=1)
        x = self.norm(x) * (1 + parts[1][:, None]) + parts[0][:, None]
        return (x,) + tuple(parts[2:])


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] token refiner (the LLM text tokens are refined, conditioned on the timestep and their own masked mean)
# --------------------------------------------------------------------------------------------------------------------------
class TokenRefinerBlock(nn.Module):
    def __init__(self, heads: int, dim_head: int, mlp_ratio: float):
        super().__init__()
        dim = heads * dim_head
        self.norm1 = nn.LayerNorm(dim, elementwise_affine=True, eps=1e-6)
        self.attn = JointAttention(dim, heads, dim_head, added_kv=False, qk_norm=False)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=True, eps=1e-6)
        self.ff = FeedForward(dim, mlp_ratio, "linear-silu")
        self.norm_out_linear = nn.Linear(dim, 2 * dim)  # HunyuanVideoAdaNorm.linear

    def forward(self, hidden_states, temb, attention_mask):
        a, _ = self.attn(self.norm1(hidden_states), None, attention_mask)
        gate_msa, gate_mlp = self.norm_out_linear(F.silu(temb)).chunk(2, dim=1)
        hidden_states = hidden_states + a * gate_msa.unsqueeze(1)
        return hidden_states + self.ff(self.norm2(hidden_states)) * gate_mlp.unsqueeze(1)


class TokenRefiner(nn.Module):
    def __init__(self, cfg: HunyuanVideoConfig):
        super().__init__()
        dim = cfg.inner_dim
        self.time_text_embed = CombinedTimestepTextProjEmbeddings(dim, cfg.text_embed_dim)
        self.proj_in = nn.Linear(cfg.text_embed_dim, dim)
        self.refiner_blocks = nn.ModuleList([TokenRefinerBlock(cfg.num_attention_heads, cfg.attention_head_dim, cfg.mlp_ratio)
                                             for _ in range(cfg.num_refiner_layers)])

    def forward(self, hidden_states, timestep, attention_mask=None):
        if attention_mask is None:
            pooled = hidden_states.mean(dim=1)
            self_mask = None
        else:
            m = attention_mask.float().unsqueeze(-1)
            pooled = ((hidden_states * m).sum(dim=1) / m.sum(dim=1)).to(hidden_states.dtype)
            B, T = attention_mask.shape
            m1 = attention_mask.bool().view(B, 1, 1, T).repeat(1, 1, T, 1)
            self_mask = (m1 & m1.transpose(2, 3)).bool()
            self_mask[:, :, :, 0] = True  # every query may look at the first token (keeps fully padded rows finite)
        temb = self.time_text_embed(timestep, pooled)
        hidden_states = self.proj_in(hidden_states)
        for blk in self.refiner_blocks:
            hidden_states = blk(hidden_states, temb, self_mask)
        return hidden_states


# --------------------------------------------------------------------------------------------------------------------------
# [upstream] blocks and model
# --------------------------------------------------------------------------------------------------------------------------
class DualStreamBlock(nn.Module):
    def __init__(self, cfg: HunyuanVideoConfig):
        super().__init__()
        dim = cfg.inner_dim
        self.norm1, self.norm1_context = AdaLayerNormZero(dim, 6), AdaLayerNormZero(dim, 6)
        self.attn = JointAttention(dim, cfg.num_attention_heads, cfg.attention_head_dim, added_kv=True)
        self.norm2 = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff = FeedForward(dim, cfg.mlp_ratio, "gelu-approximate")
        self.norm2_context = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.ff_context = FeedForward(dim, cfg.mlp_ratio, "gelu-approximate")

    def forward(self, hidden_states, encoder_hidden_states, temb, attention_mask, freqs_cis):
        n, gate_msa, shift_mlp, scale_mlp, gate_mlp = self.norm1(hidden_states, temb)
        nc, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = self.norm1_context(encoder_hidden_states, temb)
        a, ac = self.attn(n, nc, attention_mask, freqs_cis)
        hidden_states = hidden_states + a * gate_msa.unsqueeze(1)
        encoder_hidden_states = encoder_hidden_states + ac * c_gate_msa.unsqueeze(1)
        n = self.norm2(hidden_states) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        nc = self.norm2_context(encoder_hidden_states) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
        hidden_states = hidden_states + gate_mlp.unsqueeze(1) * self.ff(n)
        encoder_hidden_states = encoder_hidden_states + c_gate_mlp.unsqueeze(1) * self.ff_context(nc)
        return hidden_states, encoder_hidden_states


class SingleStreamBlock(nn.Module):
    def __init__(self, cfg: HunyuanVideoConfig):
        super().__init__()
        dim = cfg.inner_dim
        mlp_dim = int(dim * cfg.mlp_ratio)
        self.attn = JointAttention(dim, cfg.num_attention_heads, cfg.attention_head_dim, added_kv=False, pre_only=True)
        self.norm = AdaLayerNormZero(dim, 3)
        self.proj_mlp = nn.Linear(dim, mlp_dim)
        self.proj_out = nn.Linear(dim + mlp_dim, dim)

    def forward(self, hidden_states, encoder_hidden_states, temb, attention_mask, image_rotary_emb):
        n_text = encoder_hidden_states.shape[1]
        hidden_states = torch.cat([hidden_states, encoder_hidden_states], dim=1)
        residual = hidden_states
        n, gate = self.norm(hidden_states, temb)
        mlp = F.gelu(self.proj_mlp(n), approximate="tanh")
        a, ac = self.attn(n[:, :-n_text], n[:, -n_text:], attention_mask, image_rotary_emb)
        hidden_states = gate.unsqueeze(1) * self.proj_out(torch.cat([torch.cat([a, ac], dim=1), mlp], dim=2)) + residual
        return hidden_states[:, :-n_text], hidden_states[:, -n_text:]


class HunyuanVideoTransformer3DModel(nn.Module):
    def __init__(self, cfg: HunyuanVideoConfig):
        super().__init__()
        self.cfg = self.config = cfg
        dim = cfg.inner_dim
        k = (cfg.patch_size_t, cfg.patch_size, cfg.patch_size)
        self.x_embedder = nn.Conv3d(cfg.in_channels, dim, kernel_size=k, stride=k)  # HunyuanVideoPatchEmbed.proj
        self.context_embedder = TokenRefiner(cfg)
        self.time_text_embed = ConditionEmbedding(dim, cfg.pooled_projection_dim, cfg.guidance_embeds)
        self.transformer_blocks = nn.ModuleList([DualStreamBlock(cfg) for _ in range(cfg.num_layers)])
        self.single_transformer_blocks = nn.ModuleList([SingleStreamBlock(cfg) for _ in range(cfg.num_single_layers)])
        self.norm_out_linear = nn.Linear(dim, 2 * dim)  # AdaLayerNormContinuous.linear
        self.norm_out = nn.LayerNorm(dim, elementwise_affine=False, eps=1e-6)
        self.proj_out = nn.Linear(dim, cfg.patch_size_t * cfg.patch_size * cfg.patch_size * cfg.out_channels)

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections, guidance=None, return_dict: bool = True,
                **kwargs):
        B, _, F_, H, W = hidden_states.shape
        p, pt = self.cfg.patch_size, self.cfg.patch_size_t
        f, h, w = F_ // pt, H // p, W // p
        rope = rotary_tables(self.cfg, F_, H, W)
        temb = self.time_text_embed(timestep, guidance, pooled_projections)
        x = self.x_embedder(hidden_states).flatten(2).transpose(1, 2)
        enc = self.context_embedder(encoder_hidden_states, timestep, encoder_attention_mask)
        S, T = x.shape[1], enc.shape[1]
        # keys beyond each sample's real text length are masked for every query ([B, 1, 1, S + T], broadcast over heads and queries)
        mask = torch.zeros(B, S + T, dtype=torch.bool)
        eff = S + encoder_attention_mask.sum(dim=1, dtype=torch.int)
        for i in range(B):
            mask[i, : eff[i]] = True
        mask = mask.unsqueeze(1).unsqueeze(1)
        for blk in self.transformer_blocks:
            x, enc = blk(x, enc, temb, mask, rope)
        for blk in self.single_transformer_blocks:
            x, enc = blk(x, enc, temb, mask, rope)
        scale, shift = self.norm_out_linear(F.silu(temb).to(x.dtype)).chunk(2, dim=1)
        x = self.norm_out(x) * (1 + scale)[:, None, :] + shift[:, None, :]
        x = self.proj_out(x)
        x = x.reshape(B, f, h, w, -1, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7)
        out = x.flatten(6, 7).flatten(4, 5).flatten(2, 3)
        return (out,) if not return_dict else {"sample": out}


def build_model(cfg: HunyuanVideoConfig, seed: int = 0, dtype: torch.dtype = torch.bfloat16) -> HunyuanVideoTransformer3DModel:
    torch.manual_seed(seed)
    return HunyuanVideoTransformer3DModel(cfg).to(dtype)


# --------------------------------------------------------------------------------------------------------------------------
# specification forward + loss (the reference's own code paths, pinned)
# --------------------------------------------------------------------------------------------------------------------------
def spec_forward(transformer, latents: torch.Tensor, conditions: dict, sigmas: torch.Tensor, noise: torch.Tensor, scaling_factor: float = 0.476986,
                 guidance: float = 1.0, compute_posterior: bool = True, posterior_noise: Optional[torch.Tensor] = None):
    """base_specification.py:294-330.  ``compute_posterior = False``: ``latents`` are the stored VAE moments [B, 2C, F, H, W] and are sampled with
    ``posterior_noise`` first.  ``conditions``: encoder_hidden_states, encoder_attention_mask, pooled_projections.  ``sigmas`` already expanded to the
    latents' rank."""
    if not compute_posterior:
        latents = posterior_sample(latents, posterior_noise)
    latents = latents * scaling_factor
    noisy = (1.0 - sigmas) * latents + sigmas * noise  # functional/diffusion.py:4-6
    timesteps = (sigmas.flatten() * 1000.0).long()
    g = latents.new_full((latents.size(0),), fill_value=guidance) * 1000.0
    pred = transformer(hidden_states=noisy.to(latents), guidance=g, **conditions, timestep=timesteps, return_dict=False)[0]
    return pred, noise - latents, sigmas  # functional/diffusion.py:9-11
