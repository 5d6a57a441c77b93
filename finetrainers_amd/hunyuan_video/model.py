"""HunyuanVideo transformer for LoRA SFT on the MI355X (BASELINE config 5) with the call contract of the diffusers model the reference
drives (finetrainers/models/hunyuan_video/base_specification.py:318-326): ``forward(hidden_states [B, C, F, H, W], timestep, encoder_hidden_states
[B, T, 4096], encoder_attention_mask [B, T], pooled_projections [B, 768], guidance [B])`` -> ``(velocity [B, C, F, H, W],)``.

Restated in oracle/hunyuan.py ([upstream] diffusers transformer_hunyuan_video.py).  Everything outside the 60 blocks is frozen under the default LoRA target
(sft_trainer/config.py:24-26 matches ``transformer_blocks`` / ``single_transformer_blocks`` only) and runs forward-only: patch embedding as a GEMM over
patch columns, condition embedding (timestep + guidance + pooled CLIP), the masked token refiner of the LLM tokens, then 20 dual-stream and 40
single-stream blocks (block.py), the AdaLN-continuous output norm, projection and un-patchify with the backward down to block 0.

Token refiner mask: the reference allows query i to see key j iff both are real tokens (or j = 0).  Here the padded KEYS are masked (per-sample key bias)
for every query: identical for the real tokens; the padded tokens' own states differ, and they never reach the loss (their keys are masked in every block,
their outputs are not part of the prediction), so neither the loss nor any gradient sees the difference."""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..cogvideox.model import timestep_embedding
from .block import MI355XHunyuanDualBlock, MI355XHunyuanSingleBlock

bf16 = torch.bfloat16


@dataclass
class HunyuanVideoTransformerConfig:
    """diffusers ``HunyuanVideoTransformer3DModel`` config keys; defaults = hunyuanvideo-community/HunyuanVideo."""

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

    @classmethod
    def from_dict(cls, d: Dict) -> "HunyuanVideoTransformerConfig":
        known = {k: d[k] for k in cls.__dataclass_fields__ if k in d}
        if "rope_axes_dim" in known:
            known["rope_axes_dim"] = tuple(known["rope_axes_dim"])
        return cls(**known)


def rotary_tables(cfg: HunyuanVideoTransformerConfig, frames: int, height: int, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``HunyuanVideoRotaryPosEmbed`` for LATENT sizes: integer (t, h, w) grid positions, one fp32 frequency table per axis with every frequency repeated for
    its channel pair, concatenated along the channel axis -> (cos, sin) fp32 [F' H' W', head_dim]."""
    sizes = (frames // cfg.patch_size_t, height // cfg.patch_size, width // cfg.patch_size)
    grid = torch.stack(torch.meshgrid(*[torch.arange(0, n, dtype=torch.float32) for n in sizes], indexing="ij"), dim=0)
    cos, sin = [], []
    for i, dim in enumerate(cfg.rope_axes_dim):
        freqs = 1.0 / (cfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim))
        ang = torch.outer(grid[i].reshape(-1), freqs)
        cos.append(ang.cos().repeat_interleave(2, dim=1).float())
        sin.append(ang.sin().repeat_interleave(2, dim=1).float())
    return torch.cat(cos, dim=1).contiguous(), torch.cat(sin, dim=1).contiguous()


def _front_keys(cfg: HunyuanVideoTransformerConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    """Frozen parameters outside the blocks: diffusers name -> shape (the Conv3d patch embedding in its GEMM shape)."""
    D, mlp = cfg.inner_dim, int(cfg.inner_dim * cfg.mlp_ratio)
    pk = cfg.in_channels * cfg.patch_size_t * cfg.patch_size * cfg.patch_size
    po = cfg.out_channels * cfg.patch_size_t * cfg.patch_size * cfg.patch_size
    lin = lambda name, o, i: [(f"{name}.weight", (o, i)), (f"{name}.bias", (o,))]
    keys = lin("x_embedder.proj", D, pk)
    for emb in ("timestep_embedder",) + (("guidance_embedder",) if cfg.guidance_embeds else ()):
        keys += lin(f"time_text_embed.{emb}.linear_1", D, 256) + lin(f"time_text_embed.{emb}.linear_2", D, D)
    keys += lin("time_text_embed.text_embedder.linear_1", D, cfg.pooled_projection_dim) + lin("time_text_embed.text_embedder.linear_2", D, D)
    ce = "context_embedder"
    keys += lin(f"{ce}.time_text_embed.timestep_embedder.linear_1", D, 256) + lin(f"{ce}.time_text_embed.timestep_embedder.linear_2", D, D)
    keys += lin(f"{ce}.time_text_embed.text_embedder.linear_1", D, cfg.text_embed_dim) + lin(f"{ce}.time_text_embed.text_embedder.linear_2", D, D)
    keys += lin(f"{ce}.proj_in", D, cfg.text_embed_dim)
    for i in range(cfg.num_refiner_layers):
        p = f"{ce}.token_refiner.refiner_blocks.{i}"
        keys += [(f"{p}.norm1.weight", (D,)), (f"{p}.norm1.bias", (D,)), (f"{p}.norm2.weight", (D,)), (f"{p}.norm2.bias", (D,))]
        for t in ("to_q", "to_k", "to_v", "to_out.0"):
            keys += lin(f"{p}.attn.{t}", D, D)
        keys += lin(f"{p}.ff.net.0.proj", mlp, D) + lin(f"{p}.ff.net.2", D, mlp) + lin(f"{p}.norm_out.linear", 2 * D, D)
    keys += lin("norm_out.linear", 2 * D, D) + lin("proj_out", po, D)
    return keys


class _HeadFunction(torch.autograd.Function):
    """Video tokens of the last block -> AdaLN-continuous (LayerNorm without affine, (1 + scale), shift from the conditioning vector) -> proj_out."""

    @staticmethod
    def forward(ctx, m: "MI355XHunyuanVideoTransformer3DModel", x, shift, onep):
        B, S, D = x.shape
        n = ops.cog_ln_mod(x, m.ones, m.zeros, shift, onep, 0, 1e-6)
        y = ops.gemm_nt(n.view(B * S, D), m.p["proj_out.weight"], m.p["proj_out.bias"])
        ctx.m = m
        ctx.save_for_backward(x, onep)
        return y.view(B, S, -1)

    @staticmethod
    def backward(ctx, dy):
        m = ctx.m
        x, onep = ctx.saved_tensors
        B, S, D = x.shape
        dn = ops.gemm_nt(dy.contiguous().view(B * S, -1), m.proj_out_w_t, None)
        return None, ops.cog_ln_mod_bwd(x, m.ones, onep, dn.view(B, S, D), 0, 1e-6), None, None


class MI355XHunyuanVideoTransformer3DModel(nn.Module):
    def __init__(self, config: Optional[HunyuanVideoTransformerConfig] = None, device: Optional[torch.device] = None):
        super().__init__()
        self.config = c = config or HunyuanVideoTransformerConfig()
        if c.attention_head_dim != 128 or c.qk_norm != "rms_norm" or sum(c.rope_axes_dim) != 128:
            raise ValueError("this path covers the HunyuanVideo architecture: heads of 128 channels with per-head RMSNorm and a 128-channel rotary table")
        dev = device or torch.device("cuda", 0)
        D = c.inner_dim
        self.p: Dict[str, torch.Tensor] = {}
        for name, shape in _front_keys(c):  # plain dict of frozen tensors (registered as buffers under sanitised names)
            t = torch.zeros(shape, dtype=bf16, device=dev)
            self.register_buffer("front_" + name.replace(".", "_"), t)
            self.p[name] = t
        self.register_buffer("proj_out_w_t", None, persistent=False)
        self.register_buffer("ones", torch.ones(D, dtype=bf16, device=dev), persistent=False)
        self.register_buffer("zeros", torch.zeros(D, dtype=bf16, device=dev), persistent=False)
        self.transformer_blocks = nn.ModuleList([MI355XHunyuanDualBlock(D, c.num_attention_heads, c.mlp_ratio, dev) for _ in range(c.num_layers)])
        self.single_transformer_blocks = nn.ModuleList([MI355XHunyuanSingleBlock(D, c.num_attention_heads, c.mlp_ratio, dev) for _ in range(c.num_single_layers)])
        self._rope_cache: Dict[Tuple[int, int, int], Tuple[torch.Tensor, torch.Tensor]] = {}

    @property
    def device(self) -> torch.device:
        return self.ones.device

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """A diffusers ``HunyuanVideoTransformer3DModel`` state dict (peft ``.base_layer.`` infix accepted, LoRA tensors ignored here)."""
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items() if "lora_" not in k}
        missing = [k for k in self.p if k not in sd]
        if missing:
            raise KeyError(f"HunyuanVideo state dict lacks {missing[:4]}")
        for name, t in self.p.items():
            t.copy_(sd[name].to(bf16).reshape(t.shape))  # Conv3d weight [D, C, pt, p, p] -> [D, C pt p p]
        self.proj_out_w_t = ops.transpose_bf16(self.p["proj_out.weight"])
        for prefix, blocks in (("transformer_blocks", self.transformer_blocks), ("single_transformer_blocks", self.single_transformer_blocks)):
            for i, blk in enumerate(blocks):
                pre = f"{prefix}.{i}."
                blk.load_diffusers_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})

    def add_adapter(self, r: int = 64, lora_alpha: float = 64.0) -> None:
        """LoRA on to_q / to_k / to_v / to_out.0 of the 60 blocks (the default target regex): 20 x 4 + 40 x 3 = 200 adapters."""
        for blk in list(self.transformer_blocks) + list(self.single_transformer_blocks):
            blk.add_adapter(r, lora_alpha)

    def apply_activation_checkpointing(self, checkpointing_type: str = "full", n_layer: int = 1) -> "MI355XHunyuanVideoTransformer3DModel":
        """``--gradient_checkpointing`` (trainer/sft_trainer/trainer.py:155-157 -> utils/activation_checkpoint.py:24-49): "full" = every block of both
        block lists keeps only its input and runs its forward kernels a second time inside the backward, "block_skip" = every ``n_layer``-th block does.
        Same gradients bit for bit (deterministic kernels); at config 5's 32 896 tokens the saved activations shrink from ~2.4 GB to 0.2 GB per block at
        the price of one more forward pass (up to and including the attention) per checkpointed block.  "ops" (selective per-op saving of a traced graph)
        has no counterpart here: the blocks already keep exactly the tensors their backward reads."""
        if checkpointing_type == "ops":
            raise ValueError("checkpointing type 'ops' is not supported: the blocks already save only what their backward reads; use 'full' or 'block_skip'")
        if checkpointing_type not in ("full", "block_skip"):
            raise ValueError(f"Checkpointing type '{checkpointing_type}' not supported. Supported types are ['full', 'ops', 'block_skip']")
        for blocks in (self.transformer_blocks, self.single_transformer_blocks):
            for i, blk in enumerate(blocks):
                blk.gradient_checkpointing = checkpointing_type == "full" or i % max(1, int(n_layer)) == 0
        return self

    def enable_gradient_checkpointing(self) -> None:  # diffusers ModelMixin spelling
        self.apply_activation_checkpointing("full")

    def disable_gradient_checkpointing(self) -> None:
        for blk in list(self.transformer_blocks) + list(self.single_transformer_blocks):
            blk.gradient_checkpointing = False

    @property
    def is_gradient_checkpointing(self) -> bool:
        return any(blk.gradient_checkpointing for blk in list(self.transformer_blocks) + list(self.single_transformer_blocks))

    def lora_parameters(self) -> List[nn.Parameter]:
        return [p for blk in list(self.transformer_blocks) + list(self.single_transformer_blocks) for p in (blk.lora_A, blk.lora_B) if p is not None]

    def lora_state_dict(self) -> Dict[str, torch.Tensor]:
        """peft-format keys: ``transformer_blocks.N.attn.to_q.lora_A.weight`` ..."""
        out = {}
        for prefix, blocks, names in (("transformer_blocks", self.transformer_blocks, ("to_q", "to_k", "to_v", "to_out.0")),
                                      ("single_transformer_blocks", self.single_transformer_blocks, ("to_q", "to_k", "to_v"))):
            for i, blk in enumerate(blocks):
                r = blk.lora_rank_user  # (views of the user's rank inside the zero-padded storage)
                for j, n in enumerate(names):
                    out[f"{prefix}.{i}.attn.{n}.lora_A.weight"] = blk.lora_A[j, :r]
                    out[f"{prefix}.{i}.attn.{n}.lora_B.weight"] = blk.lora_B[j, :, :r]
        return out

    def lora_grad_state_dict(self) -> Dict[str, torch.Tensor]:
        """The adapters' gradients under the same peft-format keys (after a backward)."""
        out = {}
        for prefix, blocks, names in (("transformer_blocks", self.transformer_blocks, ("to_q", "to_k", "to_v", "to_out.0")),
                                      ("single_transformer_blocks", self.single_transformer_blocks, ("to_q", "to_k", "to_v"))):
            for i, blk in enumerate(blocks):
                r = blk.lora_rank_user
                for j, n in enumerate(names):
                    out[f"{prefix}.{i}.attn.{n}.lora_A.weight"] = blk.lora_A.grad[j, :r]
                    out[f"{prefix}.{i}.attn.{n}.lora_B.weight"] = blk.lora_B.grad[j, :, :r]
        return out

    @torch.no_grad()
    def load_lora_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k.replace(".default.", "."): v for k, v in sd.items()}
        for k, v in self.lora_state_dict().items():
            v.copy_(sd[k].to(v))

    @torch.no_grad()
    def apply_layerwise_casting(self, storage_dtype: torch.dtype = torch.float8_e4m3fn, real_storage: bool = True) -> int:
        """Config 5's "fake-fp8 weight cast" (trainer/sft_trainer/trainer.py:111-118 -> diffusers ``apply_layerwise_casting``): Linear weights and biases of
        the blocks are STORED in fp8 and cast up to bf16 for every forward; the up-cast is exact, so the arithmetic is that of bf16 weights holding
        fp8-representable values.  ``real_storage`` (default): the 2-D weights really live as e4m3fn bytes -- 1 byte per frozen parameter instead of the 4 of
        a bf16 copy plus its transposed twin -- and every block casts its weights up into a bf16 arena shared by all blocks right before it runs
        (``ftmi_fp8_upcast``; forward layout / transposed layout for the input-gradient GEMMs).  ``real_storage=False``: the weights are rounded to
        fp8-representable values and kept in bf16 (same numbers, no saving).  Norm layers (the AdaLN Linears included), embeddings and the output
        projection are skipped like the reference's default pattern (args.py:395).  Returns the number of tensors cast."""
        if storage_dtype != torch.float8_e4m3fn and real_storage:
            raise ValueError("real fp8 storage is e4m3fn (the reference's --layerwise_upcasting_storage_dtype default); other dtypes: real_storage=False")
        n = 0
        blocks = list(self.transformer_blocks) + list(self.single_transformer_blocks)
        for blk in blocks:
            for name, buf in blk.named_buffers():
                if buf is None or name.endswith("_t") or name in ("ones", "zeros") or name.startswith("norm"):
                    continue
                buf.copy_(buf.to(storage_dtype).to(bf16))
                n += 1
            if real_storage:
                blk.store_weights_fp8()
            else:
                for name in getattr(blk, "_TRANSPOSED", ("wq", "wk", "wv", "proj_mlp_w", "proj_out_w")):
                    setattr(blk, name + "_t", ops.transpose_bf16(getattr(blk, name)))
        if real_storage and blocks:
            n_fwd = max(blk.fp8_elements()[0] for blk in blocks)
            n_bwd = max(blk.fp8_elements()[1] for blk in blocks)
            self._weight_arena_fwd = torch.empty(n_fwd, dtype=bf16, device=self.device)
            self._weight_arena_bwd = torch.empty(n_bwd, dtype=bf16, device=self.device)
            for blk in blocks:
                blk._arena_fwd, blk._arena_bwd = self._weight_arena_fwd, self._weight_arena_bwd
        return n

    # -- frozen front: forward only -----------------------------------------------------------------------------------------------------------------
    def _lin(self, x2d: torch.Tensor, name: str) -> torch.Tensor:
        return ops.gemm_nt(x2d.contiguous(), self.p[f"{name}.weight"], self.p[f"{name}.bias"])

    def _mlp(self, x2d: torch.Tensor, name: str) -> torch.Tensor:  # TimestepEmbedding / PixArtAlphaTextProjection(act_fn = "silu")
        return self._lin(torch.nn.functional.silu(self._lin(x2d, f"{name}.linear_1")), f"{name}.linear_2")

    @torch.no_grad()
    def _conditioning(self, timestep, guidance, pooled) -> torch.Tensor:
        c = self.config
        cond = self._mlp(timestep_embedding(timestep.to(self.device), 256).to(bf16), "time_text_embed.timestep_embedder")
        if c.guidance_embeds:
            cond = cond + self._mlp(timestep_embedding(guidance.to(self.device), 256).to(bf16), "time_text_embed.guidance_embedder")
        return cond + self._mlp(pooled.to(bf16), "time_text_embed.text_embedder")

    @torch.no_grad()
    def _refine_text(self, text, timestep, mask) -> torch.Tensor:
        c = self.config
        B, T, _ = text.shape
        D, H = c.inner_dim, c.num_attention_heads
        text = text.to(bf16)
        ce = "context_embedder"
        if mask is None:
            pooled = text.mean(dim=1)
            key_bias = None
        else:
            mf = mask.to(self.device).float().unsqueeze(-1)
            pooled = ((text * mf).sum(dim=1) / mf.sum(dim=1)).to(bf16)
            key_bias = torch.zeros((B, T), dtype=torch.float32, device=self.device).masked_fill_(~mask.to(self.device).bool(), float("-inf"))
        temb = self._mlp(timestep_embedding(timestep.to(self.device), 256).to(bf16), f"{ce}.time_text_embed.timestep_embedder") + self._mlp(pooled, f"{ce}.time_text_embed.text_embedder")
        temb_silu = torch.nn.functional.silu(temb)
        h = self._lin(text.reshape(B * T, -1), f"{ce}.proj_in").view(B, T, D)
        zeros_row, ones_row = torch.zeros(B, D, dtype=bf16, device=self.device), torch.ones(B, D, dtype=bf16, device=self.device)
        heads = lambda t: t.view(B, T, H, 128).permute(0, 2, 1, 3)
        for i in range(c.num_refiner_layers):
            p = f"{ce}.token_refiner.refiner_blocks.{i}"
            n1 = ops.cog_ln_mod(h, self.p[f"{p}.norm1.weight"], self.p[f"{p}.norm1.bias"], zeros_row, ones_row, 0, 1e-6).view(B * T, D)
            o, _ = ops.attn_fwd(heads(self._lin(n1, f"{p}.attn.to_q")), heads(self._lin(n1, f"{p}.attn.to_k")), heads(self._lin(n1, f"{p}.attn.to_v")), key_bias)
            a = self._lin(o.permute(0, 2, 1, 3).reshape(B * T, D), f"{p}.attn.to_out.0")
            gates = self._lin(temb_silu, f"{p}.norm_out.linear").view(B, 2, D)
            h = ops.cog_gate_residual(h, a.view(B, T, D), gates[:, 0].contiguous(), 0)
            n2 = ops.cog_ln_mod(h, self.p[f"{p}.norm2.weight"], self.p[f"{p}.norm2.bias"], zeros_row, ones_row, 0, 1e-6).view(B * T, D)
            f = self._lin(torch.nn.functional.silu(self._lin(n2, f"{p}.ff.net.0.proj")), f"{p}.ff.net.2")
            h = ops.cog_gate_residual(h, f.view(B, T, D), gates[:, 1].contiguous(), 0)
        return h

    def _rope(self, frames: int, height: int, width: int):
        key = (frames, height, width)
        if key not in self._rope_cache:
            self._rope_cache[key] = tuple(t.to(self.device) for t in rotary_tables(self.config, frames, height, width))
        return self._rope_cache[key]

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_attention_mask, pooled_projections, guidance=None, return_dict: bool = False,
                **kwargs):
        if self.proj_out_w_t is None:
            raise RuntimeError("load_diffusers_state_dict first")
        c = self.config
        if c.guidance_embeds and guidance is None:
            raise ValueError("this checkpoint embeds the guidance scale: pass guidance [B]")
        B, C, F_, H, W = hidden_states.shape
        p, pt = c.patch_size, c.patch_size_t
        f, h, w = F_ // pt, H // p, W // p
        D, S = c.inner_dim, f * h * w
        rope = self._rope(F_, H, W)
        with torch.no_grad():
            temb = self._conditioning(timestep, guidance, pooled_projections)
            cols = hidden_states.to(bf16).view(B, C, f, pt, h, p, w, p).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B * S, C * pt * p * p)
            x = self._lin(cols, "x_embedder.proj").view(B, S, D)
            enc = self._refine_text(encoder_hidden_states, timestep, encoder_attention_mask)
            T = enc.shape[1]
            mod = self._lin(torch.nn.functional.silu(temb), "norm_out.linear").view(B, 2, D)  # AdaLayerNormContinuous: scale, shift = chunk(2)
            shift_out, onep_out = mod[:, 1].contiguous(), (1 + mod[:, 0]).contiguous()
        mask = encoder_attention_mask
        for blk in self.transformer_blocks:
            x, enc = blk(x, enc, temb, rope, text_mask=mask)
        tokens = torch.cat([enc, x], dim=1)  # the single-stream blocks work on one joint buffer, text first
        for blk in self.single_transformer_blocks:
            tokens = blk(tokens, temb, T, rope, text_mask=mask)
        y = _HeadFunction.apply(self, tokens[:, T:].contiguous(), shift_out, onep_out)
        out = y.reshape(B, f, h, w, -1, pt, p, p).permute(0, 4, 1, 5, 2, 6, 3, 7).reshape(B, -1, F_, H, W)
        return {"sample": out} if return_dict else (out,)
