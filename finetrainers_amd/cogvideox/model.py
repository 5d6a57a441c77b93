"""CogVideoX-2b DiT for LoRA SFT on the gfx950 kernels: patch embed + sincos table, time embedding, the blocks of ``block.py``, final norms,
``proj_out`` and un-patchify -- forward, and the backward that reaches the LoRA adapters (everything outside the blocks is frozen, and nothing
below block 0 needs a gradient).

Reference: [upstream] diffusers ``CogVideoXTransformer3DModel`` as the reference drives it (``finetrainers/models/cogvideox/base_specification.py:
296-333``), restated in ``oracle/cogvideox.py``.  The sincos-table checkpoints (CogVideoX-2b, BASELINE config 3), the rotary ones (5b: ``use_rotary_positional_embeddings``, RoPE on the
video rows of q / k) and the 1.5 family (``patch_size_t``: patches over two latent frames embedded by a Linear, ``ofs_embed_dim``: an offset embedding added to the
time embedding, integer-position rotary tables) are covered.

Token layout: ONE buffer ``[B, T + S, D]``, the T = ``max_text_seq_length`` text tokens first.  Orchestration is Python over C-ABI calls (see
``block.py``); torch ops touch only per-sample conditioning vectors ([B, 1920] / [B, 512]) and the host-built constant tables.
"""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

import ctypes

from .. import _lib, ops
from .._lib import CogConfig, CogWeights, check, ptr, stream_ptr
from .block import MI355XCogVideoXBlock

bf16 = torch.bfloat16


@dataclass
class CogVideoXTransformerConfig:
    """[upstream] CogVideoX-2b ``transformer/config.json`` values."""

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
    temporal_compression_ratio: int = 4
    max_text_seq_length: int = 226
    norm_eps: float = 1e-5
    spatial_interpolation_scale: float = 1.875
    temporal_interpolation_scale: float = 1.0
    ff_mult: int = 4
    use_rotary_positional_embeddings: bool = False  # 2b: sincos table added in the patch embed; 5b: rotary embedding inside the attention
    patch_size_t: Optional[int] = None   # CogVideoX 1.5: 2 (patches span patch_size_t latent frames; the patch embedding is a Linear)
    ofs_embed_dim: Optional[int] = None  # CogVideoX 1.5: 512 (an "offset" embedding added to the time embedding)
    patch_bias: bool = True              # CogVideoX 1.5: False

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim


def _sincos_1d(dim: int, pos: torch.Tensor) -> torch.Tensor:
    omega = 1.0 / 10000 ** (torch.arange(dim // 2, dtype=torch.float64) / (dim / 2.0))
    ang = torch.outer(pos.reshape(-1).to(torch.float64), omega)
    return torch.cat([torch.sin(ang), torch.cos(ang)], dim=1)


def sincos_position_table(cfg: CogVideoXTransformerConfig, height: int, width: int, frames: int) -> torch.Tensor:
    """[upstream] ``get_3d_sincos_pos_embed`` as ``CogVideoXPatchEmbed`` lays it out: rows = latent frames x (height/p) x (width/p) patches; the first
    D/4 channels encode the frame, the other 3D/4 the 2-D position (width half, then height half); spatial grid divided by the interpolation scale."""
    D, p = cfg.inner_dim, cfg.patch_size
    ph, pw = height // p, width // p
    d_sp, d_t = 3 * D // 4, D // 4
    gh = torch.arange(ph, dtype=torch.float32) / cfg.spatial_interpolation_scale
    gw = torch.arange(pw, dtype=torch.float32) / cfg.spatial_interpolation_scale
    mw, mh = torch.meshgrid(gw, gh, indexing="xy")  # [ph, pw] each: width varies fastest
    pos_sp = torch.cat([_sincos_1d(d_sp // 2, mw), _sincos_1d(d_sp // 2, mh)], dim=1)  # [ph * pw, d_sp]
    pos_t = _sincos_1d(d_t, torch.arange(frames, dtype=torch.float32) / cfg.temporal_interpolation_scale)  # [frames, d_t]
    table = torch.cat([pos_t[:, None, :].expand(frames, ph * pw, d_t), pos_sp[None].expand(frames, ph * pw, d_sp)], dim=-1)
    return table.reshape(frames * ph * pw, D).float()


def rotary_tables(cfg: CogVideoXTransformerConfig, height: int, width: int, frames: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``prepare_rotary_positional_embeddings`` (finetrainers/models/cogvideox/utils.py:8-51, both branches) for LATENT sizes:
    (cos, sin) fp32 [frames * gh * gw, 64]; head channels split t : h : w = 16 : 24 : 24, each frequency repeated for its channel pair; spatial positions
    are a linspace over the crop of the base grid (sample_height / sample_width) that matches this clip's aspect ratio."""
    p, d = cfg.patch_size, cfg.attention_head_dim
    gh, gw, bh, bw = height // p, width // p, cfg.sample_height // p, cfg.sample_width // p
    if cfg.patch_size_t is not None:
        # the ``patch_size_t`` branch (utils.py:38-51 -> [upstream] get_3d_rotary_pos_embed(grid_type="slice")): plain integer positions, one per patch row /
        # column / frame GROUP, cut out of the base grid
        frames = (frames + cfg.patch_size_t - 1) // cfg.patch_size_t
        if gh > bh or gw > bw:
            raise ValueError(f"CogVideoX 1.5 rotary tables: the clip's patch grid {gh} x {gw} exceeds the model's base grid {bh} x {bw}")
        grid_h, grid_w = torch.arange(gh, dtype=torch.float32), torch.arange(gw, dtype=torch.float32)
        grid_t = torch.arange(frames, dtype=torch.float32)
    else:
        if gh / gw > bh / bw:  # get_resize_crop_region_for_grid((gh, gw), bw, bh)
            rh, rw = bh, int(round(bh / gh * gw))
        else:
            rw, rh = bw, int(round(bw / gw * gh))
        top, left = int(round((bh - rh) / 2.0)), int(round((bw - rw) / 2.0))
        grid_h = torch.linspace(top, (top + rh) * (gh - 1) / gh, gh, dtype=torch.float32)
        grid_w = torch.linspace(left, (left + rw) * (gw - 1) / gw, gw, dtype=torch.float32)
        grid_t = torch.linspace(0, frames * (frames - 1) / frames, frames, dtype=torch.float32)

    def one(dim, pos):
        ang = torch.outer(pos, 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float32)[: dim // 2] / dim)))
        return ang.cos().repeat_interleave(2, dim=1).float(), ang.sin().repeat_interleave(2, dim=1).float()

    (ct, st), (ch, sh), (cw, sw) = one(d // 4, grid_t), one(d // 8 * 3, grid_h), one(d // 8 * 3, grid_w)

    def combine(t, h, w):
        return torch.cat([t[:, None, None, :].expand(-1, gh, gw, -1), h[None, :, None, :].expand(frames, -1, gw, -1),
                          w[None, None, :, :].expand(frames, gh, -1, -1)], dim=-1).reshape(frames * gh * gw, -1).contiguous()

    return combine(ct, ch, cw), combine(st, sh, sw)


def timestep_embedding(timesteps: torch.Tensor, dim: int) -> torch.Tensor:
    """[upstream] ``get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0)``: [cos | sin] of t * 10000^(-i / (dim/2))."""
    half = dim // 2
    freqs = torch.exp(-math.log(10000) * torch.arange(half, dtype=torch.float32, device=timesteps.device) / half)
    ang = timesteps[:, None].float() * freqs[None, :]
    return torch.cat([torch.cos(ang), torch.sin(ang)], dim=-1)


def patches_3d(latents: torch.Tensor, p: int, pt: int) -> torch.Tensor:
    """CogVideoX 1.5 patch layout ([upstream] CogVideoXPatchEmbed with ``patch_size_t``): latents [B, F, C, H, W] -> [B, (F/pt)(H/p)(W/p), C pt p p], channel
    order (channel, frame, row, column) -- also the layout ``proj_out`` produces, so the same permutation (inverted) un-patchifies the velocity."""
    B, F_, C, H, W = latents.shape
    x = latents.reshape(B, F_ // pt, pt, C, H // p, p, W // p, p).permute(0, 1, 4, 6, 3, 2, 5, 7)
    return x.reshape(B, (F_ // pt) * (H // p) * (W // p), C * pt * p * p).contiguous()


def unpatches_3d(tokens: torch.Tensor, F_: int, C: int, H: int, W: int, p: int, pt: int) -> torch.Tensor:
    B = tokens.shape[0]
    x = tokens.reshape(B, F_ // pt, H // p, W // p, C, pt, p, p).permute(0, 1, 5, 4, 2, 6, 3, 7)
    return x.reshape(B, F_, C, H, W).contiguous()


class _BlockStackFunction(torch.autograd.Function):
    """All blocks in one C call per direction (``ftmi_cog_blocks_forward`` / ``_backward``, csrc/cog_dit.hip): the workspace holds every activation the
    backward needs.  The LoRA gradients land in the model's flat gradient buffer (= the blocks' ``.grad`` views); with a data-parallel hook installed the
    backward runs in block ranges and each finished range is handed to the hook (bucketed all-reduce overlapped with the remaining blocks)."""

    @staticmethod
    def forward(ctx, m: "MI355XCogVideoXTransformer3DModel", tokens, temb_silu, rope, lora_anchor, keep_activations: bool):
        B, N, D = tokens.shape
        cfg = m._c_config(B, N)
        w = m._c_weights(rope)
        lib = _lib.load()
        nbytes = lib.ftmi_cog_workspace_bytes(ctypes.byref(cfg))
        ws = m._acquire_workspace(nbytes)
        out = torch.empty_like(tokens)
        check(lib.ftmi_cog_blocks_forward(ctypes.byref(cfg), ctypes.byref(w), ptr(tokens), ptr(temb_silu), ptr(out), ptr(ws), nbytes, stream_ptr()),
              "ftmi_cog_blocks_forward")
        if not keep_activations:  # no_grad evaluation: nothing will come back for the activations
            m._release_workspace(ws)
            return out
        ctx.m, ctx.cfg, ctx.w, ctx.ws, ctx.nbytes, ctx.rope = m, cfg, w, ws, nbytes, rope
        ctx.save_for_backward(tokens)
        return out

    @staticmethod
    def backward(ctx, dout):
        m = ctx.m
        (tokens,) = ctx.saved_tensors
        dout = dout.contiguous()
        L, r, D = m.config.num_layers, m.lora_rank, m.config.inner_dim
        n = L * 4 * r * D
        gflat = m.lora_grad_flat
        ga, gb = gflat[:n].view(L, 4, r, D), gflat[n:].view(L, 4, D, r)
        blk0 = m.transformer_blocks[0]
        accumulate = int(blk0.lora_A.grad is not None and blk0.lora_A.grad.data_ptr() == ga[0].data_ptr())
        hook = m._grad_bucket_hook
        step = m.grad_bucket_blocks if (hook is not None and m.grad_bucket_blocks > 0) else L
        lib = _lib.load()
        hi = L
        while hi > 0:
            lo = max(0, hi - step)
            check(lib.ftmi_cog_blocks_backward(ctypes.byref(ctx.cfg), ctypes.byref(ctx.w), ptr(tokens), ptr(dout), None, ptr(ga), ptr(gb), ptr(ctx.ws),
                                               ctx.nbytes, hi, lo, accumulate, stream_ptr()), "ftmi_cog_blocks_backward")
            if hook is not None:
                hook(lo, hi, ga[lo:hi], gb[lo:hi])
            hi = lo
        for i, blk in enumerate(m.transformer_blocks):
            blk.lora_A.grad, blk.lora_B.grad = ga[i], gb[i]
        m._release_workspace(ctx.ws)
        ctx.ws = None
        return None, None, None, None, None, None  # nothing below block 0 is trainable


class _HeadFunction(torch.autograd.Function):
    """Video tokens of the last block -> norm_final -> AdaLayerNorm(norm_out) -> proj_out -> un-patchify.  Backward: d tokens (text rows zero)."""

    @staticmethod
    def forward(ctx, m: "MI355XCogVideoXTransformer3DModel", tokens, onep_out, shift_out, geom):
        B, N, D = tokens.shape
        F_, H, W = geom
        T, S, p, C, pt = m.config.max_text_seq_length, N - m.config.max_text_seq_length, m.config.patch_size, m.config.out_channels, m.config.patch_size_t
        nf = torch.empty((B, S, D), dtype=bf16, device=tokens.device)
        no = torch.empty_like(nf)
        for b in range(B):  # the video rows of a sample are contiguous, the samples are T rows apart
            ops.cog_ln_mod(tokens[b:b + 1, T:], m.norm_final_w, m.norm_final_b, m._zeros_row, m._ones_row, 0, m.config.norm_eps, out=nf[b:b + 1])
        ops.cog_ln_mod(nf, m.norm_out_w, m.norm_out_b, shift_out, onep_out, 0, m.config.norm_eps, out=no)
        y = ops.gemm_nt(no.view(B * S, D), m.proj_out_w, m.proj_out_b)
        ctx.m, ctx.geom = m, geom
        ctx.save_for_backward(tokens, nf, onep_out)
        if pt is not None:
            return unpatches_3d(y.view(B, S, C * pt * p * p), F_, C, H, W, p, pt)
        return ops.cog_unpatchify(y.view(B, S, p * p * C), F_, C, H, W, p)

    @staticmethod
    def backward(ctx, dvel):
        m = ctx.m
        tokens, nf, onep_out = ctx.saved_tensors
        B, N, D = tokens.shape
        T, S, p = m.config.max_text_seq_length, N - m.config.max_text_seq_length, m.config.patch_size
        pt = m.config.patch_size_t
        dy = ops.cog_patchify(dvel.contiguous(), p) if pt is None else patches_3d(dvel.contiguous(), p, pt)  # the un-patchify's transpose is the patchify
        dno = ops.gemm_nt(dy.view(B * S, -1), m.proj_out_w_t, None)
        dnf = ops.cog_ln_mod_bwd(nf, m.norm_out_w, onep_out, dno.view(B, S, D), 0, m.config.norm_eps)
        dtok = torch.zeros_like(tokens)  # the text stream does not reach the output
        for b in range(B):
            dx = ops.cog_ln_mod_bwd(tokens[b:b + 1, T:], m.norm_final_w, m._ones_row, dnf[b:b + 1], 0, m.config.norm_eps)
            dtok[b, T:].copy_(dx[0])
        return None, dtok, None, None, None


class MI355XCogVideoXTransformer3DModel(nn.Module):
    """Same call contract as the diffusers model for the SFT path: ``forward(hidden_states [B, F, C, H, W], encoder_hidden_states [B, T, 4096],
    timestep [B])`` -> ``(velocity [B, F, C, H, W],)``."""

    def __init__(self, config: Optional[CogVideoXTransformerConfig] = None, device: Optional[torch.device] = None):
        super().__init__()
        self.config = c = config or CogVideoXTransformerConfig()
        if c.attention_head_dim != 64:
            raise ValueError("the gfx950 attention kernels need head_dim 64")
        dev = device or torch.device("cuda", 0)
        D, p = c.inner_dim, c.patch_size
        pvol = p * p * (c.patch_size_t or 1)  # elements of one patch per channel
        z = lambda *shape: torch.zeros(shape, dtype=bf16, device=dev)
        if c.ofs_embed_dim is not None:
            if c.ofs_embed_dim != c.time_embed_dim:
                raise ValueError("the ofs embedding is added to the time embedding: ofs_embed_dim must equal time_embed_dim")
            E = c.ofs_embed_dim
            for name, shape in (("ofs1_w", (E, E)), ("ofs1_b", (E,)), ("ofs2_w", (E, E)), ("ofs2_b", (E,))):
                self.register_buffer(name, z(*shape))
        for name, shape in (("patch_w", (D, c.in_channels * pvol)), ("patch_b", (D,)), ("text_w", (D, c.text_embed_dim)), ("text_b", (D,)),
                            ("time1_w", (c.time_embed_dim, D)), ("time1_b", (c.time_embed_dim,)), ("time2_w", (c.time_embed_dim, c.time_embed_dim)),
                            ("time2_b", (c.time_embed_dim,)), ("norm_final_w", (D,)), ("norm_final_b", (D,)), ("norm_out_lin_w", (2 * D, c.time_embed_dim)),
                            ("norm_out_lin_b", (2 * D,)), ("norm_out_w", (D,)), ("norm_out_b", (D,)), ("proj_out_w", (pvol * c.out_channels, D)),
                            ("proj_out_b", (pvol * c.out_channels,))):
            self.register_buffer(name, z(*shape))
        self.register_buffer("proj_out_w_t", None, persistent=False)
        self.register_buffer("_ones_row", torch.ones(1, D, dtype=bf16, device=dev), persistent=False)
        self.register_buffer("_zeros_row", torch.zeros(1, D, dtype=bf16, device=dev), persistent=False)
        self.transformer_blocks = nn.ModuleList([MI355XCogVideoXBlock(D, c.num_attention_heads, c.time_embed_dim, c.ff_mult, c.norm_eps, dev)
                                                 for _ in range(c.num_layers)])
        self._pos_cache: Dict[Tuple[int, int, int], torch.Tensor] = {}
        self.lora_flat: Optional[torch.Tensor] = None
        self.lora_rank = 0
        self.native_blocks = True  # all blocks as one C call per direction (csrc/cog_dit.hip); False: the per-block Python composition of block.py
        self._stack: Dict[str, torch.Tensor] = {}
        self._lora_copies: Optional[Dict[str, torch.Tensor]] = None
        self._lora_versions = None
        self._ws_pool: List[torch.Tensor] = []
        self._grad_bucket_hook = None  # callable(lo, hi, grad_a[lo:hi], grad_b[lo:hi]) set by the data-parallel step
        self.grad_bucket_blocks = 0

    @property
    def device(self) -> torch.device:
        return self.patch_w.device

    _KEYS = {
        "patch_embed.proj.bias": "patch_b", "patch_embed.text_proj.weight": "text_w", "patch_embed.text_proj.bias": "text_b",
        "time_embedding.linear_1.weight": "time1_w", "time_embedding.linear_1.bias": "time1_b", "time_embedding.linear_2.weight": "time2_w",
        "time_embedding.linear_2.bias": "time2_b", "norm_final.weight": "norm_final_w", "norm_final.bias": "norm_final_b",
        "norm_out.linear.weight": "norm_out_lin_w", "norm_out.linear.bias": "norm_out_lin_b", "norm_out.norm.weight": "norm_out_w",
        "norm_out.norm.bias": "norm_out_b", "proj_out.weight": "proj_out_w", "proj_out.bias": "proj_out_b",
    }

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """A diffusers ``CogVideoXTransformer3DModel`` state dict (peft ``.base_layer.`` infix accepted, LoRA tensors ignored here)."""
        sd = {k.replace(".base_layer.", "."): v for k, v in sd.items() if "lora_" not in k}
        keys = dict(self._KEYS)
        if not self.config.patch_bias:
            keys.pop("patch_embed.proj.bias")  # CogVideoX 1.5: no bias (the buffer stays zero)
        if self.config.ofs_embed_dim is not None:
            keys.update({"ofs_embedding.linear_1.weight": "ofs1_w", "ofs_embedding.linear_1.bias": "ofs1_b", "ofs_embedding.linear_2.weight": "ofs2_w",
                         "ofs_embedding.linear_2.bias": "ofs2_b"})
        for k, name in keys.items():
            getattr(self, name).copy_(sd[k].to(bf16))
        self.patch_w.copy_(sd["patch_embed.proj.weight"].reshape(self.patch_w.shape).to(bf16))  # Conv2d [D, C, p, p] -> [D, C p p]; 1.5: Linear [D, C pt p p]
        self.proj_out_w_t = ops.transpose_bf16(self.proj_out_w)
        for i, blk in enumerate(self.transformer_blocks):
            pre = f"transformer_blocks.{i}."
            blk.load_diffusers_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        self._build_stack()

    # ---- stacked weights / C structs of the native block stack --------------------------------------------------------------------------
    @torch.no_grad()
    def _build_stack(self) -> None:
        """Per-kind tensors with a leading layer axis (include/ftmi355.h: ftmi_cog_weights), built from the blocks' buffers at load time."""
        blocks = self.transformer_blocks
        cat = lambda f: torch.stack([f(b) for b in blocks]).contiguous()
        st = {
            "mod_w": cat(lambda b: torch.stack([b.norm1_lin_w, b.norm2_lin_w])), "mod_b": cat(lambda b: torch.stack([b.norm1_lin_b, b.norm2_lin_b])),
            "norm_w": cat(lambda b: torch.stack([b.norm1_w, b.norm2_w])), "norm_b": cat(lambda b: torch.stack([b.norm1_b, b.norm2_b])),
            "w_qkv": cat(lambda b: torch.cat([b.wq, b.wk, b.wv])), "b_qkv": cat(lambda b: torch.cat([b.bq, b.bk, b.bv])),
            "w_o": cat(lambda b: b.wo), "b_o": cat(lambda b: b.bo),
            "qk_norm": cat(lambda b: torch.stack([b.norm_q_w, b.norm_q_b, b.norm_k_w, b.norm_k_b])),
            "w_ff1": cat(lambda b: b.ff1_w), "b_ff1": cat(lambda b: b.ff1_b), "w_ff2": cat(lambda b: b.ff2_w), "b_ff2": cat(lambda b: b.ff2_b),
            "w_o_t": cat(lambda b: b.wo_t), "w_ff1_t": cat(lambda b: b.ff1_w_t), "w_ff2_t": cat(lambda b: b.ff2_w_t),
        }
        st["w_qkv_t"] = torch.stack([ops.transpose_bf16(wl) for wl in st["w_qkv"]]).contiguous()  # [L, D, 3D]
        self._stack = st
        D = self.config.inner_dim
        for i, b in enumerate(blocks):  # one copy of the large matrices: the blocks' buffers become views of the stacks
            b.wq, b.wk, b.wv = st["w_qkv"][i, :D], st["w_qkv"][i, D:2 * D], st["w_qkv"][i, 2 * D:]
            b.wo, b.ff1_w, b.ff2_w = st["w_o"][i], st["w_ff1"][i], st["w_ff2"][i]
            b.wo_t, b.ff1_w_t, b.ff2_w_t = st["w_o_t"][i], st["w_ff1_t"][i], st["w_ff2_t"][i]

    def _c_config(self, B: int, N: int) -> CogConfig:
        c = self.config
        return CogConfig(B=B, T=c.max_text_seq_length, S=N - c.max_text_seq_length, D=c.inner_dim, H=c.num_attention_heads, L=c.num_layers,
                         D_ff=c.ff_mult * c.inner_dim, D_temb=c.time_embed_dim, r=self.lora_rank,
                         lora_scale=(self.transformer_blocks[0].lora_scale if self.lora_rank else 0.0), eps_norm=c.norm_eps, eps_qk=1e-6, gemm_variant=8)

    @torch.no_grad()
    def _refresh_lora_copies(self) -> None:
        """bf16 (hi, lo) working copies of the flat fp32 adapters, rebuilt when the parameters changed (3 launches)."""
        if self.lora_flat is None:
            return
        ver = (self.lora_flat._version, self.lora_flat.data_ptr())
        if self._lora_copies is not None and ver == self._lora_versions:
            return
        L, r, D = self.config.num_layers, self.lora_rank, self.config.inner_dim
        if self._lora_copies is None:
            z = lambda *shape: torch.zeros(shape, dtype=bf16, device=self.device)
            self._lora_copies = {"lora_a_sp": z(L, 4, 2 * r, D), "lora_bt_sp": z(L, 4, 2 * r, D), "lora_b_ext": z(L, 4, D, 3 * r),
                                 "lora_at_ext": z(L, 4, D, 3 * r), "lora_at_qkv_ext": z(L, D, 9 * r)}
        n = L * 4 * r * D
        cp = self._lora_copies
        check(_lib.load().ftmi_lora_refresh_n(ptr(self.lora_flat[:n]), ptr(self.lora_flat[n:]), ptr(cp["lora_a_sp"]), ptr(cp["lora_bt_sp"]), ptr(cp["lora_b_ext"]),
                                               ptr(cp["lora_at_ext"]), ptr(cp["lora_at_qkv_ext"]), L, 4, r, D, stream_ptr()), "ftmi_lora_refresh_n")
        self._lora_versions = ver

    def _c_weights(self, rope) -> CogWeights:
        w = CogWeights()
        for k, v in self._stack.items():
            setattr(w, k, v.data_ptr())
        if self.lora_rank:
            self._refresh_lora_copies()
            for k, v in self._lora_copies.items():
                setattr(w, k, v.data_ptr())
        if rope is not None:
            w.rope_cos, w.rope_sin = rope[0].data_ptr(), rope[1].data_ptr()
        return w

    def _acquire_workspace(self, nbytes: int) -> torch.Tensor:
        for i, t in enumerate(self._ws_pool):
            if t.numel() >= nbytes:
                return self._ws_pool.pop(i)
        self._ws_pool.clear()
        return torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _release_workspace(self, ws: Optional[torch.Tensor]) -> None:
        if ws is not None:
            self._ws_pool.append(ws)

    def add_adapter(self, r: int = 64, lora_alpha: float = 64.0) -> None:
        """LoRA on to_q / to_k / to_v / to_out.0 of every block (the default target regex, sft_trainer/config.py:24-26).  All adapters live in ONE
        flat fp32 buffer ``lora_flat`` = [A of every block | B of every block]; the blocks' Parameters are views."""
        L, D = len(self.transformer_blocks), self.config.inner_dim
        rp = -(-int(r) // 64) * 64  # storage rank: zero-padded to a multiple of 64 (block.add_adapter); the kernels see rp, files and state dicts r
        n = L * 4 * rp * D
        self.lora_flat = torch.zeros(2 * n, dtype=torch.float32, device=self.device)
        a_all, b_all = self.lora_flat[:n].view(L, 4, rp, D), self.lora_flat[n:].view(L, 4, D, rp)
        self.lora_grad_flat = torch.zeros_like(self.lora_flat)  # laid out like lora_flat; the blocks' .grad tensors are views of it
        ga_all, gb_all = self.lora_grad_flat[:n].view(L, 4, rp, D), self.lora_grad_flat[n:].view(L, 4, D, rp)
        for i, blk in enumerate(self.transformer_blocks):
            blk.add_adapter(r, lora_alpha, a_all[i], b_all[i])
            blk._grad_a_view, blk._grad_b_view = ga_all[i], gb_all[i]
        self.lora_rank = rp
        self.lora_rank_user = int(r)

    def flat_lora_grad(self) -> torch.Tensor:
        """The flat fp32 gradient buffer the blocks' backward wrote (``blk.lora_A.grad`` / ``lora_B.grad`` are its views)."""
        return self.lora_grad_flat

    def lora_state_dict(self) -> Dict[str, torch.Tensor]:
        """peft-format keys: ``transformer_blocks.N.attn1.to_q.lora_A.weight`` ... (views of the user's rank inside the padded storage)"""
        out = {}
        r = self.lora_rank_user
        for i, blk in enumerate(self.transformer_blocks):
            for j, n in enumerate(("to_q", "to_k", "to_v", "to_out.0")):
                out[f"transformer_blocks.{i}.attn1.{n}.lora_A.weight"] = blk.lora_A[j, :r]
                out[f"transformer_blocks.{i}.attn1.{n}.lora_B.weight"] = blk.lora_B[j, :, :r]
        return out

    @torch.no_grad()
    def load_lora_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        sd = {k.replace(".default.", "."): v for k, v in sd.items()}
        for k, v in self.lora_state_dict().items():
            v.copy_(sd[k].to(v))

    def lora_parameters(self) -> List[nn.Parameter]:
        return [p for blk in self.transformer_blocks for p in (blk.lora_A, blk.lora_B) if p is not None]

    def _pos_table(self, frames: int, height: int, width: int) -> torch.Tensor:
        key = (frames, height, width)
        if key not in self._pos_cache:
            c = self.config
            tab = torch.zeros(1, c.max_text_seq_length + frames * (height // c.patch_size) * (width // c.patch_size), c.inner_dim)
            tab[0, c.max_text_seq_length:] = sincos_position_table(c, height, width, frames)
            self._pos_cache[key] = tab.to(device=self.device, dtype=bf16)
        return self._pos_cache[key]

    @torch.no_grad()
    def _embed(self, hidden_states, encoder_hidden_states, timestep, ofs=None):
        c = self.config
        B, F_, C, H, W = hidden_states.shape
        T, D, p, pt = c.max_text_seq_length, c.inner_dim, c.patch_size, c.patch_size_t
        if encoder_hidden_states.shape[1] != T:
            raise ValueError(f"CogVideoX expects {T} text tokens (max_text_seq_length), got {encoder_hidden_states.shape[1]}")
        if pt is not None and F_ % pt:
            raise ValueError(f"CogVideoX 1.5: the latent frame count {F_} must be a multiple of patch_size_t = {pt} (the specification pads it: _pad_frames)")
        S = (F_ // (pt or 1)) * (H // p) * (W // p)
        tokens = torch.empty((B, T + S, D), dtype=bf16, device=self.device)
        patches = ops.cog_patchify(hidden_states.to(bf16), p) if pt is None else patches_3d(hidden_states.to(bf16), p, pt)
        text = encoder_hidden_states.to(bf16).contiguous()
        if pt is not None and not c.use_rotary_positional_embeddings:
            raise ValueError("CogVideoX 1.5 checkpoints use rotary position embeddings")
        pos = None if c.use_rotary_positional_embeddings else self._pos_table(F_, H, W)
        for b in range(B):
            ops.gemm_nt(text[b], self.text_w, self.text_b, out=tokens[b, :T])
            ops.gemm_nt(patches[b], self.patch_w, self.patch_b, out=tokens[b, T:])
            if pos is not None:
                ops.cog_gate_residual(pos, tokens[b:b + 1], self._ones_row, 0, out=tokens[b:b + 1])  # + sincos table (text rows: + 0)
        t_emb = timestep_embedding(timestep.to(self.device), D).to(bf16)
        emb = ops.gemm_nt(torch.nn.functional.silu(ops.gemm_nt(t_emb, self.time1_w, self.time1_b)), self.time2_w, self.time2_b)
        if c.ofs_embed_dim is not None:  # [upstream] emb = emb + ofs_embedding(ofs_proj(ofs))
            o_emb = timestep_embedding(ofs.to(self.device), c.ofs_embed_dim).to(bf16)
            emb = emb + ops.gemm_nt(torch.nn.functional.silu(ops.gemm_nt(o_emb, self.ofs1_w, self.ofs1_b)), self.ofs2_w, self.ofs2_b)
        mod = ops.gemm_nt(torch.nn.functional.silu(emb), self.norm_out_lin_w, self.norm_out_lin_b)  # AdaLayerNorm: shift, scale = chunk(2)
        return tokens, emb, (1 + mod[:, D:]).contiguous(), mod[:, :D].contiguous()

    def forward(self, hidden_states, encoder_hidden_states, timestep, image_rotary_emb=None, ofs=None, return_dict: bool = False, **kwargs):
        c = self.config
        if (ofs is not None) != (c.ofs_embed_dim is not None):
            raise ValueError("ofs must be given exactly for the checkpoints with an ofs embedding (CogVideoX 1.5: ofs_embed_dim)")
        if c.use_rotary_positional_embeddings != (image_rotary_emb is not None):
            raise ValueError("image_rotary_emb must be given exactly for the rotary checkpoints (use_rotary_positional_embeddings)")
        if image_rotary_emb is not None:
            image_rotary_emb = tuple(t.to(device=self.device, dtype=torch.float32).contiguous() for t in image_rotary_emb)
        if self.proj_out_w_t is None:
            raise RuntimeError("load_diffusers_state_dict first")
        tokens, emb, onep_out, shift_out = self._embed(hidden_states, encoder_hidden_states, timestep, ofs)
        T = self.config.max_text_seq_length
        if self.native_blocks and self.lora_flat is not None:
            temb_silu = torch.nn.functional.silu(emb.to(bf16)).contiguous()
            # the last argument only tells autograd that this node has trainable inputs: the gradients go straight into lora_grad_flat
            tokens = _BlockStackFunction.apply(self, tokens, temb_silu, image_rotary_emb, self.transformer_blocks[0].lora_A, torch.is_grad_enabled())
        else:
            for blk in self.transformer_blocks:
                tokens = blk(tokens, emb, T, image_rotary_emb)
        B, F_, C, H, W = hidden_states.shape
        vel = _HeadFunction.apply(self, tokens, onep_out, shift_out, (F_, H, W))
        return {"sample": vel} if return_dict else (vel,)
