"""Wan-T2V transformer for FULL fine-tuning on the MI355X (BASELINE config 4) with the call contract of the diffusers model the reference
drives (finetrainers/models/wan/base_specification.py:476-487): ``forward(hidden_states [B, C, F, H, W], timestep [B], encoder_hidden_states [B, T, 4096])``
-> ``(velocity [B, C, F, H, W],)``.

Parameters live in flat bf16 buffers -- one per block (block.py) and one ``root`` buffer for everything outside the blocks (patch embedding, condition
embedder, output table and projection) -- exactly the units FSDP-2 shards in the reference (parallel/ptd.py:466-499: ``fully_shard`` per block, then the
model).  Gradients are flat fp32 buffers of the same layouts, written by the backward kernels; nothing goes through ``.grad``.

Restated in oracle/wan.py ([upstream] diffusers transformer_wan.py + the reference's patched condition embedder, patches/models/wan/patch.py:17-33)."""

from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import ops
from ..cogvideox.model import timestep_embedding
from .block import MI355XWanBlock

bf16 = torch.bfloat16


@dataclass
class WanTransformerConfig:
    """diffusers ``WanTransformer3DModel`` config keys; defaults = Wan2.1-T2V-1.3B."""

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
    qk_norm: str = "rms_norm_across_heads"
    eps: float = 1e-6
    image_dim: Optional[int] = None
    rope_max_seq_len: int = 1024

    @property
    def inner_dim(self) -> int:
        return self.num_attention_heads * self.attention_head_dim

    @classmethod
    def from_dict(cls, d: Dict) -> "WanTransformerConfig":
        known = {k: d[k] for k in cls.__dataclass_fields__ if k in d}
        if "patch_size" in known:
            known["patch_size"] = tuple(known["patch_size"])
        return cls(**known)


def rotary_tables(cfg: WanTransformerConfig, frames: int, height: int, width: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """``WanRotaryPosEmbed`` for LATENT sizes: (cos, sin) fp32 [F' H' W', head_dim / 2]; the head's complex pairs split t : h : w, position tables in
    float64 like the reference (get_1d_rotary_pos_embed(freqs_dtype=float64))."""
    pt, ph, pw = cfg.patch_size
    f, h, w = frames // pt, height // ph, width // pw
    d = cfg.attention_head_dim
    hw = 2 * (d // 6)
    dims = (d - 2 * hw, hw, hw)

    def ang(dim, n):
        freq = 1.0 / (10000.0 ** (torch.arange(0, dim, 2, dtype=torch.float64)[: dim // 2] / dim))
        return torch.outer(torch.arange(n, dtype=torch.float64), freq)

    at = ang(dims[0], f).view(f, 1, 1, -1).expand(f, h, w, -1)
    ah = ang(dims[1], h).view(1, h, 1, -1).expand(f, h, w, -1)
    aw = ang(dims[2], w).view(1, 1, w, -1).expand(f, h, w, -1)
    a = torch.cat([at, ah, aw], dim=-1).reshape(f * h * w, d // 2)
    return torch.cos(a).float().contiguous(), torch.sin(a).float().contiguous()


class RootLayout:
    """Parameters outside the blocks, diffusers names, in one flat buffer."""

    def __init__(self, cfg: WanTransformerConfig):
        D = cfg.inner_dim
        pt, ph, pw = cfg.patch_size
        pk = cfg.in_channels * pt * ph * pw
        po = cfg.out_channels * pt * ph * pw
        if pk % 64 != 0 or po % 64 != 0:
            raise ValueError("patch embedding / output projection widths must be multiples of 64 for the GEMM")
        self.entries: List[Tuple[str, Tuple[int, ...]]] = [
            ("patch_embedding.weight", (D, pk)), ("patch_embedding.bias", (D,)),
            ("condition_embedder.time_embedder.linear_1.weight", (D, cfg.freq_dim)), ("condition_embedder.time_embedder.linear_1.bias", (D,)),
            ("condition_embedder.time_embedder.linear_2.weight", (D, D)), ("condition_embedder.time_embedder.linear_2.bias", (D,)),
            ("condition_embedder.time_proj.weight", (6 * D, D)), ("condition_embedder.time_proj.bias", (6 * D,)),
            ("condition_embedder.text_embedder.linear_1.weight", (D, cfg.text_dim)), ("condition_embedder.text_embedder.linear_1.bias", (D,)),
            ("condition_embedder.text_embedder.linear_2.weight", (D, D)), ("condition_embedder.text_embedder.linear_2.bias", (D,)),
            ("scale_shift_table", (1, 2, D)),
            ("proj_out.weight", (po, D)), ("proj_out.bias", (po,)),
        ]
        self.offsets: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = 0
        for name, shape in self.entries:
            self.offsets[name] = (off, shape)
            off += (math.prod(shape) + 63) // 64 * 64
        self.total = off

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        off, shape = self.offsets[name]
        return flat[off:off + math.prod(shape)].view(shape)

    def named_views(self, flat: torch.Tensor) -> Dict[str, torch.Tensor]:
        return {n: self.view(flat, n) for n, _ in self.entries}


class _LinearFunction(torch.autograd.Function):
    """y = act(x W^T + b) with trainable W, b: MFMA GEMM forward (optional GELU-tanh epilogue), input gradient as an NT GEMM against W^T, weight gradient
    with the token-reduction GEMM (dW += dY^T X, fp32, straight into the flat gradient buffer), bias gradient as column sums."""

    @staticmethod
    def forward(ctx, x, w, b, gw, gb, gelu: bool, need_dx: bool, anchor):  # anchor: a requires-grad dummy so that autograd visits layers fed by data
        x2d = x.reshape(-1, x.shape[-1])
        if gelu:
            y, pre = ops.gemm_nt(x2d, w, b, epilogue=1, want_out2=True)
        else:
            y, pre = ops.gemm_nt(x2d, w, b), None
        ctx.args = (x2d, w, gw, gb, pre, need_dx, x.shape)
        return y.view(*x.shape[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2d, w, gw, gb, pre, need_dx, xshape = ctx.args
        ctx.args = None
        dy2d = dy.reshape(-1, dy.shape[-1]).contiguous()
        if pre is not None:  # through the GELU: dz = dy * gelu'(pre), with the identity "GEMM" folded away -- elementwise on a [rows, N] tensor
            dy2d = _dgelu(dy2d, pre)
        ops.gemm_tn(dy2d, x2d, out=gw)
        ops.wan_colsum(dy2d, gb)
        dx = ops.gemm_nt(dy2d, ops.transpose_bf16(w), None).view(xshape) if need_dx else None
        return dx, None, None, None, None, None, None, None


def _dgelu(dy: torch.Tensor, pre: torch.Tensor) -> torch.Tensor:
    """dy * d/dz gelu_tanh(z) in fp32, one bf16 rounding (torch's GeluBackward on bf16 tensors).  Only the text embedder's [B T, D] tensor comes through here."""
    z = pre.float()
    k0, k1 = 0.7978845608028654, 0.044715
    u = k0 * (z + k1 * z * z * z)
    t = torch.tanh(u)
    dg = 0.5 * (1 + t) + 0.5 * z * (1 - t * t) * k0 * (1 + 3 * k1 * z * z)
    return (dy.float() * dg).to(bf16)


class _LnModFunction(torch.autograd.Function):
    """y = bf(LN(float(x)) * (1 + scale_b) + shift_b) with fp32 [B, D] shift / scale (the output norm); returns gradients for x, shift and scale."""

    @staticmethod
    def forward(ctx, x, shift, scale, eps: float):
        B, S, D = x.shape
        shift, scale = shift.contiguous(), scale.contiguous()
        y = ops.wan_ln(x.view(B * S, D), S, shift=shift, scale=scale, eps=eps)
        ctx.save_for_backward(x, scale)
        ctx.eps = eps
        return y.view(B, S, D)

    @staticmethod
    def backward(ctx, dy):
        x, scale = ctx.saved_tensors
        B, S, D = x.shape
        red = torch.zeros((2, B, D), dtype=torch.float32, device=x.device)
        dx = ops.wan_ln_bwd(x.view(B * S, D), dy.contiguous().view(B * S, D), S, scale=scale, eps=ctx.eps, red1=red[0], red2=red[1], red_per_batch=True)
        return dx.view(B, S, D), red[0], red[1], None


class MI355XWanTransformer3DModel(nn.Module):
    def __init__(self, config: Optional[WanTransformerConfig] = None, device: Optional[torch.device] = None):
        super().__init__()
        self.config = c = config or WanTransformerConfig()
        if c.attention_head_dim != 128 or c.image_dim is not None or c.qk_norm != "rms_norm_across_heads" or not c.cross_attn_norm:
            raise ValueError("this path covers the T2V architecture: heads of 128, RMSNorm across heads, cross_attn_norm, no image branch")
        dev = device or torch.device("cuda", 0)
        self.root_layout = RootLayout(c)
        self.root = nn.Parameter(torch.zeros(self.root_layout.total, dtype=bf16, device=dev), requires_grad=False)
        self.root_grad: Optional[torch.Tensor] = None
        self._root_src: Optional[torch.Tensor] = None  # sharded training: the all-gathered root parameters
        self.blocks = nn.ModuleList([MI355XWanBlock(c.inner_dim, c.num_attention_heads, c.ffn_dim, c.eps, dev) for _ in range(c.num_layers)])
        self._rope_cache: Dict[Tuple[int, int, int], Tuple[torch.Tensor, torch.Tensor]] = {}
        self._anchor = torch.zeros(1, dtype=bf16, device=dev, requires_grad=True)  # tells autograd that the graph has trainable inputs

    @property
    def device(self) -> torch.device:
        return self.root.device

    # -- parameters ---------------------------------------------------------------------------------------------------------------------------
    def _root_params(self) -> torch.Tensor:
        return self.root.data if self._root_src is None else self._root_src

    def rparam(self, name: str) -> torch.Tensor:
        return self.root_layout.view(self._root_params(), name)

    def rgrad(self, name: str) -> torch.Tensor:
        return self.root_layout.view(self.root_grad, name)

    def zero_grad_flat(self) -> None:
        if self.root_grad is None:
            self.root_grad = torch.zeros(self.root_layout.total, dtype=torch.float32, device=self.device)
        else:
            self.root_grad.zero_()
        for blk in self.blocks:
            blk.zero_grad_flat()

    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        """A diffusers ``WanTransformer3DModel`` state dict."""
        for name, view in self.root_layout.named_views(self.root.data).items():
            view.copy_(sd[name].to(bf16).reshape(view.shape))  # Conv3d weight [D, C, pt, ph, pw] -> [D, C pt ph pw]
        for i, blk in enumerate(self.blocks):
            pre = f"blocks.{i}."
            blk.load_diffusers_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})

    def state_dict_views(self) -> Dict[str, torch.Tensor]:
        """{diffusers parameter name: view of the flat buffers} (the patch embedding in its GEMM shape [D, C pt ph pw])."""
        if self.root.numel() < self.root_layout.total:
            raise RuntimeError("the parameters are sharded over the ranks: use the step object's gathered_state_dict()")
        out = dict(self.root_layout.named_views(self.root.data))
        for i, blk in enumerate(self.blocks):
            out.update({f"blocks.{i}.{k}": v for k, v in blk.state_dict_views().items()})
        return out

    def named_grads(self) -> Dict[str, torch.Tensor]:
        out = dict(self.root_layout.named_views(self.root_grad))
        for i, blk in enumerate(self.blocks):
            out.update({f"blocks.{i}.{k}": v for k, v in blk.named_grads().items()})
        return out

    def flat_units(self):
        """(name, parameter buffer, gradient buffer) of the shardable units: root first, then the blocks."""
        return [("root", self.root.data, self.root_grad)] + [(f"blocks.{i}", b.flat.data, b.grad_flat) for i, b in enumerate(self.blocks)]

    # -- forward ------------------------------------------------------------------------------------------------------------------------------
    def _rope(self, frames: int, height: int, width: int):
        key = (frames, height, width)
        if key not in self._rope_cache:
            self._rope_cache[key] = tuple(t.to(self.device) for t in rotary_tables(self.config, frames, height, width))
        return self._rope_cache[key]

    def _linear(self, x, name: str, gelu: bool = False, need_dx: bool = True):
        if self.root_grad is None:
            self.zero_grad_flat()
        w, b = self.rparam(f"{name}.weight"), self.rparam(f"{name}.bias")
        return _LinearFunction.apply(x, w, b, self.rgrad(f"{name}.weight"), self.rgrad(f"{name}.bias"), gelu, need_dx, self._anchor)

    def forward(self, hidden_states, timestep, encoder_hidden_states, encoder_hidden_states_image=None, return_dict: bool = False, **kwargs):
        if encoder_hidden_states_image is not None:
            raise NotImplementedError("the image-to-video branch is not part of this path")
        c = self.config
        B, C, F_, H, W = hidden_states.shape
        pt, ph, pw = c.patch_size
        f, h, w = F_ // pt, H // ph, W // pw
        D, S = c.inner_dim, f * h * w
        rope = self._rope(F_, H, W)
        # Conv3d(kernel = stride = patch) as a GEMM over the patch columns (c, pt, ph, pw), tokens in (f, h, w) order
        cols = hidden_states.to(bf16).view(B, C, f, pt, h, ph, w, pw).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(B, S, C * pt * ph * pw)
        x = self._linear(cols, "patch_embedding", need_dx=False)
        # condition embedder, the reference's patched forward (patches/models/wan/patch.py:17-33): the sinusoid takes the text dtype
        t_emb = timestep_embedding(timestep.to(self.device), c.freq_dim).to(bf16)
        temb = self._linear(torch.nn.functional.silu(self._linear(t_emb, "condition_embedder.time_embedder.linear_1", need_dx=False)),
                            "condition_embedder.time_embedder.linear_2")
        tproj = self._linear(torch.nn.functional.silu(temb), "condition_embedder.time_proj").unflatten(1, (6, -1))
        enc = self._linear(self._linear(encoder_hidden_states.to(bf16), "condition_embedder.text_embedder.linear_1", gelu=True, need_dx=False),
                           "condition_embedder.text_embedder.linear_2")
        for blk in self.blocks:
            x = blk(x, enc, tproj, rope)
        # output: (scale_shift_table + temb) in bf16, (1 + scale) in bf16, the normalised tokens in fp32
        table = self.rparam("scale_shift_table")
        mod = _TableAdd.apply(table, temb.unsqueeze(1), self.rgrad("scale_shift_table"))  # [B, 2, D] bf16
        shift, onep = mod[:, 0], 1 + mod[:, 1]
        y = _LnModFunction.apply(x, shift.float(), onep.float() - 1, c.eps)  # 1 + (onep - 1) is exact: the kernel multiplies by bf(1 + scale)
        y = self._linear(y, "proj_out")
        out = y.reshape(B, f, h, w, pt, ph, pw, -1).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, -1, F_, H, W)
        return {"sample": out} if return_dict else (out,)


class _TableAdd(torch.autograd.Function):
    """table (bf16 parameter view [1, 2, D]) + temb [B, 1, D] in bf16; the table's gradient (sum over the samples) goes to its fp32 gradient slice."""

    @staticmethod
    def forward(ctx, table, temb, gtable):
        ctx.gtable = gtable
        return table + temb

    @staticmethod
    def backward(ctx, d):
        ctx.gtable.add_(d.float().sum(0, keepdim=True))
        return None, d.sum(1, keepdim=True), None
