"""A NARROW LTX-Video transformer run through the production kernels by zero padding.

The gfx950 kernels of this backend are built for LTX-Video's production geometry -- width 2048 = 32 heads x 64, channel counts and LoRA ranks in multiples of 64.
The reference's own trainer tests, however, run a DUMMY model (tests/models/ltx_video/base_specification.py:47-58: ``LTXVideoTransformer3DModel(in_channels=8,
out_channels=8, num_attention_heads=4, attention_head_dim=8, cross_attention_dim=32, num_layers=1, caption_channels=32)``), and a drop-in has to run what the
reference's tests run.  This module embeds such a model in the wide layout:

* residual-stream space (width ``Dv = heads * head_dim``): channel c -> wide channel c (a prefix of the 2048);
* head space (outputs of to_q / to_k / to_v, inputs of to_out, norm_q / norm_k weights, RoPE pairs): channel ``h * head_dim + j`` -> wide channel ``64 h + j``;
* feed-forward space (``ff_mult * Dv``), latent channels, caption channels, LoRA rank: prefixes of their padded sizes.

Every padded weight, bias, table entry and norm weight is an exact zero, so every padded activation and every padded gradient is an exact zero through the whole
forward and backward (zeros are preserved by GEMMs with zero rows / columns, by GELU / SiLU, by RMS norms, by attention over all-zero heads whose value rows are
zero); the two things the kernels have to be told are over how many channels a normalisation takes its mean (``ftmi_ltx_config.d_valid``; the final LayerNorm also
masks the padded channels) and the true head width for the softmax scale (``head_dim_valid``).  The result is the narrow model's arithmetic on the production
kernels -- no second code path, no CPU fallback.  It costs what the wide model costs per block; it exists for the reference's smoke tests, not for speed.
"""
import math
import re
from typing import Dict, Iterable, Optional, Tuple

import torch
from torch import nn

from .transformer import LORA_ORDER, LTXTransformerConfig, MI355XLTXVideoTransformer3DModel, bf16, ltx_rope_tables

WIDE, WIDE_HEADS, WIDE_HEAD_DIM = 2048, 32, 64


def _pad64(n: int) -> int:
    return -(-int(n) // 64) * 64


def is_native(config: LTXTransformerConfig) -> bool:
    """True where the kernels take the configuration as it is."""
    return (config.inner_dim == WIDE and config.attention_head_dim == WIDE_HEAD_DIM and config.in_channels % 64 == 0 and config.out_channels % 64 == 0
            and config.caption_channels % 64 == 0)


class NarrowLayout:
    """Index maps of a narrow configuration inside the wide layout (pure host logic: tested without a GPU)."""

    def __init__(self, config: LTXTransformerConfig):
        H, hd = config.num_attention_heads, config.attention_head_dim
        Dv = H * hd
        if config.patch_size != 1 or config.patch_size_t != 1:
            raise ValueError("patch size must be 1 (as for the production model)")
        if H > WIDE_HEADS or hd > WIDE_HEAD_DIM or hd % 2:
            raise ValueError(f"a narrow model needs at most {WIDE_HEADS} heads of an even width <= {WIDE_HEAD_DIM}, got {H} x {hd}")
        if Dv % 6 != 2:
            raise ValueError("the compact RoPE table assumes inner_dim % 6 == 2 (LTX: 2048, the reference's dummy: 32)")
        if config.cross_attention_dim != Dv:
            raise ValueError("LTX projects the caption to the transformer's width: cross_attention_dim must equal heads * head_dim")
        self.config, self.Dv, self.H, self.hd = config, Dv, H, hd
        self.ff = config.ff_mult * Dv
        self.wide = LTXTransformerConfig(in_channels=_pad64(config.in_channels), out_channels=_pad64(config.out_channels), patch_size=1, patch_size_t=1,
                                         num_attention_heads=WIDE_HEADS, attention_head_dim=WIDE_HEAD_DIM, cross_attention_dim=WIDE,
                                         num_layers=config.num_layers, caption_channels=_pad64(config.caption_channels), norm_eps=config.norm_eps,
                                         qk_norm_eps=config.qk_norm_eps, ff_mult=config.ff_mult)
        self.idx_d = torch.arange(Dv)
        self.idx_h = torch.tensor([WIDE_HEAD_DIM * h + j for h in range(H) for j in range(hd)])
        # wide RoPE pair of narrow pair p (channels 2p, 2p + 1 of the flat width: head (2p) // hd, offset (2p) % hd)
        self.idx_pair = torch.tensor([(WIDE_HEAD_DIM * ((2 * p) // hd) + (2 * p) % hd) // 2 for p in range(Dv // 2)])

    # -- spaces: "d" residual stream, "h" head space, "ff", "in", "out", "cap", "t" (the 256 sinusoid channels), "6d" / "2d" (stacked tables) ------------------
    def _index(self, space: str) -> Tuple[torch.Tensor, int]:
        c, w = self.config, self.wide
        if space == "d":
            return self.idx_d, WIDE
        if space == "h":
            return self.idx_h, WIDE
        if space == "ff":
            return torch.arange(self.ff), WIDE * c.ff_mult
        if space == "in":
            return torch.arange(c.in_channels), w.in_channels
        if space == "out":
            return torch.arange(c.out_channels), w.out_channels
        if space == "cap":
            return torch.arange(c.caption_channels), w.caption_channels
        if space == "t":
            return torch.arange(256), 256
        if space in ("6d", "2d"):
            k = int(space[0])
            return torch.cat([self.idx_d + i * WIDE for i in range(k)]), k * WIDE
        raise KeyError(space)

    def widen(self, t: torch.Tensor, *spaces: str) -> torch.Tensor:
        """Zero-padded wide image of a narrow tensor; one space name per dimension ("-" keeps a dimension as it is)."""
        out_shape, idx = [], []
        for dim, sp in enumerate(spaces):
            if sp == "-":
                out_shape.append(t.shape[dim])
                idx.append(torch.arange(t.shape[dim]))
            else:
                ix, n = self._index(sp)
                if ix.numel() != t.shape[dim]:
                    raise ValueError(f"dimension {dim} has {t.shape[dim]} entries, the {sp!r} space of this model {ix.numel()}")
                out_shape.append(n)
                idx.append(ix)
        out = torch.zeros(out_shape, dtype=t.dtype, device=t.device)
        out[torch.meshgrid(*[i.to(t.device) for i in idx], indexing="ij")] = t
        return out

    def narrow(self, t: torch.Tensor, *spaces: str) -> torch.Tensor:
        """The narrow entries of a wide tensor (inverse of ``widen``)."""
        idx = [torch.arange(t.shape[d]) if sp == "-" else self._index(sp)[0] for d, sp in enumerate(spaces)]
        return t[torch.meshgrid(*[i.to(t.device) for i in idx], indexing="ij")]

    # diffusers parameter name -> spaces per dimension
    def spaces_of(self, key: str) -> Tuple[str, ...]:
        k = re.sub(r"^transformer_blocks\.\d+\.", "", key.replace(".base_layer.", "."))
        table = {
            "proj_in.weight": ("d", "in"), "proj_in.bias": ("d",),
            "time_embed.emb.timestep_embedder.linear_1.weight": ("d", "t"), "time_embed.emb.timestep_embedder.linear_1.bias": ("d",),
            "time_embed.emb.timestep_embedder.linear_2.weight": ("d", "d"), "time_embed.emb.timestep_embedder.linear_2.bias": ("d",),
            "time_embed.linear.weight": ("6d", "d"), "time_embed.linear.bias": ("6d",),
            "caption_projection.linear_1.weight": ("d", "cap"), "caption_projection.linear_1.bias": ("d",),
            "caption_projection.linear_2.weight": ("d", "d"), "caption_projection.linear_2.bias": ("d",),
            "proj_out.weight": ("out", "d"), "proj_out.bias": ("out",),
            "ff.net.0.proj.weight": ("ff", "d"), "ff.net.0.proj.bias": ("ff",), "ff.net.2.weight": ("d", "ff"), "ff.net.2.bias": ("d",),
        }
        if k in table:
            return table[k]
        if k == "scale_shift_table":
            return ("-", "d")  # [2, D] at the top level, [6, D] in a block
        m = re.fullmatch(r"(attn[12])\.(to_q|to_k|to_v|to_out\.0|norm_q|norm_k)\.(weight|bias)", k)
        if m:
            _, mod, leaf = m.groups()
            if mod in ("norm_q", "norm_k"):
                return ("h",)
            if mod == "to_out.0":
                return ("d", "h") if leaf == "weight" else ("d",)
            return ("h", "d") if leaf == "weight" else ("h",)  # (attn2.to_k / to_v read the caption projected to the transformer's width)
        raise KeyError(f"no layout rule for parameter {key!r}")

    def lora_spaces(self, adapter: int) -> Tuple[str, str]:
        """(input space of A's columns, output space of B's rows) of adapter ``adapter`` in LORA_ORDER."""
        return ("h", "d") if LORA_ORDER[adapter].endswith("to_out.0") else ("d", "h")

    def rope_wide(self, cos: torch.Tensor, sin: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Narrow per-pair tables [S, Dv / 2] -> wide [S, 1024] (cos = 1, sin = 0 on the padded pairs: the identity rotation of a zero)."""
        S = cos.shape[0]
        wc = torch.ones(S, WIDE // 2, dtype=cos.dtype)
        ws = torch.zeros(S, WIDE // 2, dtype=sin.dtype)
        wc[:, self.idx_pair] = cos
        ws[:, self.idx_pair] = sin
        return wc.contiguous(), ws.contiguous()


class MI355XNarrowLTXVideoTransformer3DModel(nn.Module):
    """``LTXVideoTransformer3DModel`` of a geometry narrower than the production one (the reference's dummy fixture), on the production kernels."""

    def __init__(self, config: LTXTransformerConfig, device: Optional[torch.device] = None, gemm_variant: int = 8):
        super().__init__()
        self.config = config
        self.layout = NarrowLayout(config)
        self.inner = MI355XLTXVideoTransformer3DModel(self.layout.wide, device=device, gemm_variant=gemm_variant)
        self.inner._narrow = (self.layout.Dv, self.layout.hd)  # -> ftmi_ltx_config.d_valid / head_dim_valid
        self.inner.norm_q.zero_(), self.inner.norm_k.zero_(), self.inner.norm_q2.zero_(), self.inner.norm_k2.zero_()

    @property
    def device(self) -> torch.device:
        return self.inner.device

    @property
    def dtype(self) -> torch.dtype:
        return bf16

    # The engine state -- the flat LoRA storage, its gradient buffer, the gradient-exchange hooks -- lives in the wide module; the fused optimisation step
    # (finetrainers_amd.trainer.MI355XSFTStep) and the data-parallel backend read and set it through this object as they do on the production class.
    _ENGINE_STATE = ("_lora_versions", "grad_bucket_blocks", "_grad_bucket_hook", "_grad_bucket_finish", "gradient_checkpointing", "gemm_variant")

    def __getattr__(self, name: str):
        try:
            return super().__getattr__(name)
        except AttributeError:
            if name != "inner" and "inner" in self.__dict__.get("_modules", {}):
                return getattr(self._modules["inner"], name)
            raise

    def __setattr__(self, name: str, value) -> None:
        if name in self._ENGINE_STATE and "inner" in self.__dict__.get("_modules", {}):
            setattr(self._modules["inner"], name, value)
        else:
            super().__setattr__(name, value)

    # ---- weights -------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        lay = self.layout
        wide = {}
        lora = {}
        for k, v in sd.items():
            if "lora_" in k:
                lora[k] = v
                continue
            wide[k.replace(".base_layer.", ".")] = lay.widen(v.detach().to("cpu"), *lay.spaces_of(k))
        self.inner.load_diffusers_state_dict(wide)
        if lora and self.inner.lora_A is not None:
            self.load_lora_state_dict(lora)

    @torch.no_grad()
    def init_random_(self, seed: int = 0) -> "MI355XNarrowLTXVideoTransformer3DModel":
        """Random weights of the narrow architecture (nn.Linear default init over the NARROW fan-in; tables randn / sqrt(Dv); norm weights 1)."""
        lay = self.layout
        g = torch.Generator().manual_seed(seed)
        sd = {}
        for key, view in self.inner._base_views().items():
            spaces = lay.spaces_of(key)
            shape = [view.shape[d] if sp == "-" else lay._index(sp)[0].numel() for d, sp in enumerate(spaces)]
            if key.endswith("scale_shift_table"):
                sd[key] = torch.randn(shape, generator=g) / lay.Dv ** 0.5
            elif ".norm_" in key:
                sd[key] = torch.ones(shape)
            else:
                fan_in = shape[-1] if len(shape) == 2 else {"ff.net.2.bias": lay.ff, "proj_in.bias": self.config.in_channels, "caption_projection.linear_1.bias": self.config.caption_channels,
                                                           "time_embed.emb.timestep_embedder.linear_1.bias": 256}.get(re.sub(r"^transformer_blocks\.\d+\.", "", key), lay.Dv)
                sd[key] = (torch.rand(shape, generator=g) * 2 - 1) / math.sqrt(fan_in)
        self.load_diffusers_state_dict(sd)
        return self

    def add_adapter(self, adapter_config=None, adapter_name: str = "default", *, r: Optional[int] = None, lora_alpha: Optional[float] = None,
                    target_modules=None) -> None:
        """peft ``LoraConfig(r, lora_alpha, init_lora_weights=True)`` on the narrow model: A ~ kaiming-uniform(a = sqrt(5)) over the NARROW fan-in, B = 0; the padded
        rows / columns are zeros and stay zeros (exact-zero gradients, see the module docstring)."""
        self.inner.add_adapter(adapter_config, adapter_name, r=r, lora_alpha=lora_alpha, target_modules=target_modules)
        lay, r = self.layout, self.inner.lora_rank
        bound = math.sqrt(6.0 / ((1 + 5.0) * lay.Dv))
        with torch.no_grad():
            self.inner._lora_A_full.zero_()
            for i in range(8):
                cols = lay._index(lay.lora_spaces(i)[0])[0].to(self.device)
                a = torch.empty(self.config.num_layers, r, lay.Dv, device=self.device).uniform_(-bound, bound)
                self.inner._lora_A_full[:, i, :r][:, :, cols] = a
        self.inner._lora_versions = None

    @property
    def lora_A(self):
        return self.inner.lora_A

    @property
    def lora_B(self):
        return self.inner.lora_B

    def lora_views(self) -> Iterable[Tuple[str, torch.Tensor, torch.Tensor]]:
        """(module path, A [r, Dv], B [Dv, r]) of every adapter: the narrow entries (copies -- the Parameters themselves are the wide ``inner.lora_A / lora_B``)."""
        lay = self.layout
        for l in range(self.config.num_layers):
            for i, n in enumerate(LORA_ORDER):
                a_sp, b_sp = lay.lora_spaces(i)
                yield (f"transformer_blocks.{l}.{n}", lay.narrow(self.inner.lora_A[l, i], "-", a_sp), lay.narrow(self.inner.lora_B[l, i], b_sp, "-"))

    def lora_state_dict(self, adapter_name: Optional[str] = None) -> Dict[str, torch.Tensor]:
        mid = f".{adapter_name}" if adapter_name else ""
        out = {}
        for path, a, b in self.lora_views():
            out[f"{path}.lora_A{mid}.weight"] = a
            out[f"{path}.lora_B{mid}.weight"] = b
        return out

    def lora_grad_state_dict(self) -> Dict[str, torch.Tensor]:
        """The narrow entries of the LoRA gradients, peft-format keys (tests compare them with the oracle's)."""
        lay, out = self.layout, {}
        ga, gb = self.inner.lora_A.grad, self.inner.lora_B.grad
        for l in range(self.config.num_layers):
            for i, n in enumerate(LORA_ORDER):
                a_sp, b_sp = lay.lora_spaces(i)
                out[f"transformer_blocks.{l}.{n}.lora_A.weight"] = lay.narrow(ga[l, i], "-", a_sp)
                out[f"transformer_blocks.{l}.{n}.lora_B.weight"] = lay.narrow(gb[l, i], b_sp, "-")
        return out

    @torch.no_grad()
    def load_lora_state_dict(self, sd: Dict[str, torch.Tensor]) -> None:
        lay = self.layout
        sd = {k.replace(".default.", "."): v for k, v in sd.items()}
        for l in range(self.config.num_layers):
            for i, n in enumerate(LORA_ORDER):
                a_sp, b_sp = lay.lora_spaces(i)
                p = f"transformer_blocks.{l}.{n}"
                self.inner.lora_A[l, i].copy_(lay.widen(sd[f"{p}.lora_A.weight"].float().cpu(), "-", a_sp))
                self.inner.lora_B[l, i].copy_(lay.widen(sd[f"{p}.lora_B.weight"].float().cpu(), b_sp, "-"))
        self.inner._lora_versions = None

    # ---- the reference's save / resume paths (trainer.py:279-306, parallel/ptd.py:313-321) read and write state dicts: NARROW shapes, the production class's keys ----
    def state_dict(self, *args, destination=None, prefix: str = "", keep_vars: bool = False, **kwargs) -> Dict[str, torch.Tensor]:
        lay = self.layout
        out = destination if destination is not None else {}
        for k, v in self.inner._base_views().items():
            out[prefix + k] = lay.narrow(v.detach(), *lay.spaces_of(k))
        if self.inner.lora_A is not None:
            for k, v in self.lora_state_dict(adapter_name="default").items():
                out[prefix + k] = v.detach()
        return out

    def load_state_dict(self, state_dict: Dict[str, torch.Tensor], strict: bool = True, assign: bool = False):
        if assign:
            raise ValueError("load_state_dict(assign=True) is not supported: the parameters are views of one flat buffer of the wide layout")
        sd = {k.replace(".base_layer.", "."): v for k, v in state_dict.items()}
        base = {k: v for k, v in sd.items() if "lora_" not in k}
        lora = {k: v for k, v in sd.items() if "lora_" in k}
        missing = []
        if base:
            want = {k.replace(".base_layer.", ".") for k in self.inner._base_views()}
            missing = sorted(want - set(base))
            if strict and (missing or set(base) - want):
                raise RuntimeError(f"load_state_dict: missing {missing[:5]}, unexpected {sorted(set(base) - want)[:5]}")
            if not missing:
                self.load_diffusers_state_dict(base)
        if lora:
            if self.inner.lora_A is None:
                raise RuntimeError("load_state_dict: LoRA tensors given but no adapter attached (call add_adapter first)")
            self.load_lora_state_dict(lora)
        return nn.modules.module._IncompatibleKeys(missing, [])

    def enable_gradient_checkpointing(self) -> None:
        self.inner.enable_gradient_checkpointing()

    def disable_gradient_checkpointing(self) -> None:
        self.inner.disable_gradient_checkpointing()

    # ---- forward (patch.py:38-50 signature) ----------------------------------------------------------------------------------------------
    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor, timestep: torch.Tensor,
                encoder_attention_mask: Optional[torch.Tensor], num_frames: int, height: int, width: int,
                rope_interpolation_scale=None, return_dict: bool = True, *args, **kwargs):
        lay, c, w = self.layout, self.config, self.layout.wide
        if hidden_states.shape[-1] != c.in_channels or encoder_hidden_states.shape[-1] != c.caption_channels:
            raise ValueError("channel counts do not match the configuration")
        x = torch.nn.functional.pad(hidden_states.to(bf16), (0, w.in_channels - c.in_channels))
        text = torch.nn.functional.pad(encoder_hidden_states.to(bf16), (0, w.caption_channels - c.caption_channels))
        key = (num_frames, height, width, None if rope_interpolation_scale is None else tuple(float(v) for v in rope_interpolation_scale))
        if key not in self.inner._rope_cache:
            cos, sin = ltx_rope_tables(num_frames, height, width, rope_interpolation_scale, dim=lay.Dv)
            wc, ws = lay.rope_wide(cos, sin)
            self.inner._rope_cache[key] = (wc.to(self.device), ws.to(self.device))
        out = self.inner(x, text, timestep, encoder_attention_mask, num_frames, height, width, rope_interpolation_scale, return_dict=False)[0]
        out = out[..., : c.out_channels].contiguous()  # (consumers hand the prediction to kernels that take dense rows)
        if not return_dict:
            return (out,)
        return {"sample": out}


def build_ltx_transformer(config: LTXTransformerConfig, device: Optional[torch.device] = None, gemm_variant: int = 8) -> nn.Module:
    """The module for ``config``: the production class where the kernels take the geometry as it is, the zero-padded embedding otherwise."""
    if is_native(config):
        return MI355XLTXVideoTransformer3DModel(config, device=device, gemm_variant=gemm_variant)
    return MI355XNarrowLTXVideoTransformer3DModel(config, device=device, gemm_variant=gemm_variant)
