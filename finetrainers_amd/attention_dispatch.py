"""Attention-provider plugin for the MI355X kernels.

Mirror of the reference's operator plugin point (finetrainers/models/attention_dispatch.py): the same
registry class, decorator, context manager and ``attention_dispatch(query, key, value, attn_mask, dropout_p,
is_causal, scale, enable_gqa, attention_kwargs)`` entry, with ONE provider registered -- ``mi355x`` -- that runs
``ftmi_attn_fwd`` / ``ftmi_attn_bwd``.  The reference's other providers (flash / sage / xformers / flex / ATen
variants) are deliberately absent: no multi-backend dispatch.  With the reference installed, the same function is
registered into ITS registry by ``register_into_finetrainers()``, after which
``--attn_provider_training transformer:mi355x`` selects it for an unmodified diffusers model.

Layout contract (docs/models/attention.md:118): query/key/value ``[B, heads, S, head_dim]``, any strides with a
contiguous head_dim; returns ``[B, heads, S_q, head_dim]`` in the input dtype; differentiable.
"""

from __future__ import annotations

import contextlib
import inspect
import os
from enum import Enum
from typing import Any, Callable, Dict, List, Optional

import torch

from . import ops

# finetrainers/constants.py:7-8
FINETRAINERS_ATTN_PROVIDER = os.environ.get("FINETRAINERS_ATTN_PROVIDER", "mi355x")
FINETRAINERS_ATTN_CHECKS = os.getenv("FINETRAINERS_ATTN_CHECKS", "0") in {"1", "ON", "YES", "TRUE"}


class AttentionProvider(str, Enum):
    MI355X = "mi355x"


class _AttentionProviderRegistry:
    """attention_dispatch.py:295-362 (context-parallel attributes omitted: CP is not on the DP path)."""

    _providers: Dict[AttentionProvider, Callable] = {}
    _constraints: Dict[AttentionProvider, List[Callable]] = {}
    _supports_cp: Dict[AttentionProvider, bool] = {}
    _supported_arg_names: Dict[AttentionProvider, set] = {}

    _active_provider = AttentionProvider(FINETRAINERS_ATTN_PROVIDER)
    _checks_enabled = FINETRAINERS_ATTN_CHECKS

    @classmethod
    def register(cls, provider: AttentionProvider, constraints: Optional[List[Callable]] = None, supports_cp: bool = False):
        def decorator(func):
            cls._providers[provider] = func
            cls._constraints[provider] = constraints or []
            cls._supports_cp[provider] = supports_cp
            cls._supported_arg_names[provider] = set(inspect.signature(func).parameters.keys())
            return func

        return decorator

    @classmethod
    def get_active_provider(cls):
        return cls._active_provider, cls._providers[cls._active_provider]

    @classmethod
    def list_providers(cls):
        return list(cls._providers.keys())

    @classmethod
    def supports_context_parallel(cls, provider: AttentionProvider):
        if provider not in cls._providers:
            raise ValueError(f"Provider {provider} is not registered.")
        return cls._supports_cp.get(provider, False)


@contextlib.contextmanager
def attention_provider(provider: AttentionProvider = AttentionProvider.MI355X, *, mesh=None, convert_to_fp32: bool = True,
                       rotate_method: str = "allgather"):
    """attention_dispatch.py:365-402."""
    if provider not in _AttentionProviderRegistry._providers:
        raise ValueError(f"Provider {provider} is not registered.")
    if mesh is not None and not _AttentionProviderRegistry.supports_context_parallel(provider):
        raise ValueError(f"Provider {provider} does not support context parallelism.")
    old = _AttentionProviderRegistry._active_provider
    _AttentionProviderRegistry._active_provider = provider
    try:
        yield
    finally:
        _AttentionProviderRegistry._active_provider = old


def attention_dispatch(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_mask: Optional[torch.Tensor] = None,
                       dropout_p: float = 0.0, is_causal: bool = False, scale: Optional[float] = None, enable_gqa: bool = False,
                       attention_kwargs: Optional[Dict[str, Any]] = None) -> torch.Tensor:
    """attention_dispatch.py:405-447."""
    attention_kwargs = attention_kwargs or {}
    provider_name, provider_fn = _AttentionProviderRegistry.get_active_provider()
    kwargs = {"query": query, "key": key, "value": value, "attn_mask": attn_mask, "dropout_p": dropout_p, "is_causal": is_causal,
              "scale": scale, "enable_gqa": enable_gqa, **attention_kwargs}
    if _AttentionProviderRegistry._checks_enabled:
        for check in _AttentionProviderRegistry._constraints.get(provider_name):
            check(**kwargs)
    kwargs = {k: v for k, v in kwargs.items() if k in _AttentionProviderRegistry._supported_arg_names[provider_name]}
    return provider_fn(**kwargs)


# ---- constraint helpers (same style as attention_dispatch.py:460-519: raise ValueError) ------------------------


def _check_device_gpu(query, key, value, **kwargs) -> None:
    if not (query.is_cuda and key.is_cuda and value.is_cuda):
        raise ValueError("Query, key, and value must be on the GPU for the mi355x provider.")


def _check_qkv_dtype_bf16(query, key, value, **kwargs) -> None:
    if not (query.dtype == key.dtype == value.dtype == torch.bfloat16):
        raise ValueError("Query, key, and value must be bfloat16 for the mi355x provider.")


def _check_head_dim(query, key, value, **kwargs) -> None:
    d = query.shape[-1]
    if d not in (64, 128) or key.shape[-1] != d or value.shape[-1] != d:
        raise ValueError("The mi355x provider is built for head_dim 64 (LTX-Video, CogVideoX) and 128 (Wan, HunyuanVideo).")


def _check_no_dropout_causal_gqa(dropout_p=0.0, is_causal=False, enable_gqa=False, **kwargs) -> None:
    if dropout_p != 0.0 or is_causal or enable_gqa:
        raise ValueError("The mi355x provider supports non-causal attention without dropout or GQA.")


# -inf (and bool masks, which mean -inf) travel as a large finite negative: exp2 of it is exactly 0 for every kept key, and a row whose
# keys are ALL masked stays finite (uniform over the masked keys) instead of producing NaN in a training step
_MASKED = -1.0e30


def _key_bias_from_mask(attn_mask: Optional[torch.Tensor], B: int, H: int, Sk: int) -> Optional[torch.Tensor]:
    """``attn_mask`` in torch SDPA's convention -- bool (True = keep) or additive float, broadcastable to [B, H, S_q, S_k] with
    right-aligned dimensions -- whose value does not depend on the query (LTX's text mask is [B, H, 1, T]) -> fp32 key
    bias [B, S_k] (one row per sample) or [B, H, S_k] (the mask differs between heads).  Anything else raises ``ValueError``."""
    if attn_mask is None:
        return None
    m = attn_mask
    if m.dim() > 4:
        raise ValueError("mi355x provider: attn_mask has more than 4 dimensions")
    while m.dim() < 4:  # torch aligns mask dimensions to the right: [L, S] -> [1, 1, L, S], [X, L, S] -> [1, X, L, S] (X = heads)
        m = m.unsqueeze(0)
    if m.shape[-1] != Sk or m.shape[2] != 1 or m.shape[0] not in (1, B) or m.shape[1] not in (1, H):
        raise ValueError(f"mi355x provider: attn_mask {tuple(attn_mask.shape)} must broadcast to [B={B}, H={H}, 1, S_k={Sk}] (no per-query masks)")
    if m.shape[1] == 1 or m.stride(1) == 0:  # one mask for all heads (also an expanded view of one)
        m = m[:, :1]
    if m.dtype == torch.bool:
        m = torch.zeros(m.shape, dtype=torch.float32, device=m.device).masked_fill(~m, _MASKED)
    else:
        m = m.float().clamp_min(_MASKED)
    m = m[:, :, 0]  # [B|1, H|1, Sk]
    if m.shape[1] == 1:
        return m[:, 0].expand(B, Sk).contiguous()
    return m.expand(B, H, Sk).contiguous()  # a mask materialised per head is honoured per head, not reduced to head 0


class _MI355XAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, key, value, key_bias, scale):
        out, lse = ops.attn_fwd(query, key, value, key_bias, scale)
        ctx.save_for_backward(query, key, value, out, lse, key_bias if key_bias is not None else torch.empty(0, device=query.device))
        ctx.has_bias = key_bias is not None
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dout):
        query, key, value, out, lse, key_bias = ctx.saved_tensors
        dq, dk, dv = ops.attn_bwd(query, key, value, out, lse, dout, key_bias if ctx.has_bias else None, ctx.scale)
        return dq, dk, dv, None, None


@_AttentionProviderRegistry.register(
    AttentionProvider.MI355X,
    constraints=[_check_device_gpu, _check_qkv_dtype_bf16, _check_head_dim, _check_no_dropout_causal_gqa],
    supports_cp=False,
)
def _mi355x_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, attn_mask: Optional[torch.Tensor] = None,
                      dropout_p: float = 0.0, is_causal: bool = False, scale: Optional[float] = None, enable_gqa: bool = False) -> torch.Tensor:
    if dropout_p != 0.0 or is_causal or enable_gqa:
        raise ValueError("mi355x provider: dropout, causal masking and GQA are not supported")
    B, H, _, _ = query.shape
    key_bias = _key_bias_from_mask(attn_mask, B, H, key.shape[2])
    return _MI355XAttention.apply(query, key, value, key_bias, scale)


def register_into_finetrainers() -> bool:
    """Register the provider into the reference's own registry when finetrainers is importable (it needs an
    ``AttentionProvider`` enum member named MI355X = "mi355x").  Returns False when the reference is
    not installed (this container) -- the mirror registry above is then the only one."""
    try:
        from finetrainers.models import attention_dispatch as ref  # type: ignore
    except Exception:
        return False
    member = getattr(ref.AttentionProvider, "MI355X", None)
    if member is None:
        raise RuntimeError('finetrainers.models.attention_dispatch.AttentionProvider lacks MI355X = "mi355x" (INTEGRATION.md step 1)')
    ref._AttentionProviderRegistry.register(member, constraints=[_check_device_gpu, _check_qkv_dtype_bf16, _check_head_dim], supports_cp=False)(
        _mi355x_attention
    )
    return True
