"""Thin tensor-level wrappers over the C ABI (include/ftmi355.h).  torch.Tensor in, torch.Tensor out;
all compute happens inside libftmi355 on the current HIP stream.  No CPU / eager fallbacks."""

from __future__ import annotations

import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import AttnDesc, check, ptr, require_gpu_tensor, stream_ptr

bf16 = torch.bfloat16


def _strides3(t: torch.Tensor) -> Tuple[int, int, int]:
    """[B, H, S, d] tensor (any strides, d contiguous) -> (batch, head, token) element strides."""
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError("attention tensors must be [B, H, S, d] with contiguous head_dim")
    return (t.stride(0), t.stride(1), t.stride(2))


def _desc(q, k, v, o, scale, dout=None, dq=None, dk=None, dv=None) -> AttnDesc:
    B, H, Sq, d = q.shape
    Sk = k.shape[2]
    if d != 64:
        raise ValueError(f"mi355x attention supports head_dim 64, got {d}")
    if k.shape != (B, H, Sk, d) or v.shape != (B, H, Sk, d):
        raise ValueError("mi355x attention: key/value shapes must be [B, H, Sk, 64] (no GQA)")
    desc = AttnDesc()
    desc.B, desc.H, desc.Sq, desc.Sk, desc.d = B, H, Sq, Sk, d
    desc.scale = float(scale)
    for name, t in (("q_strides", q), ("k_strides", k), ("v_strides", v), ("o_strides", o), ("do_strides", dout),
                    ("dq_strides", dq), ("dk_strides", dk), ("dv_strides", dv)):
        if t is not None:
            s = _strides3(t)
            arr = getattr(desc, name)
            arr[0], arr[1], arr[2] = s
    return desc


def attn_fwd(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, key_bias: Optional[torch.Tensor] = None,
             scale: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """q,k,v [B,H,S,64] bf16 (strided views allowed).  Returns (out [B,H,Sq,64] laid out as [B,Sq,H,64], lse [B,H,Sq])."""
    for n, t in (("query", q), ("key", k), ("value", v)):
        require_gpu_tensor(t, n, bf16)
    B, H, Sq, d = q.shape
    scale = (1.0 / d**0.5) if scale is None else scale
    out = torch.empty((B, Sq, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3)
    lse = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    if key_bias is not None:
        require_gpu_tensor(key_bias, "key_bias", torch.float32)
        key_bias = key_bias.contiguous()
    desc = _desc(q, k, v, out, scale)
    check(_lib.load().ftmi_attn_fwd(ctypes.byref(desc), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(key_bias), stream_ptr()), "ftmi_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, out, lse, dout, key_bias=None, scale=None):
    B, H, Sq, d = q.shape
    Sk = k.shape[2]
    scale = (1.0 / d**0.5) if scale is None else scale
    if dout.stride(3) != 1:
        dout = dout.contiguous()
    dq = torch.empty((B, Sq, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3)
    dk = torch.empty((B, Sk, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3)
    dv = torch.empty((B, Sk, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3)
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    desc = _desc(q, k, v, out, scale, dout, dq, dk, dv)
    check(_lib.load().ftmi_attn_bwd(ctypes.byref(desc), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(dout), ptr(dq), ptr(dk), ptr(dv),
                                     ptr(delta), ptr(key_bias), stream_ptr()), "ftmi_attn_bwd")
    return dq, dk, dv


def gemm_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, alpha: float = 1.0, epilogue: int = 0,
            resid: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
            aux: Optional[torch.Tensor] = None, want_out2: bool = False, variant: int = 8):
    """out[M,N] = epilogue(alpha * x[M,K] @ w[N,K]^T + bias)."""
    require_gpu_tensor(x, "x", bf16)
    require_gpu_tensor(w, "w", bf16)
    M, K = x.shape
    N = w.shape[0]
    out = torch.empty((M, N), dtype=bf16, device=x.device)
    out2 = torch.empty((M, N), dtype=bf16, device=x.device) if want_out2 else None
    check(_lib.load().ftmi_gemm_nt(M, N, K, ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), float(alpha), ptr(out), N, epilogue,
                                    ptr(out2), ptr(resid), ptr(gate), rows_per_batch, ptr(aux), variant, stream_ptr()), "ftmi_gemm_nt")
    return (out, out2) if want_out2 else out


def gemm_tn(u: torch.Tensor, v: torch.Tensor, out: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
    """out[P,Q] (fp32) += scale * u[M,P]^T @ v[M,Q]."""
    M, P = u.shape
    Q = v.shape[1]
    if out is None:
        out = torch.zeros((P, Q), dtype=torch.float32, device=u.device)
    check(_lib.load().ftmi_gemm_tn(M, P, Q, ptr(u), u.stride(0), ptr(v), v.stride(0), ptr(out), out.stride(0), float(scale), stream_ptr()),
          "ftmi_gemm_tn")
    return out


def transpose_bf16(x: torch.Tensor) -> torch.Tensor:
    rows, cols = x.shape
    out = torch.empty((cols, rows), dtype=bf16, device=x.device)
    check(_lib.load().ftmi_transpose_bf16(ptr(x), ptr(out), rows, cols, stream_ptr()), "ftmi_transpose_bf16")
    return out


def linear_lora_fwd(x, w, bias, a_bf, b_bf, lora_scale: float, variant: int = 8):
    M, K = x.shape
    N = w.shape[0]
    r = 0 if a_bf is None else a_bf.shape[0]
    y = torch.empty((M, N), dtype=bf16, device=x.device)
    xa = torch.empty((M, r), dtype=bf16, device=x.device) if r else None
    check(_lib.load().ftmi_linear_lora_fwd(M, K, N, r, float(lora_scale), ptr(x), ptr(w), ptr(bias), ptr(a_bf), ptr(b_bf), ptr(y), ptr(xa),
                                            variant, stream_ptr()), "ftmi_linear_lora_fwd")
    return y, xa


def linear_lora_bwd(x, dy, xa, w_t, a_t, b_t, lora_scale: float, grad_a=None, grad_b=None, need_dx: bool = True, variant: int = 8):
    """Backward of ``linear_lora_fwd``: returns (dx [M,K] bf16 or None, grad_a [r,K] fp32, grad_b [N,r] fp32); the gradient
    buffers are accumulated into when given (``.grad`` semantics).  ``w_t = W^T``, ``a_t = A^T``, ``b_t = B^T`` in bf16."""
    M, N = dy.shape
    K = w_t.shape[0] if w_t is not None else x.shape[1]
    r = 0 if a_t is None else a_t.shape[1]
    dx = torch.empty((M, K), dtype=bf16, device=dy.device) if need_dx else None
    dxa = torch.empty((M, r), dtype=bf16, device=dy.device) if r else None
    if r:
        grad_a = torch.zeros((r, K), dtype=torch.float32, device=dy.device) if grad_a is None else grad_a
        grad_b = torch.zeros((N, r), dtype=torch.float32, device=dy.device) if grad_b is None else grad_b
    check(_lib.load().ftmi_linear_lora_bwd(M, K, N, r, float(lora_scale), ptr(x), ptr(dy), ptr(xa), ptr(w_t), ptr(a_t), ptr(b_t), ptr(dxa), ptr(dx),
                                            ptr(grad_a), ptr(grad_b), variant, stream_ptr()), "ftmi_linear_lora_bwd")
    return dx, grad_a, grad_b


def noise_pack(latents, noise, mean, std, sigma, sigma_first=None, first_frame_tokens: int = 0):
    """latents/noise [B,C,F,H,W] bf16 -> (x_t, target) [B, F*H*W, C] bf16."""
    B, C = latents.shape[:2]
    S = latents[0, 0].numel()
    latents = latents.contiguous()
    noise = noise.contiguous()
    xt = torch.empty((B, S, C), dtype=bf16, device=latents.device)
    target = torch.empty((B, S, C), dtype=bf16, device=latents.device)
    check(_lib.load().ftmi_ltx_noise_pack(ptr(latents), ptr(noise), ptr(mean), ptr(std), ptr(sigma), ptr(sigma_first), first_frame_tokens,
                                           ptr(xt), ptr(target), B, C, S, stream_ptr()), "ftmi_ltx_noise_pack")
    return xt, target


def mse_loss(pred, target, weight: Optional[torch.Tensor], want_grad: bool = True, grad_scale: float = 1.0):
    B = pred.shape[0]
    per = pred[0].numel()
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    check(_lib.load().ftmi_mse_loss(ptr(pred), ptr(target), ptr(weight), ptr(loss), ptr(dpred), B, per, float(grad_scale), stream_ptr()),
          "ftmi_mse_loss")
    return loss, dpred


def clip_adamw_step(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                    weight_decay: float = 1e-4, max_norm: float = 1.0, scratch: Optional[torch.Tensor] = None,
                    grad_norm_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n = params.numel()
    if scratch is None:
        scratch = torch.empty((2,), dtype=torch.float32, device=params.device)
    if grad_norm_out is None:
        grad_norm_out = torch.empty((1,), dtype=torch.float32, device=params.device)
    check(_lib.load().ftmi_clip_adamw_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), n, float(max_norm), float(lr),
                                            float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), ptr(scratch),
                                            ptr(grad_norm_out), stream_ptr()), "ftmi_clip_adamw_step")
    return grad_norm_out
