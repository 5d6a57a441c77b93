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


def _desc(q, k, v, o, scale, dout=None, dq=None, dk=None, dv=None, key_bias=None) -> AttnDesc:
    B, H, Sq, d = q.shape
    Sk = k.shape[2]
    if d not in (64, 128):
        raise ValueError(f"mi355x attention supports head_dim 64 and 128, got {d}")
    if k.shape != (B, H, Sk, d) or v.shape != (B, H, Sk, d):
        raise ValueError("mi355x attention: key/value shapes must be [B, H, Sk, head_dim] (no GQA)")
    desc = AttnDesc()
    desc.B, desc.H, desc.Sq, desc.Sk, desc.d = B, H, Sq, Sk, d
    desc.scale = float(scale)
    if key_bias is not None:
        if key_bias.shape not in ((B, Sk), (B, H, Sk)) or key_bias.stride(-1) != 1:
            raise ValueError("key_bias must be fp32 [B, Sk] or [B, H, Sk] with contiguous keys")
        desc.bias_strides[0] = key_bias.stride(0)
        desc.bias_strides[1] = key_bias.stride(1) if key_bias.dim() == 3 else 0
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
    desc = _desc(q, k, v, out, scale, key_bias=key_bias)
    check(_lib.load().ftmi_attn_fwd(ctypes.byref(desc), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(key_bias), stream_ptr()), "ftmi_attn_fwd")
    return out, lse


def attn_bwd(q, k, v, out, lse, dout, key_bias=None, scale=None, dv_out=None):
    """``dv_out``: an existing [B, H, Sk, d] bf16 view (head_dim contiguous) that receives dV, e.g. the value third of a fused d(q|k|v) buffer."""
    B, H, Sq, d = q.shape
    Sk = k.shape[2]
    scale = (1.0 / d**0.5) if scale is None else scale
    if dout.stride(3) != 1:
        dout = dout.contiguous()
    dq = torch.empty((B, Sq, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3)
    dk = torch.empty((B, Sk, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3)
    dv = torch.empty((B, Sk, H, d), dtype=bf16, device=q.device).permute(0, 2, 1, 3) if dv_out is None else dv_out
    if dv.shape != (B, H, Sk, d) or dv.dtype != bf16:
        raise ValueError("attn_bwd: dv_out must be a [B, H, Sk, head_dim] bf16 view")
    delta = torch.empty((B, H, Sq), dtype=torch.float32, device=q.device)
    desc = _desc(q, k, v, out, scale, dout, dq, dk, dv, key_bias=key_bias)
    check(_lib.load().ftmi_attn_bwd(ctypes.byref(desc), ptr(q), ptr(k), ptr(v), ptr(out), ptr(lse), ptr(dout), ptr(dq), ptr(dk), ptr(dv),
                                     ptr(delta), ptr(key_bias), stream_ptr()), "ftmi_attn_bwd")
    return dq, dk, dv


def gemm_nt(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None, alpha: float = 1.0, epilogue: int = 0,
            resid: Optional[torch.Tensor] = None, gate: Optional[torch.Tensor] = None, rows_per_batch: int = 0,
            aux: Optional[torch.Tensor] = None, want_out2: bool = False, variant: int = 8, out: Optional[torch.Tensor] = None):
    """out[M,N] = epilogue(alpha * x[M,K] @ w[N,K]^T + bias).  ``out``: write into an existing [M, N] bf16 view (row stride free, a multiple of 8)."""
    require_gpu_tensor(x, "x", bf16)
    require_gpu_tensor(w, "w", bf16)
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty((M, N), dtype=bf16, device=x.device)
    elif out.shape != (M, N) or out.dtype != bf16 or out.stride(1) != 1:
        raise ValueError("gemm_nt: out must be an [M, N] bf16 view with contiguous columns")
    # out2 (allocated here, contiguous) / resid / aux share ONE row stride, which need not be out's (include/ftmi355.h: ld_side)
    out2 = torch.empty((M, N), dtype=bf16, device=x.device) if want_out2 else None
    side = [t for t in (out2, resid, aux) if t is not None]
    for t in side:
        if t.shape != (M, N) or t.stride(1) != 1 or t.stride(0) != side[0].stride(0):
            raise ValueError("gemm_nt: out2 / resid / aux must be [M, N] bf16 views with contiguous columns and one common row stride")
    ld_side = side[0].stride(0) if side else 0
    check(_lib.load().ftmi_gemm_nt(M, N, K, ptr(x), x.stride(0), ptr(w), w.stride(0), ptr(bias), float(alpha), ptr(out), out.stride(0), epilogue,
                                    ptr(out2), ptr(resid), ptr(gate), rows_per_batch, ptr(aux), ld_side, variant, stream_ptr()), "ftmi_gemm_nt")
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


def fp8_upcast(w8: torch.Tensor, out: Optional[torch.Tensor] = None, transpose: bool = False) -> torch.Tensor:
    """bf16 copy of a [rows, cols] ``torch.float8_e4m3fn`` weight (exact), optionally transposed to [cols, rows]; ``out``: a contiguous bf16 tensor of the
    result's size (e.g. a slice of a per-model weight arena)."""
    if w8.dtype != torch.float8_e4m3fn or w8.dim() != 2 or not w8.is_contiguous() or not w8.is_cuda:
        raise ValueError("fp8_upcast: a contiguous 2-D float8_e4m3fn tensor on the GPU is required")
    rows, cols = w8.shape
    shape = (cols, rows) if transpose else (rows, cols)
    if out is None:
        out = torch.empty(shape, dtype=bf16, device=w8.device)
    elif out.numel() != rows * cols or out.dtype != bf16 or not out.is_contiguous():
        raise ValueError("fp8_upcast: out must be a contiguous bf16 tensor with rows * cols elements")
    check(_lib.load().ftmi_fp8_upcast(ptr(w8), ptr(out), rows, cols, int(transpose), stream_ptr()), "ftmi_fp8_upcast")
    return out.view(shape)


def transpose_bf16(x: torch.Tensor) -> torch.Tensor:
    rows, cols = x.shape
    out = torch.empty((cols, rows), dtype=bf16, device=x.device)
    check(_lib.load().ftmi_transpose_bf16(ptr(x), ptr(out), rows, cols, stream_ptr()), "ftmi_transpose_bf16")
    return out


def lora_split(w: torch.Tensor, sp: bool = False, ext: bool = False, t_sp: bool = False, t_ext: bool = False):
    """bf16 (hi, lo) working copies of an fp32 LoRA matrix w [rows, cols] (include/ftmi355.h: ftmi_lora_split).  Returns the requested
    layouts in the order (sp [2 rows, cols], ext [rows, 3 cols], t_sp [2 cols, rows], t_ext [cols, 3 rows])."""
    require_gpu_tensor(w, "w", torch.float32)
    w = w.contiguous()
    rows, cols = w.shape
    mk = lambda want, shape: torch.empty(shape, dtype=bf16, device=w.device) if want else None
    o = (mk(sp, (2 * rows, cols)), mk(ext, (rows, 3 * cols)), mk(t_sp, (2 * cols, rows)), mk(t_ext, (cols, 3 * rows)))
    check(_lib.load().ftmi_lora_split(ptr(w), rows, cols, ptr(o[0]), ptr(o[1]), ptr(o[2]), ptr(o[3]), stream_ptr()), "ftmi_lora_split")
    return tuple(t for t in o if t is not None)


def linear_lora_fwd(x, w, bias, lora_a, lora_b, lora_scale: float, variant: int = 8):
    """peft ``lora.Linear`` forward over a frozen bf16 Linear with fp32 adapters ``lora_a`` [r,K], ``lora_b`` [N,r] (None: plain
    linear).  Returns (y [M,N] bf16, xa [M,3r]: lora_scale * x A^T as bf16 (hi | lo | hi) planes, kept for the backward)."""
    M, K = x.shape
    N = w.shape[0]
    r = 0 if lora_a is None else lora_a.shape[0]
    y = torch.empty((M, N), dtype=bf16, device=x.device)
    xa = torch.empty((M, 3 * r), dtype=bf16, device=x.device) if r else None
    a_sp = b_ext = None
    if r:
        (a_sp,) = lora_split(lora_a, sp=True)
        (b_ext,) = lora_split(lora_b, ext=True)
    check(_lib.load().ftmi_linear_lora_fwd(M, K, N, r, float(lora_scale), ptr(x), ptr(w), ptr(bias), ptr(a_sp), ptr(b_ext), ptr(y), ptr(xa),
                                            variant, stream_ptr()), "ftmi_linear_lora_fwd")
    return y, xa


def linear_lora_bwd(x, dy, xa, w_t, lora_a, lora_b, lora_scale: float, grad_a=None, grad_b=None, need_dx: bool = True, variant: int = 8):
    """Backward of ``linear_lora_fwd``: returns (dx [M,K] bf16 or None, grad_a [r,K] fp32, grad_b [N,r] fp32); the gradient
    buffers are accumulated into when given (``.grad`` semantics).  ``w_t = W^T`` bf16; ``lora_a`` / ``lora_b`` fp32."""
    M, N = dy.shape
    K = w_t.shape[0] if w_t is not None else x.shape[1]
    r = 0 if lora_a is None else lora_a.shape[0]
    dx = torch.empty((M, K), dtype=bf16, device=dy.device) if need_dx else None
    dxa = torch.empty((M, 3 * r), dtype=bf16, device=dy.device) if r else None
    bt_sp = at_ext = None
    if r:
        (at_ext,) = lora_split(lora_a, t_ext=True)
        (bt_sp,) = lora_split(lora_b, t_sp=True)
        grad_a = torch.zeros((r, K), dtype=torch.float32, device=dy.device) if grad_a is None else grad_a
        grad_b = torch.zeros((N, r), dtype=torch.float32, device=dy.device) if grad_b is None else grad_b
    check(_lib.load().ftmi_linear_lora_bwd(M, K, N, r, float(lora_scale), ptr(x), ptr(dy), ptr(xa), ptr(w_t), ptr(bt_sp), ptr(at_ext), ptr(dxa),
                                            ptr(dx), ptr(grad_a), ptr(grad_b), variant, stream_ptr()), "ftmi_linear_lora_bwd")
    return dx, grad_a, grad_b


def norm_modulate(x, shift, onep, rows_per_batch: int, eps: float = 1e-6, layernorm: bool = False):
    """y = bf16(bf16(norm(x)) * onep[b]) + shift[b]; x [rows, 2048] bf16, shift / onep [B, 2048] bf16 (onep = 1 + scale)."""
    rows, D = x.shape
    y = torch.empty_like(x)
    check(_lib.load().ftmi_norm_modulate_fwd(ptr(x), ptr(shift), ptr(onep), shift.stride(0), ptr(y), rows, rows_per_batch, D, float(eps), int(layernorm),
                                              stream_ptr()), "ftmi_norm_modulate_fwd")
    return y


def norm_modulate_bwd(x, dy, onep, rows_per_batch: int, eps: float = 1e-6, layernorm: bool = False, dres=None):
    rows, D = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().ftmi_norm_modulate_bwd(ptr(x), ptr(dy), ptr(onep), onep.stride(0), ptr(dres), ptr(dx), rows, rows_per_batch, D, float(eps),
                                              int(layernorm), stream_ptr()), "ftmi_norm_modulate_bwd")
    return dx


def qknorm_rope(x, w, cos=None, sin=None, rows_per_batch: Optional[int] = None, eps: float = 1e-5):
    """rope(bf16(rms_norm(x) * w)); x [rows, 2048] bf16 (row stride free), cos / sin fp32 [rows_per_batch, 1024] or None."""
    rows, D = x.shape
    y = torch.empty((rows, D), dtype=bf16, device=x.device)
    check(_lib.load().ftmi_qknorm_rope_fwd(ptr(x), x.stride(0), ptr(w), ptr(cos), ptr(sin), ptr(y), D, rows, rows_per_batch or rows, D, float(eps),
                                            stream_ptr()), "ftmi_qknorm_rope_fwd")
    return y


def qknorm_rope_bwd(x, w, dy, cos=None, sin=None, rows_per_batch: Optional[int] = None, eps: float = 1e-5):
    rows, D = x.shape
    dx = torch.empty((rows, D), dtype=bf16, device=x.device)
    check(_lib.load().ftmi_qknorm_rope_bwd(ptr(x), x.stride(0), ptr(w), ptr(cos), ptr(sin), ptr(dy), dy.stride(0), ptr(dx), D, rows, rows_per_batch or rows, D,
                                            float(eps), stream_ptr()), "ftmi_qknorm_rope_bwd")
    return dx


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


def ddim_add_noise(latents, noise, sqrt_alpha, sqrt_one_minus_alpha, scaling_factor: float = 1.0):
    """CogVideoX noising: (x0 = bf16(latents * scaling_factor), noisy = scheduler.add_noise(x0, noise, t)); per-sample coefficients fp32 [B]."""
    require_gpu_tensor(latents, "latents", bf16)
    latents, noise = latents.contiguous(), noise.contiguous()
    B = latents.shape[0]
    x0, noisy = torch.empty_like(latents), torch.empty_like(latents)
    check(_lib.load().ftmi_ddim_add_noise(ptr(latents), ptr(noise), ptr(sqrt_alpha), ptr(sqrt_one_minus_alpha), float(scaling_factor), ptr(x0), ptr(noisy), B,
                                           latents[0].numel(), stream_ptr()), "ftmi_ddim_add_noise")
    return x0, noisy


def ddim_get_velocity(sample, noise, sqrt_alpha, sqrt_one_minus_alpha):
    """scheduler.get_velocity(sample, noise, t) = sqrt(a) noise - sqrt(1 - a) sample (bf16 op by op)."""
    require_gpu_tensor(sample, "sample", bf16)
    sample, noise = sample.contiguous(), noise.contiguous()
    out = torch.empty_like(sample)
    check(_lib.load().ftmi_ddim_get_velocity(ptr(sample), ptr(noise), ptr(sqrt_alpha), ptr(sqrt_one_minus_alpha), ptr(out), sample.shape[0], sample[0].numel(),
                                              stream_ptr()), "ftmi_ddim_get_velocity")
    return out


# ---- CogVideoX block, row-wise stages (include/ftmi355.h: ftmi_cog_*) ------------------------------------------------------------------
def _cog_rows(x):
    require_gpu_tensor(x, "x", bf16)
    if x.dim() != 3 or not x.is_contiguous():
        raise ValueError("expected a contiguous [B, tokens, D] bf16 tensor (text tokens of a sample first)")
    return x.shape[0] * x.shape[1], x.shape[2], x.shape[1]


def cog_ln_mod(x, w, b, shift, onep, text_len: int, eps: float = 1e-5, out=None):
    """CogVideoXLayerNormZero body: bf(bf(LayerNorm(x; w, b)) * onep) + shift; shift / onep [B, 2, D] (text, video) or [B, D] (text_len = 0)."""
    rows, D, rpb = _cog_rows(x)
    y = torch.empty_like(x) if out is None else out
    if y.shape != x.shape or not y.is_contiguous():
        raise ValueError("cog_ln_mod: out must be a contiguous tensor of x's shape")
    check(_lib.load().ftmi_cog_ln_mod_fwd(ptr(x), ptr(w), ptr(b), ptr(shift.contiguous()), ptr(onep.contiguous()), ptr(y), rows, D, rpb, int(text_len),
                                          float(eps), stream_ptr()), "ftmi_cog_ln_mod_fwd")
    return y


def cog_ln_mod_bwd(x, w, onep, dy, text_len: int, eps: float = 1e-5, dres=None):
    rows, D, rpb = _cog_rows(x)
    dx = torch.empty_like(x)
    check(_lib.load().ftmi_cog_ln_mod_bwd(ptr(x), ptr(w), ptr(onep.contiguous()), ptr(dy.contiguous()), ptr(dres), ptr(dx), rows, D, rpb, int(text_len),
                                          float(eps), stream_ptr()), "ftmi_cog_ln_mod_bwd")
    return dx


def _rope_args(rope, rows_per_batch, text_len):
    if rope is None:
        return None, None, 0, 0
    cos, sin = rope
    require_gpu_tensor(cos, "rope cos", torch.float32)
    require_gpu_tensor(sin, "rope sin", torch.float32)
    if cos.shape != sin.shape or cos.dim() != 2 or cos.shape[1] != 64 or cos.shape[0] != rows_per_batch - text_len or not cos.is_contiguous() or not sin.is_contiguous():
        raise ValueError(f"rotary tables must be contiguous fp32 [video tokens = {rows_per_batch - text_len}, 64], got {tuple(cos.shape)}")
    return cos, sin, int(rows_per_batch), int(text_len)


def cog_head_ln(x2d, w, b, eps: float = 1e-6, out=None, rope=None, rows_per_batch: int = 0, text_len: int = 0):
    """LayerNorm over each 64-channel head of the rows of x2d [M, D] (a column slice of a wider buffer is fine: row stride = x2d.stride(0));
    ``rope = (cos, sin)`` fp32 [S, 64]: rotary embedding on the rows at position >= text_len of every rows_per_batch-token sample."""
    require_gpu_tensor(x2d, "x", bf16)
    M, D = x2d.shape
    cos, sin, rpb, tl = _rope_args(rope, rows_per_batch, text_len)
    if x2d.stride(1) != 1:
        raise ValueError("head LayerNorm: channels must be contiguous")
    y = torch.empty((M, D), dtype=bf16, device=x2d.device) if out is None else out
    if y.stride(0) != x2d.stride(0):
        y_c = torch.empty_strided((M, D), (x2d.stride(0), 1), dtype=bf16, device=x2d.device)
    else:
        y_c = y
    check(_lib.load().ftmi_cog_head_ln_fwd(ptr(x2d), x2d.stride(0), ptr(w), ptr(b), ptr(y_c), M, D, float(eps), ptr(cos), ptr(sin), rpb, tl, stream_ptr()),
          "ftmi_cog_head_ln_fwd")
    if y_c is not y:
        y.copy_(y_c)
    return y


def cog_head_ln_bwd(x2d, w, dy2d, eps: float = 1e-6, rope=None, rows_per_batch: int = 0, text_len: int = 0):
    require_gpu_tensor(x2d, "x", bf16)
    M, D = x2d.shape
    cos, sin, rpb, tl = _rope_args(rope, rows_per_batch, text_len)
    if x2d.stride(0) != dy2d.stride(0) or x2d.stride(1) != 1 or dy2d.stride(1) != 1:
        raise ValueError("head LayerNorm backward: x and dy must share one row stride")
    dx = torch.empty_strided((M, D), (x2d.stride(0), 1), dtype=bf16, device=x2d.device)
    check(_lib.load().ftmi_cog_head_ln_bwd(ptr(x2d), x2d.stride(0), ptr(w), ptr(dy2d), ptr(dx), M, D, float(eps), ptr(cos), ptr(sin), rpb, tl, stream_ptr()),
          "ftmi_cog_head_ln_bwd")
    return dx


def cog_gate_residual(res, y, gate, text_len: int, out=None):
    """res + bf(gate * y) per segment (res None: bf(gate * y)); ``out`` may be ``y`` or ``res`` themselves (element-wise, in place)."""
    rows, D, rpb = _cog_rows(y)
    out = torch.empty_like(y) if out is None else out
    if out.shape != y.shape or not out.is_contiguous() or (res is not None and (res.shape != y.shape or not res.is_contiguous())):
        raise ValueError("cog_gate_residual: res / out must be contiguous tensors of y's shape")
    check(_lib.load().ftmi_cog_gate_residual(ptr(res), ptr(y), ptr(gate.contiguous()), ptr(out), rows, D, rpb, int(text_len), stream_ptr()),
          "ftmi_cog_gate_residual")
    return out


def cog_patchify(latents, patch: int):
    """latents [B, F, C, H, W] bf16 -> tokens [B, F (H/p) (W/p), C p p] (channel order (c, py, px): the flattened Conv2d weight's)."""
    require_gpu_tensor(latents, "latents", bf16)
    B, F_, C, H, W = latents.shape
    latents = latents.contiguous()
    out = torch.empty((B, F_ * (H // patch) * (W // patch), C * patch * patch), dtype=bf16, device=latents.device)
    check(_lib.load().ftmi_cog_patchify(ptr(latents), ptr(out), B, F_, C, H, W, int(patch), stream_ptr()), "ftmi_cog_patchify")
    return out


def cog_unpatchify(tokens, F_: int, C: int, H: int, W: int, patch: int):
    """tokens [B, F (H/p) (W/p), C p p] bf16 -> latents [B, F, C, H, W]."""
    require_gpu_tensor(tokens, "tokens", bf16)
    B = tokens.shape[0]
    if tokens.shape[1:] != (F_ * (H // patch) * (W // patch), C * patch * patch):
        raise ValueError(f"cog_unpatchify: tokens {tuple(tokens.shape)} do not match F={F_} C={C} H={H} W={W} p={patch}")
    tokens = tokens.contiguous()
    out = torch.empty((B, F_, C, H, W), dtype=bf16, device=tokens.device)
    check(_lib.load().ftmi_cog_unpatchify(ptr(tokens), ptr(out), B, F_, C, H, W, int(patch), stream_ptr()), "ftmi_cog_unpatchify")
    return out


def posterior_sample(moments, eps):
    """moments [B, 2C, ...] (mean | logvar along dim 1), eps [B, C, ...] ~ N(0,1) -> mean + exp(0.5 * clamp(logvar, -30, 20)) * eps."""
    require_gpu_tensor(moments, "moments", bf16)
    require_gpu_tensor(eps, "eps", bf16)
    if moments.shape[1] != 2 * eps.shape[1] or moments.shape[2:] != eps.shape[2:] or moments.shape[0] != eps.shape[0]:
        raise ValueError(f"posterior_sample: moments {tuple(moments.shape)} do not hold (mean | logvar) for eps {tuple(eps.shape)}")
    moments, eps = moments.contiguous(), eps.contiguous()
    out = torch.empty_like(eps)
    check(_lib.load().ftmi_posterior_sample(ptr(moments), ptr(eps), ptr(out), eps.shape[0], eps[0].numel(), stream_ptr()), "ftmi_posterior_sample")
    return out


MSE_SCRATCH_FLOATS_PER_SAMPLE = 256  # FTMI_MSE_SCRATCH_FLOATS_PER_SAMPLE of include/ftmi355.h


def mse_loss(pred, target, weight: Optional[torch.Tensor], want_grad: bool = True, grad_scale: float = 1.0):
    B = pred.shape[0]
    per = pred[0].numel()
    loss = torch.empty((1,), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    scratch = torch.empty((MSE_SCRATCH_FLOATS_PER_SAMPLE * B,), dtype=torch.float32, device=pred.device)  # caller-owned partial sums (ftmi355.h)
    check(_lib.load().ftmi_mse_loss(ptr(pred), ptr(target), ptr(weight), ptr(loss), ptr(dpred), B, per, float(grad_scale), ptr(scratch), stream_ptr()),
          "ftmi_mse_loss")
    return loss, dpred


def clip_grad_norm_(grads: torch.Tensor, max_norm: float, scratch: Optional[torch.Tensor] = None, grad_norm_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """In-place global-norm clip of a flat fp32 gradient buffer; returns the pre-clip norm (device scalar)."""
    require_gpu_tensor(grads, "grads", torch.float32)
    if not grads.is_contiguous():
        raise ValueError("clip_grad_norm_: the flat gradient buffer must be contiguous")
    if scratch is None:
        scratch = torch.empty((CLIP_SCRATCH_FLOATS,), dtype=torch.float32, device=grads.device)
    if grad_norm_out is None:
        grad_norm_out = torch.empty((1,), dtype=torch.float32, device=grads.device)
    check(_lib.load().ftmi_clip_grad_norm(ptr(grads), grads.numel(), float(max_norm), ptr(scratch), ptr(grad_norm_out), stream_ptr()), "ftmi_clip_grad_norm")
    return grad_norm_out


CLIP_SCRATCH_FLOATS = 2050  # include/ftmi355.h FTMI_CLIP_SCRATCH_FLOATS


def clip_adamw_step(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.95), eps: float = 1e-8,
                    weight_decay: float = 1e-4, max_norm: float = 1.0, scratch: Optional[torch.Tensor] = None,
                    grad_norm_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    n = params.numel()
    if scratch is None:
        scratch = torch.empty((CLIP_SCRATCH_FLOATS,), dtype=torch.float32, device=params.device)
    elif scratch.numel() < CLIP_SCRATCH_FLOATS or scratch.dtype != torch.float32:
        raise ValueError(f"clip_adamw_step: scratch must hold {CLIP_SCRATCH_FLOATS} fp32 values (FTMI_CLIP_SCRATCH_FLOATS)")
    if grad_norm_out is None:
        grad_norm_out = torch.empty((1,), dtype=torch.float32, device=params.device)
    check(_lib.load().ftmi_clip_adamw_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), n, float(max_norm), float(lr),
                                            float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), ptr(scratch),
                                            ptr(grad_norm_out), stream_ptr()), "ftmi_clip_adamw_step")
    return grad_norm_out


# ---- Wan-T2V full fine-tune (include/ftmi355.h: ftmi_wan_*; csrc/wan.hip) --------------------------------------------------------------------
def _rows2d(t: torch.Tensor, name: str) -> torch.Tensor:
    require_gpu_tensor(t, name, bf16)
    if t.dim() != 2 or t.stride(1) != 1 or t.stride(0) % 8 != 0:
        raise ValueError(f"{name}: a 2-D bf16 view with contiguous columns and a row stride that is a multiple of 8")
    return t


def _f32(t: Optional[torch.Tensor], name: str, shape=None) -> Optional[torch.Tensor]:
    if t is None:
        return None
    require_gpu_tensor(t, name, torch.float32)
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if t.stride(-1) != 1:
        raise ValueError(f"{name}: contiguous columns required")
    return t


def _wan_call(name: str, x, y, rows_per_batch, eps=1e-6, w=None, b=None, shift=None, scale=None, dy=None, dres=None, red1=None, red2=None, red_per_batch=False,
              rope=None, head_dim=0):
    rows, D = x.shape
    a = _lib.WanRowArgs()
    a.x, a.ld_x = x.data_ptr(), x.stride(0)
    a.w, a.b = ptr(w), ptr(b)
    a.shift, a.scale = ptr(shift), ptr(scale)
    a.mod_bstride = scale.stride(0) if scale is not None else 0
    if shift is not None and scale is not None and shift.stride(0) != scale.stride(0) and rows > rows_per_batch:
        raise ValueError("shift and scale must share their sample stride")
    a.dy, a.ld_dy = ptr(dy), (dy.stride(0) if dy is not None else 0)
    a.dres = ptr(dres)
    a.y, a.ld_y = ptr(y), (y.stride(0) if y is not None else 0)
    a.red1, a.red2, a.red_per_batch = ptr(red1), ptr(red2), int(bool(red_per_batch))
    if rope is not None:
        cos, sin = rope
        _f32(cos, "rope cos", (rows_per_batch, head_dim // 2))
        _f32(sin, "rope sin", (rows_per_batch, head_dim // 2))
        if not (cos.is_contiguous() and sin.is_contiguous()):
            raise ValueError("rope tables must be contiguous")
        a.rope_cos, a.rope_sin, a.head_dim = cos.data_ptr(), sin.data_ptr(), int(head_dim)
    a.rows, a.D, a.rows_per_batch, a.eps = rows, D, int(rows_per_batch), float(eps)
    check(getattr(_lib.load(), f"ftmi_wan_{name}")(ctypes.byref(a), stream_ptr()), f"ftmi_wan_{name}")


def wan_ln(x, rows_per_batch: int, w=None, b=None, shift=None, scale=None, eps: float = 1e-6, out=None):
    """y = bf(LN(float(x)) [* w + b] [* (1 + scale_b) + shift_b]); x [rows, D] bf16, shift / scale fp32 [B, D] views (sample stride free)."""
    x = _rows2d(x, "x")
    B = x.shape[0] // rows_per_batch
    out = torch.empty(x.shape, dtype=bf16, device=x.device) if out is None else _rows2d(out, "out")
    _wan_call("ln_fwd", x, out, rows_per_batch, eps, w=w, b=b, shift=_f32(shift, "shift", (B, x.shape[1])), scale=_f32(scale, "scale", (B, x.shape[1])))
    return out


def wan_ln_bwd(x, dy, rows_per_batch: int, w=None, scale=None, eps: float = 1e-6, dres=None, red1=None, red2=None, red_per_batch: bool = False):
    """dx = bf([dres +] bf(LN'(x)[dy * (w | 1 + scale_b)])); red1 += sum dy, red2 += sum dy * xhat (fp32 [B, D] views when red_per_batch, else [D])."""
    x, dy = _rows2d(x, "x"), _rows2d(dy, "dy")
    dx = torch.empty(x.shape, dtype=bf16, device=x.device)
    if dres is not None and (_rows2d(dres, "dres").stride(0) != dx.stride(0)):
        raise ValueError("dres must be contiguous like dx")
    B, D = x.shape[0] // rows_per_batch, x.shape[1]
    if red_per_batch:
        for r in (red1, red2):
            if r is not None and (tuple(r.shape) != (B, D) or r.stride(0) != D):
                raise ValueError("per-sample column sums must be contiguous [B, D] fp32")
    _wan_call("ln_bwd", x, dx, rows_per_batch, eps, w=w, scale=_f32(scale, "scale", (B, D)), dy=dy, dres=dres, red1=_f32(red1, "red1"), red2=_f32(red2, "red2"),
              red_per_batch=red_per_batch)
    return dx


def wan_rms_rope(x, w, rows_per_batch: int, rope=None, head_dim: int = 128, eps: float = 1e-6, out=None):
    x = _rows2d(x, "x")
    out = torch.empty(x.shape, dtype=bf16, device=x.device) if out is None else _rows2d(out, "out")
    _wan_call("rms_rope_fwd", x, out, rows_per_batch, eps, w=w, rope=rope, head_dim=head_dim)
    return out


def wan_rms_rope_bwd(x, w, dy, rows_per_batch: int, rope=None, head_dim: int = 128, eps: float = 1e-6, dweight=None, out=None):
    """dx of the RMSNorm (+ rotary) above; dweight (fp32 [D]) += sum dn * xhat."""
    x, dy = _rows2d(x, "x"), _rows2d(dy, "dy")
    out = torch.empty(x.shape, dtype=bf16, device=x.device) if out is None else _rows2d(out, "out")
    _wan_call("rms_rope_bwd", x, out, rows_per_batch, eps, w=w, dy=dy, rope=rope, head_dim=head_dim, red2=_f32(dweight, "dweight", (x.shape[1],)))
    return out


def wan_gate_res(x, y, rows_per_batch: int, gate=None, out=None):
    """out = bf(float(x) + float(y) * gate_b)  (gate fp32 [B, D]; None: bf(x + y))."""
    x, y = _rows2d(x, "x"), _rows2d(y, "y")
    out = torch.empty(x.shape, dtype=bf16, device=x.device) if out is None else _rows2d(out, "out")
    _wan_call("gate_res_fwd", x, out, rows_per_batch, scale=_f32(gate, "gate", (x.shape[0] // rows_per_batch, x.shape[1])), dy=y)
    return out


def wan_gate_res_bwd(dout, y, gate, rows_per_batch: int, dgate=None):
    """dy = bf(d out * gate_b); dgate (fp32 contiguous [B, D]) += sum_rows d out * y."""
    dout, y = _rows2d(dout, "dout"), _rows2d(y, "y")
    B, D = dout.shape[0] // rows_per_batch, dout.shape[1]
    if dgate is not None and (tuple(dgate.shape) != (B, D) or dgate.stride(0) != D):
        raise ValueError("dgate must be contiguous [B, D] fp32")
    dy = torch.empty(dout.shape, dtype=bf16, device=dout.device)
    _wan_call("gate_res_bwd", dout, dy, rows_per_batch, scale=_f32(gate, "gate", (B, D)), dy=y, red1=_f32(dgate, "dgate"), red_per_batch=True)
    return dy


def wan_colsum(x, out):
    """out (fp32 [N]) += sum over the rows of x [rows, N] (bias gradients)."""
    x = _rows2d(x, "x")
    _wan_call("colsum", x, None, x.shape[0], red1=_f32(out, "out", (x.shape[1],)))
    return out


def grad_sumsq(grads: torch.Tensor, scratch: torch.Tensor) -> torch.Tensor:
    """scratch[0] <- sum g^2 of a flat fp32 gradient (shard), order-fixed; returns scratch[:1]."""
    require_gpu_tensor(grads, "grads", torch.float32)
    if scratch.numel() < CLIP_SCRATCH_FLOATS or scratch.dtype != torch.float32:
        raise ValueError(f"grad_sumsq: scratch must hold {CLIP_SCRATCH_FLOATS} fp32 values")
    check(_lib.load().ftmi_grad_sumsq(ptr(grads), grads.numel(), ptr(scratch), stream_ptr()), "ftmi_grad_sumsq")
    return scratch[:1]


def adamw_bf16_step(params, grads, exp_avg, exp_avg_sq, step: int, lr: float, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4,
                    sumsq: Optional[torch.Tensor] = None, max_norm: float = 1.0, grad_norm_out: Optional[torch.Tensor] = None) -> None:
    """torch.optim.AdamW on flat bf16 parameters / moments with an fp32 gradient (clipped by the global norm sqrt(sumsq[0]) when given)."""
    for t, n in ((params, "params"), (exp_avg, "exp_avg"), (exp_avg_sq, "exp_avg_sq")):
        require_gpu_tensor(t, n, bf16)
    require_gpu_tensor(grads, "grads", torch.float32)
    if not (params.numel() == grads.numel() == exp_avg.numel() == exp_avg_sq.numel()):
        raise ValueError("adamw_bf16_step: size mismatch")
    check(_lib.load().ftmi_adamw_bf16_step(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), params.numel(), ptr(sumsq), float(max_norm), float(lr),
                                            float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), ptr(grad_norm_out), stream_ptr()),
          "ftmi_adamw_bf16_step")


def clip_by_sumsq_(grads: torch.Tensor, sumsq: torch.Tensor, max_norm: float, grad_norm_out: Optional[torch.Tensor] = None) -> None:
    """grads (flat fp32) *= min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)), in place."""
    require_gpu_tensor(grads, "grads", torch.float32)
    require_gpu_tensor(sumsq, "sumsq", torch.float32)
    check(_lib.load().ftmi_clip_by_sumsq(ptr(grads), grads.numel(), ptr(sumsq), float(max_norm), ptr(grad_norm_out), stream_ptr()), "ftmi_clip_by_sumsq")


# ---- HunyuanVideo (include/ftmi355.h: ftmi_head_rms_rope_*) ----------------------------------------------------------------------------------
def head_rms_rope(x2d, w, head_dim: int = 128, eps: float = 1e-6, rope=None, rows_per_batch: int = 0, rope_from: int = 0, out=None):
    """Per-head RMSNorm of x2d [rows, D] (a bf16 view, any row stride % 8) + real-form rotary embedding on the rows at position >= rope_from of each sample."""
    x2d = _rows2d(x2d, "x")
    rows, D = x2d.shape
    out = torch.empty((rows, D), dtype=bf16, device=x2d.device) if out is None else _rows2d(out, "out")
    cos, sin = (None, None) if rope is None else rope
    if rope is not None:
        n = (rows_per_batch or rows) - rope_from
        _f32(cos, "rope cos", (n, head_dim))
        _f32(sin, "rope sin", (n, head_dim))
    check(_lib.load().ftmi_head_rms_rope_fwd(ptr(x2d), x2d.stride(0), ptr(w), ptr(out), out.stride(0), rows, D, int(head_dim), float(eps), ptr(cos), ptr(sin),
                                             int(rows_per_batch or rows), int(rope_from), stream_ptr()), "ftmi_head_rms_rope_fwd")
    return out


def head_rms_rope_bwd(x2d, w, dy2d, head_dim: int = 128, eps: float = 1e-6, rope=None, rows_per_batch: int = 0, rope_from: int = 0, out=None):
    x2d, dy2d = _rows2d(x2d, "x"), _rows2d(dy2d, "dy")
    rows, D = x2d.shape
    out = torch.empty((rows, D), dtype=bf16, device=x2d.device) if out is None else _rows2d(out, "out")
    cos, sin = (None, None) if rope is None else rope
    check(_lib.load().ftmi_head_rms_rope_bwd(ptr(x2d), x2d.stride(0), ptr(w), ptr(dy2d), dy2d.stride(0), ptr(out), out.stride(0), rows, D, int(head_dim), float(eps),
                                             ptr(cos), ptr(sin), int(rows_per_batch or rows), int(rope_from), stream_ptr()), "ftmi_head_rms_rope_bwd")
    return out
