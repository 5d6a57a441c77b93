"""ctypes binding of libftmi355.so (C ABI: include/ftmi355.h).

The library is the product path: there is NO fallback.  If the shared object is missing or a symbol
is absent, importing callers get a RuntimeError telling them to build it (``python -m
finetrainers_amd.csrc.build``).  PyTorch appears here only as the owner of device memory and of the
HIP stream the kernels are enqueued on.
"""

from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_long, c_size_t, c_void_p
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# FTMI_LIB_PATH: tools-only override (A/B of two builds of the library in one gpurun call); the product always loads the in-tree .so
LIB_PATH = os.environ.get("FTMI_LIB_PATH") or os.path.join(_HERE, "libftmi355.so")

FTMI_ERR_INVALID = -1
FTMI_ERR_UNSUPPORTED = -2
FTMI_ERR_LAUNCH = -3

EPI_STORE, EPI_GELU, EPI_RESID, EPI_DGELU = 0, 1, 2, 3


class AttnDesc(Structure):
    _fields_ = [
        ("B", c_int), ("H", c_int), ("Sq", c_int), ("Sk", c_int), ("d", c_int),
        ("q_strides", c_long * 3), ("k_strides", c_long * 3), ("v_strides", c_long * 3), ("o_strides", c_long * 3),
        ("do_strides", c_long * 3), ("dq_strides", c_long * 3), ("dk_strides", c_long * 3), ("dv_strides", c_long * 3),
        ("bias_strides", c_long * 2),
        ("scale", c_float),
    ]


class LtxConfig(Structure):
    _fields_ = [
        ("B", c_int), ("S", c_int), ("T", c_int),
        ("D", c_int), ("H", c_int), ("L", c_int),
        ("C_in", c_int), ("C_out", c_int),
        ("D_ff", c_int), ("D_cap", c_int),
        ("r", c_int),
        ("lora_scale", c_float), ("eps_norm", c_float), ("eps_qk", c_float),
        ("gemm_variant", c_int),
        ("checkpoint", c_int),
        ("d_valid", c_int),
        ("head_dim_valid", c_int),
    ]


LTX_WEIGHT_FIELDS = [
    "proj_in_w", "proj_in_b", "time_l1_w", "time_l1_b", "time_l2_w", "time_l2_b", "time_lin_w", "time_lin_b",
    "cap_l1_w", "cap_l1_b", "cap_l2_w", "cap_l2_b", "tables", "table_out", "proj_out_w", "proj_out_b", "proj_out_w_t",
    "w_qkv", "b_qkv", "w_qkv_t", "norm_q", "norm_k", "w_o", "b_o", "w_o_t", "w_q2", "b_q2", "w_q2_t", "w_kv2", "b_kv2",
    "norm_q2", "norm_k2", "w_o2", "b_o2", "w_o2_t", "w_ff1", "b_ff1", "w_ff1_t", "w_ff2", "b_ff2", "w_ff2_t",
    "lora_a_sp", "lora_bt_sp", "lora_b_ext", "lora_at_ext", "lora_at_qkv_ext", "rope_cos", "rope_sin",
]


class LtxWeights(Structure):
    _fields_ = [(n, c_void_p) for n in LTX_WEIGHT_FIELDS]


class CogConfig(Structure):
    _fields_ = [
        ("B", c_int), ("T", c_int), ("S", c_int),
        ("D", c_int), ("H", c_int), ("L", c_int),
        ("D_ff", c_int), ("D_temb", c_int),
        ("r", c_int),
        ("lora_scale", c_float), ("eps_norm", c_float), ("eps_qk", c_float),
        ("gemm_variant", c_int),
    ]


COG_WEIGHT_FIELDS = [
    "mod_w", "mod_b", "norm_w", "norm_b", "w_qkv", "b_qkv", "w_o", "b_o", "qk_norm", "w_ff1", "b_ff1", "w_ff2", "b_ff2",
    "w_qkv_t", "w_o_t", "w_ff1_t", "w_ff2_t", "lora_a_sp", "lora_bt_sp", "lora_b_ext", "lora_at_ext", "lora_at_qkv_ext", "rope_cos", "rope_sin",
]


class CogWeights(Structure):
    _fields_ = [(n, c_void_p) for n in COG_WEIGHT_FIELDS]


class HySingleConfig(Structure):
    """include/ftmi355.h: ftmi_hy_single_config."""

    _fields_ = [("B", c_int), ("T", c_int), ("S", c_int), ("D", c_int), ("H", c_int), ("mlp", c_int), ("r", c_int), ("lora_scale", c_float), ("eps", c_float),
                ("gemm_variant", c_int)]


HY_SINGLE_WEIGHT_FIELDS = [
    "norm_lin_w", "norm_lin_b", "proj_mlp_w", "proj_mlp_b", "wq", "bq", "wk", "bk", "wv", "bv", "norm_q_w", "norm_k_w", "proj_out_w", "proj_out_b",
    "wq_t", "wk_t", "wv_t", "proj_mlp_w_t", "proj_out_w_t", "lora_a", "lora_b", "ones", "zeros",
]


class HySingleWeights(Structure):
    _fields_ = [(n, c_void_p) for n in HY_SINGLE_WEIGHT_FIELDS]


class HyDualConfig(Structure):
    """include/ftmi355.h: ftmi_hy_dual_config."""

    _fields_ = [("T", c_int), ("S", c_int), ("D", c_int), ("H", c_int), ("mlp", c_int), ("r", c_int), ("lora_scale", c_float), ("eps", c_float), ("gemm_variant", c_int)]


HY_DUAL_WEIGHT_FIELDS = [
    "norm1_lin_w", "norm1_lin_b", "norm1c_lin_w", "norm1c_lin_b", "wq", "bq", "wk", "bk", "wv", "bv", "wo", "bo",
    "add_q_w", "add_q_b", "add_k_w", "add_k_b", "add_v_w", "add_v_b", "add_out_w", "add_out_b", "norm_q_w", "norm_k_w", "norm_added_q_w", "norm_added_k_w",
    "ff1_w", "ff1_b", "ff2_w", "ff2_b", "ffc1_w", "ffc1_b", "ffc2_w", "ffc2_b",
    "wq_t", "wk_t", "wv_t", "wo_t", "add_q_w_t", "add_k_w_t", "add_v_w_t", "add_out_w_t", "ff1_w_t", "ff2_w_t", "ffc1_w_t", "ffc2_w_t",
    "lora_a", "lora_b", "ones", "zeros",
]


class HyDualWeights(Structure):
    _fields_ = [(n, c_void_p) for n in HY_DUAL_WEIGHT_FIELDS]


class WanBlockConfig(Structure):
    """include/ftmi355.h: ftmi_wan_block_config."""

    _fields_ = [("B", c_int), ("S", c_int), ("T", c_int), ("D", c_int), ("H", c_int), ("F", c_int), ("eps", c_float), ("gemm_variant", c_int)]


class WanRowArgs(Structure):
    """include/ftmi355.h: ftmi_wan_row_args."""

    _fields_ = [
        ("x", c_void_p), ("ld_x", c_long), ("w", c_void_p), ("b", c_void_p), ("shift", c_void_p), ("scale", c_void_p), ("mod_bstride", c_long),
        ("dy", c_void_p), ("ld_dy", c_long), ("dres", c_void_p), ("y", c_void_p), ("ld_y", c_long), ("red1", c_void_p), ("red2", c_void_p),
        ("red_per_batch", c_int), ("rope_cos", c_void_p), ("rope_sin", c_void_p), ("head_dim", c_int), ("rows", c_int), ("D", c_int),
        ("rows_per_batch", c_int), ("eps", c_float),
    ]


_SIGS = {
    "ftmi_version": (c_int, []),
    "ftmi_last_error": (c_int, [c_char_p, c_size_t]),
    "ftmi_prof_enable": (c_int, [c_int]),
    "ftmi_prof_summary": (c_int, [c_int, POINTER(ctypes.c_double), POINTER(c_long), POINTER(ctypes.c_double), POINTER(c_long),
                                  POINTER(ctypes.c_double), c_int]),
    "ftmi_attn_fwd": (c_int, [POINTER(AttnDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ftmi_attn_bwd": (c_int, [POINTER(AttnDesc)] + [c_void_p] * 12),
    "ftmi_linear_lora_fwd": (c_int, [c_int, c_int, c_int, c_int, c_float] + [c_void_p] * 7 + [c_int, c_void_p]),
    "ftmi_linear_lora_bwd": (c_int, [c_int, c_int, c_int, c_int, c_float] + [c_void_p] * 10 + [c_int, c_void_p]),
    "ftmi_gemm_nt_plan": (c_int, [c_int, c_int, c_int, c_int, c_int]),
    "ftmi_reload_switches": (c_int, []),
    "ftmi_fused_status": (c_int, []),
    "ftmi_allreduce_unique_id": (c_int, [c_void_p]),
    "ftmi_allreduce_init": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "ftmi_allreduce_bucket": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ftmi_allreduce_wait": (c_int, [c_void_p, c_void_p]),
    "ftmi_allreduce_buckets_issued": (c_long, [c_void_p]),
    "ftmi_allreduce_version": (c_int, []),
    "ftmi_allreduce_destroy": (c_int, [c_void_p]),
    "ftmi_gemm_nt": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_float, c_void_p, c_long, c_int,
                             c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_long, c_int, c_void_p]),
    "ftmi_gemm_tn": (c_int, [c_int, c_int, c_int, c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_float, c_void_p]),
    "ftmi_fp8_upcast": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ftmi_transpose_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ftmi_norm_modulate_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ftmi_norm_modulate_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_int, c_void_p]),
    "ftmi_qknorm_rope_fwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_float, c_void_p]),
    "ftmi_qknorm_rope_bwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_float,
                                     c_void_p]),
    "ftmi_ltx_workspace_bytes": (c_size_t, [POINTER(LtxConfig)]),
    "ftmi_ltx_workspace_offset": (c_int, [POINTER(LtxConfig), c_char_p, c_int, POINTER(c_size_t)]),
    "ftmi_ltx_forward": (c_int, [POINTER(LtxConfig), POINTER(LtxWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_size_t, c_void_p]),
    "ftmi_ltx_backward": (c_int, [POINTER(LtxConfig), POINTER(LtxWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_size_t, c_void_p]),
    "ftmi_ltx_backward_range": (c_int, [POINTER(LtxConfig), POINTER(LtxWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_size_t, c_int, c_int, c_int, c_void_p]),
    "ftmi_ltx_noise_pack": (c_int, [c_void_p] * 6 + [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ftmi_ddim_add_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "ftmi_ddim_get_velocity": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "ftmi_posterior_sample": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "ftmi_clip_grad_norm": (c_int, [c_void_p, c_long, c_float, c_void_p, c_void_p, c_void_p]),
    "ftmi_cog_ln_mod_fwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ftmi_cog_ln_mod_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ftmi_cog_head_ln_fwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ftmi_cog_head_ln_bwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ftmi_cog_gate_residual": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ftmi_cog_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ftmi_cog_unpatchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ftmi_mse_loss": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_float, c_void_p, c_void_p]),
    "ftmi_clip_adamw_step": (c_int, [c_void_p] * 4 + [c_long] + [c_float] * 6 + [c_int, c_void_p, c_void_p, c_void_p]),
    "ftmi_lora_refresh": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_void_p]),
    **{f"ftmi_wan_{n}": (c_int, [POINTER(WanRowArgs), c_void_p]) for n in ("ln_fwd", "ln_bwd", "rms_rope_fwd", "rms_rope_bwd", "gate_res_fwd",
                                                                           "gate_res_bwd", "colsum")},
    "ftmi_grad_sumsq": (c_int, [c_void_p, c_long, c_void_p, c_void_p]),
    "ftmi_head_rms_rope_fwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ftmi_head_rms_rope_bwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_long, c_int, c_int, c_int, c_float, c_void_p, c_void_p, c_int, c_int,
                                       c_void_p]),
    "ftmi_clip_by_sumsq": (c_int, [c_void_p, c_long, c_void_p, c_float, c_void_p, c_void_p]),
    "ftmi_adamw_bf16_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_float, c_float, c_float, c_float, c_float, c_float, c_int,
                                     c_void_p, c_void_p]),
    "ftmi_lora_refresh_n": (c_int, [c_void_p] * 7 + [c_int, c_int, c_int, c_int, c_void_p]),
    "ftmi_cog_workspace_bytes": (c_size_t, [POINTER(CogConfig)]),
    "ftmi_cog_blocks_forward": (c_int, [POINTER(CogConfig), POINTER(CogWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ftmi_cog_blocks_backward": (c_int, [POINTER(CogConfig), POINTER(CogWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                                         c_int, c_int, c_int, c_void_p]),
    "ftmi_lora_split": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "ftmi_hy_single_saved_bytes": (c_size_t, [POINTER(HySingleConfig)]),
    "ftmi_hy_single_scratch_bytes": (c_size_t, [POINTER(HySingleConfig)]),
    "ftmi_hy_single_forward": (c_int, [POINTER(HySingleConfig), POINTER(HySingleWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "ftmi_wan_block_saved_bytes": (c_size_t, [POINTER(WanBlockConfig)]),
    "ftmi_wan_block_scratch_bytes": (c_size_t, [POINTER(WanBlockConfig)]),
    "ftmi_wan_block_param_elements": (c_size_t, [POINTER(WanBlockConfig)]),
    "ftmi_wan_block_forward": (c_int, [POINTER(WanBlockConfig), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "ftmi_wan_block_backward": (c_int, [POINTER(WanBlockConfig), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "ftmi_hy_dual_saved_bytes": (c_size_t, [POINTER(HyDualConfig)]),
    "ftmi_hy_dual_scratch_bytes": (c_size_t, [POINTER(HyDualConfig)]),
    "ftmi_hy_dual_forward": (c_int, [POINTER(HyDualConfig), POINTER(HyDualWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "ftmi_hy_dual_backward": (c_int, [POINTER(HyDualConfig), POINTER(HyDualWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
    "ftmi_hy_single_backward": (c_int, [POINTER(HySingleConfig), POINTER(HySingleWeights), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS.keys())

# research builds only (FTMI_EXPERIMENTAL=1): bound when the library exports them (the #ifdef FTMI_EXPERIMENTAL section of include/ftmi355.h)
_EXPERIMENTAL_SIGS = {
    "ftmi_gemm_sk_plan": (c_int, [c_int, c_int, c_int, c_int, c_int, c_int, c_int, POINTER(c_int)]),
    "ftmi_gemm_sk_status": (c_int, []),
    "ftmi_gemm_sk_trace": (c_int, [POINTER(ctypes.c_ulonglong), c_int]),
}

_lib: Optional[ctypes.CDLL] = None


def lib_available() -> bool:
    return os.path.exists(LIB_PATH)


def load() -> ctypes.CDLL:
    """Load libftmi355.so and bind every symbol of include/ftmi355.h.  Fails loudly."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"libftmi355.so not found at {LIB_PATH}: the MI355X backend has no fallback path. "
            "Build it with `python -m finetrainers_amd.csrc.build` (needs hipcc, --offload-arch=gfx950)."
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"libftmi355.so does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in _EXPERIMENTAL_SIGS.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    buf = ctypes.create_string_buffer(512)
    load().ftmi_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc == 0:
        return
    msg = f"{what}: {last_error()} (code {rc})"
    if rc in (FTMI_ERR_INVALID, FTMI_ERR_UNSUPPORTED):
        raise ValueError(msg)
    raise RuntimeError(msg)


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def require_gpu_tensor(t: torch.Tensor, name: str, dtype: Optional[torch.dtype] = None) -> None:
    if not t.is_cuda:
        raise ValueError(f"{name} must live in GPU memory (the MI355X backend has no CPU path)")
    if dtype is not None and t.dtype != dtype:
        raise ValueError(f"{name} must be {dtype}, got {t.dtype}")
