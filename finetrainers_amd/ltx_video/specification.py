"""``MI355XLTXVideoModelSpecification`` -- the ModelSpecification plugin for the MI355X LTX-Video backend.

Mirrors the part of finetrainers' ``ModelSpecification`` interface the SFT step touches
(finetrainers/models/modeling_utils.py:26-300; LTX implementation finetrainers/models/ltx_video/
base_specification.py:93-459): ``_resolution_dim_keys``, ``load_diffusion_models``, ``collate_conditions``,
``collate_latents`` and ``forward`` (same names, argument meaning and return convention).  Everything that
feeds the step from outside the hot path (VAE / T5 loading, validation pipeline, saving) is out of scope for
this backend and raises ``NotImplementedError`` -- the reference's own spec stays in charge of those.

``forward`` = base_specification.py:271-345: normalise latents, draw noise, flow-match mix (with the 10 %
first-frame conditioning branch), pack, timesteps, DiT call, target.  Normalise + mix + pack + target run in one
gfx950 kernel (``ftmi_ltx_noise_pack``); the DiT runs in ``MI355XLTXVideoTransformer3DModel``.
"""

from __future__ import annotations

import random
from typing import Any, Dict, List, Optional, Tuple

import torch

from .. import ops
from ..utils.reference_base import as_drop_in, keep_or_default
from .transformer import LTXTransformerConfig, MI355XLTXVideoTransformer3DModel

# finetrainers/models/modeling_utils.py:22
IGNORE_KEYS_FOR_COLLATION = {"height", "width", "num_frames", "frame_rate", "rope_interpolation_scale", "return_dict", "attention_kwargs",
                             "cross_attention_kwargs", "joint_attention_kwargs", "latents_mean", "latents_std"}


class FlowMatchSigmas:
    """The one thing the step needs from ``FlowMatchEulerDiscreteScheduler``: its sigma table
    (1000 entries, 1.0 -> 0.001, shift 1.0) and ``config.num_train_timesteps``."""

    class _Cfg:
        num_train_timesteps = 1000

    config = _Cfg()

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0):
        t = torch.linspace(1, num_train_timesteps, num_train_timesteps, dtype=torch.float32).flip(0)
        s = t / num_train_timesteps
        self.sigmas = shift * s / (1 + (shift - 1) * s)
        self.config = type("Cfg", (), {"num_train_timesteps": num_train_timesteps})()


class MI355XLTXVideoModelSpecification:
    def __init__(
        self,
        pretrained_model_name_or_path: Optional[str] = "Lightricks/LTX-Video",
        tokenizer_id: Optional[str] = None,
        text_encoder_id: Optional[str] = None,
        transformer_id: Optional[str] = None,
        vae_id: Optional[str] = None,
        text_encoder_dtype: torch.dtype = torch.bfloat16,
        transformer_dtype: torch.dtype = torch.bfloat16,
        vae_dtype: torch.dtype = torch.bfloat16,
        revision: Optional[str] = None,
        cache_dir: Optional[str] = None,
        condition_model_processors: Optional[list] = None,
        latent_model_processors: Optional[list] = None,
        transformer_config: Optional[LTXTransformerConfig] = None,
        gemm_variant: int = 8,
        **kwargs,
    ) -> None:
        """Same keyword arguments as the reference constructor (modeling_utils.py:33-70, ltx_video/base_specification.py:94-122; built
        by train.py:48-66); ``transformer_config`` / ``gemm_variant`` are the MI355X additions."""
        if transformer_dtype != torch.bfloat16:
            raise ValueError("the MI355X backend computes in bf16 (fp32 accumulation); transformer_dtype must be torch.bfloat16")
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.tokenizer_id, self.text_encoder_id, self.transformer_id, self.vae_id = tokenizer_id, text_encoder_id, transformer_id, vae_id
        self.text_encoder_dtype, self.transformer_dtype, self.vae_dtype = text_encoder_dtype, transformer_dtype, vae_dtype
        self.revision, self.cache_dir = revision, cache_dir
        self.condition_model_processors = keep_or_default(self, "condition_model_processors", condition_model_processors, [])
        self.latent_model_processors = keep_or_default(self, "latent_model_processors", latent_model_processors, [])
        self.transformer_config = transformer_config or LTXTransformerConfig()
        self.vae_config = None
        self.gemm_variant = gemm_variant
        self.first_frame_conditioning_p = 0.1  # base_specification.py:282
        self.min_first_frame_sigma = 0.25      # base_specification.py:283

    # modeling_utils.py:81-86 / ltx base_specification.py:131-133
    @property
    def _resolution_dim_keys(self) -> Dict[str, Tuple[int, ...]]:
        return {"latents": (2, 3, 4)}

    def load_diffusion_models(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, device: Optional[torch.device] = None,
                              random_init_seed: Optional[int] = None) -> Dict[str, Any]:
        """-> {"transformer": nn.Module, "scheduler": ...} (base_specification.py:173-190).

        Called with no arguments, as the trainer does (trainer/sft_trainer/trainer.py:88), it loads the frozen base weights from
        ``transformer_id`` or ``<pretrained_model_name_or_path>/transformer`` (a local diffusers directory, read with safetensors) and
        RAISES when there is none -- it never silently trains on noise.  ``state_dict`` injects a diffusers-format state dict (parity
        tests); ``random_init_seed`` explicitly requests random weights of the production architecture (synthetic benchmarks)."""
        from .. import wire

        cfg = self.transformer_config
        if state_dict is None and random_init_seed is None:
            directory = wire.resolve_transformer_dir(self.pretrained_model_name_or_path, self.transformer_id)
            disk_cfg = wire.load_transformer_config(directory)
            if disk_cfg:
                cfg = LTXTransformerConfig(**{k: disk_cfg[k] for k in ("in_channels", "out_channels", "patch_size", "patch_size_t", "num_attention_heads",
                                                                      "attention_head_dim", "cross_attention_dim", "num_layers", "caption_channels")
                                              if k in disk_cfg})
                self.transformer_config = cfg
            state_dict = wire.load_transformer_state_dict(directory)
        # (the production geometry runs as it is; a narrower one -- the reference's dummy fixture, tests/models/ltx_video/base_specification.py:47-58 -- is embedded in
        #  it by zero padding and runs on the same kernels: ltx_video/narrow.py)
        from .narrow import build_ltx_transformer

        transformer = build_ltx_transformer(cfg, device=device, gemm_variant=self.gemm_variant)
        if state_dict is not None:
            transformer.load_diffusers_state_dict(state_dict)
        else:
            transformer.init_random_(random_init_seed)
        return {"transformer": transformer, "scheduler": FlowMatchSigmas()}

    def _save_lora_weights(self, directory: str, transformer_state_dict: Optional[Dict[str, torch.Tensor]] = None, scheduler=None,
                           metadata: Optional[Dict[str, str]] = None, *args, **kwargs) -> None:
        """base_specification.py:379-397: ``pytorch_lora_weights.safetensors`` with ``transformer.``-prefixed peft keys + metadata
        (the trainer passes ``{"lora_config": json}``, trainer.py:283-298).  Loadable by the reference's ``load_lora_weights``."""
        from .. import wire

        if transformer_state_dict is not None:
            wire.save_lora_weights(directory, transformer_state_dict, metadata)
        if scheduler is not None:
            import json
            import os

            sdir = os.path.join(directory, "scheduler")
            os.makedirs(sdir, exist_ok=True)
            with open(os.path.join(sdir, "scheduler_config.json"), "w") as f:  # FlowMatchEulerDiscreteScheduler().save_pretrained equivalent
                json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "num_train_timesteps": scheduler.config.num_train_timesteps, "shift": 1.0}, f, indent=2)



    def forward(
        self,
        transformer: MI355XLTXVideoTransformer3DModel,
        condition_model_conditions: Dict[str, torch.Tensor],
        latent_model_conditions: Dict[str, torch.Tensor],
        sigmas: torch.Tensor,
        generator: Optional[torch.Generator] = None,
        compute_posterior: bool = True,
        noise: Optional[torch.Tensor] = None,
        first_frame_sigma: Optional[torch.Tensor] = None,
        force_first_frame_branch: Optional[bool] = None,
        **kwargs,
    ) -> Tuple[torch.Tensor, ...]:
        """Same contract as the reference: returns ``(pred, target, sigmas[B,S,1])``.

        Extra keyword-only hooks for parity runs (the reference draws the branch from Python's global
        RNG): ``noise`` injects the N(0,1) draw, ``first_frame_sigma`` ([B] fp32) injects the first-frame sigma,
        ``force_first_frame_branch`` pins the 10 % branch on/off."""
        latents = latent_model_conditions.pop("latents")
        if not compute_posterior:
            # --enable_precomputation (trainer.py:374): "latents" are the VAE posterior's moments [B, 2C, F, H, W]; draw the sample here
            # (base_specification.py:285-289).  ``posterior_noise`` injects the N(0,1) draw for parity runs.
            moments = latents.to(torch.bfloat16)
            eps = kwargs.pop("posterior_noise", None)
            if eps is None:
                shp = (moments.shape[0], moments.shape[1] // 2) + tuple(moments.shape[2:])
                eps = torch.randn(shp, generator=generator, device=moments.device, dtype=torch.bfloat16)
            latents = ops.posterior_sample(moments, eps.to(device=moments.device, dtype=torch.bfloat16))
        latents_mean = latent_model_conditions.pop("latents_mean")
        latents_std = latent_model_conditions.pop("latents_std")
        num_frames = latent_model_conditions.get("num_frames", latents.shape[2])
        height = latent_model_conditions.get("height", latents.shape[3])
        width = latent_model_conditions.get("width", latents.shape[4])
        dev = latents.device
        B, C = latents.shape[:2]
        latents = latents.to(torch.bfloat16)
        # per-channel statistics broadcast over the batch (evident intent of base_specification.py:427-436; identical
        # to the reference at B == 1)
        mean = latents_mean.reshape(-1)[:C].to(device=dev, dtype=torch.float32).contiguous()
        std = latents_std.reshape(-1)[:C].to(device=dev, dtype=torch.float32).contiguous()
        if noise is None:
            noise = torch.zeros_like(latents).normal_(generator=generator)
        sig = sigmas.reshape(-1).to(device=dev, dtype=torch.float32).contiguous()
        if sig.numel() != B:
            raise ValueError("sigmas must hold one value per sample")

        take_branch = force_first_frame_branch
        if take_branch is None:
            take_branch = first_frame_sigma is not None or random.random() < self.first_frame_conditioning_p
        sig_first = None
        tokens_first = 0
        if take_branch:
            if first_frame_sigma is None:
                first_frame_sigma = torch.rand_like(sig) * sig
            sig_first = torch.min(first_frame_sigma.reshape(-1).to(sig), sig.new_full(sig.shape, self.min_first_frame_sigma)).contiguous()
            tokens_first = height * width  # tokens of latent frame 0 (patch size 1)

        noisy, target = ops.noise_pack(latents, noise.to(torch.bfloat16), mean, std, sig, sig_first, tokens_first)
        S = noisy.shape[1]
        sigmas_bs1 = sig.view(-1, 1, 1).expand(-1, S, -1)
        timesteps = (sig * 1000.0).long()  # one per sample; every token of a sample shares it (:319-320)

        # base_specification.py:324-334
        frame_rate, temporal_compression_ratio, vae_spatial_compression_ratio = 25, 8, 32
        rope_interpolation_scale = [1 / (frame_rate / temporal_compression_ratio), vae_spatial_compression_ratio, vae_spatial_compression_ratio]

        pred = transformer(
            hidden_states=noisy,
            encoder_hidden_states=condition_model_conditions["encoder_hidden_states"],
            encoder_attention_mask=condition_model_conditions.get("encoder_attention_mask"),
            timestep=timesteps,
            num_frames=num_frames,
            height=height,
            width=width,
            rope_interpolation_scale=rope_interpolation_scale,
            return_dict=False,
        )[0]
        return pred, target, sigmas_bs1


# The public class: these overrides on top of the reference's own LTXVideoModelSpecification when finetrainers is importable (prepare_conditions,
# prepare_latents, load_condition_models, load_latent_models, load_pipeline, validation are then inherited), on StandaloneModelSpecification otherwise
MI355XLTXVideoModelSpecification = as_drop_in(MI355XLTXVideoModelSpecification, "finetrainers.models.ltx_video", "LTXVideoModelSpecification")
