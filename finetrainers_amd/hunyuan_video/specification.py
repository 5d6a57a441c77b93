"""The arithmetic of ``HunyuanVideoModelSpecification.forward`` around the DiT call (finetrainers/models/hunyuan_video/base_specification.py:294-330) and the
SFT loss (trainer/sft_trainer/trainer.py:463-481), restated in oracle/hunyuan.py ``spec_forward`` (pinned to the reference's own ``forward`` by the
``hunyuan.spec_*`` golden fixtures): posterior draw (when the batch carries stored moments) or given latents, ``latents * vae.scaling_factor``, flow-match
mix, integer timesteps, ``guidance * 1000``, the DiT call with the condition dict, target ``noise - latents``.  The scaling and the mix are torch
elementwise ops on the (small) latent tensors with the reference's rounding points; the posterior draw and the loss are library kernels."""

from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .. import ops
from ..utils.reference_base import as_drop_in, keep_or_default

bf16 = torch.bfloat16


from ..ltx_video.specification import IGNORE_KEYS_FOR_COLLATION  # modeling_utils.py: keys passed through from the first sample


class MI355XHunyuanVideoSpecOps:
    def __init__(self, scaling_factor: float = 0.476986):
        self.scaling_factor = scaling_factor  # AutoencoderKLHunyuanVideo config [upstream]

    @property
    def _resolution_dim_keys(self) -> Dict[str, Tuple[int, ...]]:
        return {"latents": (2, 3, 4)}  # base_specification.py:152-153 ([B, C, F, H, W])

    def noise_and_target(self, latents: torch.Tensor, sigmas: torch.Tensor, compute_posterior: bool = True, posterior_noise: Optional[torch.Tensor] = None,
                         noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        latents = latents.to(bf16)
        B = latents.shape[0]
        if not compute_posterior:  # the batch carries the VAE moments [B, 2C, F, H, W]
            shape = (B, latents.shape[1] // 2) + tuple(latents.shape[2:])
            if posterior_noise is None:
                posterior_noise = torch.zeros(shape, dtype=bf16, device=latents.device).normal_(generator=generator)
            latents = ops.posterior_sample(latents.contiguous().view(B, latents.shape[1], -1), posterior_noise.to(bf16).contiguous().view(B, shape[1], -1)).view(shape)
        latents = latents * self.scaling_factor
        if noise is None:
            noise = torch.zeros_like(latents).normal_(generator=generator)
        noise = noise.to(bf16)
        s = sigmas.view(B, 1, 1, 1, 1).to(latents.device)
        noisy = ((1.0 - s) * latents + s * noise).to(latents)  # functional/diffusion.py:4-6
        return noisy, noise - latents, (sigmas.flatten() * 1000.0).long(), latents

    def forward(self, transformer: Callable, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], sigmas: torch.Tensor, guidance: float = 1.0,
                compute_posterior: bool = True, posterior_noise=None, noise=None, generator=None):
        """``conditions``: encoder_hidden_states, encoder_attention_mask, pooled_projections -> (pred, target, sigmas)."""
        noisy, target, timesteps, scaled = self.noise_and_target(latents, sigmas, compute_posterior, posterior_noise, noise, generator)
        g = scaled.new_full((scaled.size(0),), fill_value=guidance) * 1000.0
        pred = transformer(hidden_states=noisy, guidance=g, **conditions, timestep=timesteps, return_dict=False)[0]
        return pred, target, sigmas

    def loss_backward(self, pred: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
        loss, dpred = ops.mse_loss(pred.detach().contiguous(), target.contiguous(), None, want_grad=True, grad_scale=grad_scale)
        pred.backward(dpred)
        return loss.reshape(()) * grad_scale


class MI355XHunyuanVideoModelSpecification(MI355XHunyuanVideoSpecOps):
    """Mirror of ``HunyuanVideoModelSpecification`` (finetrainers/models/hunyuan_video/base_specification.py:100-420) for the SFT hot path: same constructor
    keywords, ``_resolution_dim_keys``, ``load_diffusion_models``, ``collate_*``, ``forward`` with the reference's signature, ``_save_lora_weights``.
    Text encoders (Llama + CLIP), VAE, pipeline and validation stay with the reference."""

    def __init__(self, pretrained_model_name_or_path: Optional[str] = "hunyuanvideo-community/HunyuanVideo", tokenizer_id: Optional[str] = None,
                 text_encoder_id: Optional[str] = None, transformer_id: Optional[str] = None, vae_id: Optional[str] = None,
                 text_encoder_dtype: torch.dtype = torch.bfloat16, transformer_dtype: torch.dtype = torch.bfloat16, vae_dtype: torch.dtype = torch.bfloat16,
                 revision: Optional[str] = None, cache_dir: Optional[str] = None, condition_model_processors: Optional[list] = None,
                 latent_model_processors: Optional[list] = None, transformer_config=None, vae_scaling_factor: float = 0.476986, **kwargs) -> None:
        if transformer_dtype != torch.bfloat16:
            raise ValueError("the MI355X backend computes in bf16 (fp32 accumulation); transformer_dtype must be torch.bfloat16 "
                             "(fp8 storage = --layerwise_upcasting_modules transformer -> transformer.apply_layerwise_casting())")
        super().__init__(scaling_factor=vae_scaling_factor)
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.tokenizer_id, self.text_encoder_id, self.transformer_id, self.vae_id = tokenizer_id, text_encoder_id, transformer_id, vae_id
        self.text_encoder_dtype, self.transformer_dtype, self.vae_dtype = text_encoder_dtype, transformer_dtype, vae_dtype
        self.revision, self.cache_dir = revision, cache_dir
        self.condition_model_processors = keep_or_default(self, "condition_model_processors", condition_model_processors, [])
        self.latent_model_processors = keep_or_default(self, "latent_model_processors", latent_model_processors, [])
        self.transformer_config = transformer_config

    def load_diffusion_models(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, device: Optional[torch.device] = None) -> Dict[str, object]:
        """-> {"transformer", "scheduler"} (base_specification.py:191-210).  With no ``state_dict`` the frozen weights come from ``transformer_id`` or
        ``<pretrained_model_name_or_path>/transformer`` (a local diffusers directory); a path that does not resolve RAISES -- never random weights."""
        from .. import wire
        from ..ltx_video.specification import FlowMatchSigmas
        from .model import HunyuanVideoTransformerConfig, MI355XHunyuanVideoTransformer3DModel

        cfg = self.transformer_config
        if state_dict is None:
            directory = wire.resolve_transformer_dir(self.pretrained_model_name_or_path, self.transformer_id)
            disk = wire.load_transformer_config(directory)
            if disk:
                cfg = HunyuanVideoTransformerConfig.from_dict(disk)
            state_dict = wire.load_transformer_state_dict(directory)
        cfg = cfg or HunyuanVideoTransformerConfig()
        self.transformer_config = cfg
        transformer = MI355XHunyuanVideoTransformer3DModel(cfg, device=device)
        transformer.load_diffusers_state_dict(state_dict)
        return {"transformer": transformer, "scheduler": FlowMatchSigmas()}

    @staticmethod
    def _collate(data):
        out = {}
        for k in data[0]:
            if k in IGNORE_KEYS_FOR_COLLATION:
                out[k] = data[0][k]
                continue
            vals = [d[k] for d in data]
            out[k] = torch.cat(vals) if torch.is_tensor(vals[0]) else vals  # per-sample non-tensor fields stay a list (modeling_utils.py collate_*)
        return out

    def collate_conditions(self, data):
        return self._collate(data)

    def collate_latents(self, data):
        return self._collate(data)

    def forward(self, transformer, condition_model_conditions: Dict[str, torch.Tensor], latent_model_conditions: Dict[str, torch.Tensor],
                sigmas: torch.Tensor, guidance: float = 1.0, scheduler=None, generator: Optional[torch.Generator] = None, compute_posterior: bool = True,
                noise: Optional[torch.Tensor] = None, posterior_noise: Optional[torch.Tensor] = None, **kwargs):
        """base_specification.py:294-330 -> (pred, target, sigmas); ``compute_posterior = False``: "latents" are the stored VAE moments [B, 2C, F, H, W]."""
        latents = latent_model_conditions.pop("latents")
        cond = {k: condition_model_conditions[k] for k in ("encoder_hidden_states", "encoder_attention_mask", "pooled_projections")}
        return MI355XHunyuanVideoSpecOps.forward(self, transformer, latents, cond, sigmas, guidance=guidance, compute_posterior=compute_posterior,
                                                 posterior_noise=posterior_noise, noise=noise, generator=generator)

    def _save_lora_weights(self, directory: str, transformer_state_dict: Optional[Dict[str, torch.Tensor]] = None, scheduler=None,
                           metadata: Optional[Dict[str, str]] = None, *args, **kwargs) -> None:
        """base_specification.py:383-399: ``pytorch_lora_weights.safetensors`` (``transformer.``-prefixed peft keys + metadata) and the scheduler config."""
        import json
        import os

        from .. import wire

        if transformer_state_dict is not None:
            wire.save_lora_weights(directory, transformer_state_dict, metadata)
        if scheduler is not None:
            os.makedirs(os.path.join(directory, "scheduler"), exist_ok=True)
            with open(os.path.join(directory, "scheduler", "scheduler_config.json"), "w") as f:
                json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "num_train_timesteps": 1000, "shift": 1.0}, f, indent=2)


# The public class: these overrides on top of the reference's own HunyuanVideoModelSpecification when finetrainers is importable (prepare_conditions,
# prepare_latents, load_condition_models, load_latent_models, load_pipeline, validation are then inherited), on StandaloneModelSpecification otherwise
MI355XHunyuanVideoModelSpecification = as_drop_in(MI355XHunyuanVideoModelSpecification, "finetrainers.models.hunyuan_video", "HunyuanVideoModelSpecification")
