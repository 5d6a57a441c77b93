"""The arithmetic of ``WanModelSpecification.forward`` around the DiT call (finetrainers/models/wan/base_specification.py:433-493) and the SFT loss
(trainer/sft_trainer/trainer.py:463-481), restated in oracle/wan.py: ``spec_forward`` / ``sft_loss`` (pinned to the reference's own code by the
``wan.spec.*`` golden fixtures).

The reference forces ``compute_posterior = False`` for Wan (:446): the batch carries the stored VAE moments [B, 2C, F, H, W]; mean AND log-variance are
normalised (``_normalize_latents`` :571-577 -- it MULTIPLIES by ``latents_std``, the processors hand over 1 / std), a latent is sampled from the
posterior, mixed with noise by the flow-match rule, and the target is ``noise - latents``.  First cut: the normalisation and the flow-match mix are
torch elementwise ops on the (small) latent tensors, with the reference's rounding points; the posterior draw and the loss are library kernels."""

from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .. import ops
from ..utils.reference_base import as_drop_in, keep_or_default

bf16 = torch.bfloat16


class MI355XWanSpecOps:
    @property
    def _resolution_dim_keys(self) -> Dict[str, Tuple[int, ...]]:
        return {"latents": (2, 3, 4)}  # base_specification.py:333-335 ([B, C, F, H, W])

    @staticmethod
    def normalize_latents(latents: torch.Tensor, latents_mean: torch.Tensor, latents_std: torch.Tensor) -> torch.Tensor:
        mean = latents_mean.view(1, -1, 1, 1, 1).to(device=latents.device)
        std = latents_std.view(1, -1, 1, 1, 1).to(device=latents.device)
        return ((latents.float() - mean) * std).to(latents)

    def noise_and_target(self, moments: torch.Tensor, latents_mean: torch.Tensor, latents_std: torch.Tensor, sigmas: torch.Tensor,
                         posterior_noise: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        """-> (noisy [B, C, F, H, W], target, timesteps [B] long).  ``sigmas`` broadcastable to the latents ([B] or [B, 1, 1, 1, 1])."""
        moments = moments.to(bf16)
        B, C2 = moments.shape[:2]
        mu, logvar = torch.chunk(moments, 2, dim=1)
        norm = torch.cat([self.normalize_latents(mu, latents_mean, latents_std), self.normalize_latents(logvar, latents_mean, latents_std)], dim=1).contiguous()
        shape = (B, C2 // 2) + tuple(moments.shape[2:])
        if posterior_noise is None:
            posterior_noise = torch.zeros(shape, dtype=bf16, device=moments.device).normal_(generator=generator)
        latents = ops.posterior_sample(norm.view(B, C2, -1), posterior_noise.to(bf16).contiguous().view(B, C2 // 2, -1)).view(shape)
        if noise is None:
            noise = torch.zeros_like(latents).normal_(generator=generator)
        noise = noise.to(bf16)
        s = sigmas.view(B, 1, 1, 1, 1).to(latents.device)
        noisy = ((1.0 - s) * latents + s * noise).to(latents)  # functional/diffusion.py:4-6 (fp32 sigmas promote the mix; one cast back)
        return noisy, noise - latents, (sigmas.flatten() * 1000.0).long()

    def forward(self, transformer: Callable, moments: torch.Tensor, encoder_hidden_states: torch.Tensor, sigmas: torch.Tensor, latents_mean: torch.Tensor,
                latents_std: torch.Tensor, posterior_noise=None, noise=None, generator=None):
        noisy, target, timesteps = self.noise_and_target(moments, latents_mean, latents_std, sigmas, posterior_noise, noise, generator)
        pred = transformer(hidden_states=noisy, timestep=timesteps, encoder_hidden_states=encoder_hidden_states, return_dict=False)[0]
        return pred, target, sigmas

    def loss_backward(self, pred: torch.Tensor, target: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
        """MSE with unit weights (flow_weighting_scheme "none"), mean over everything but the batch, then over the batch; loss and d loss / d pred come out
        of one kernel and seed the backward of ``pred``'s graph."""
        loss, dpred = ops.mse_loss(pred.detach().contiguous(), target.contiguous(), None, want_grad=True, grad_scale=grad_scale)
        pred.backward(dpred)
        return loss.reshape(()) * grad_scale


IGNORE_KEYS_FOR_COLLATION = {"height", "width", "num_frames", "frame_rate", "rope_interpolation_scale", "return_dict", "attention_kwargs",
                             "cross_attention_kwargs", "joint_attention_kwargs", "latents_mean", "latents_std"}  # models/modeling_utils.py:22


class MI355XWanModelSpecification(MI355XWanSpecOps):
    """Mirror of ``WanModelSpecification`` (finetrainers/models/wan/base_specification.py:210-577) for the SFT hot path (T2V): same constructor keywords,
    ``_resolution_dim_keys``, ``load_diffusion_models``, ``collate_*`` (``latents_mean`` / ``latents_std`` pass through uncollated, modeling_utils.py:22),
    ``forward`` with the reference's signature, ``_save_model`` writing a diffusers transformer directory.  Text encoder, VAE, pipeline and validation stay
    with the reference."""

    def __init__(self, pretrained_model_name_or_path: Optional[str] = "Wan-AI/Wan2.1-T2V-1.3B-Diffusers", tokenizer_id: Optional[str] = None,
                 text_encoder_id: Optional[str] = None, transformer_id: Optional[str] = None, vae_id: Optional[str] = None,
                 text_encoder_dtype: torch.dtype = torch.bfloat16, transformer_dtype: torch.dtype = torch.bfloat16, vae_dtype: torch.dtype = torch.bfloat16,
                 revision: Optional[str] = None, cache_dir: Optional[str] = None, condition_model_processors: Optional[list] = None,
                 latent_model_processors: Optional[list] = None, transformer_config=None, **kwargs) -> None:
        if transformer_dtype != torch.bfloat16:
            raise ValueError("the MI355X backend computes in bf16 (fp32 accumulation); transformer_dtype must be torch.bfloat16")
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.tokenizer_id, self.text_encoder_id, self.transformer_id, self.vae_id = tokenizer_id, text_encoder_id, transformer_id, vae_id
        self.text_encoder_dtype, self.transformer_dtype, self.vae_dtype = text_encoder_dtype, transformer_dtype, vae_dtype
        self.revision, self.cache_dir = revision, cache_dir
        self.condition_model_processors = keep_or_default(self, "condition_model_processors", condition_model_processors, [])
        self.latent_model_processors = keep_or_default(self, "latent_model_processors", latent_model_processors, [])
        self.transformer_config = transformer_config

    def load_diffusion_models(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, device: Optional[torch.device] = None) -> Dict[str, object]:
        """-> {"transformer", "scheduler"} (base_specification.py:310-330).  With no ``state_dict`` the weights come from ``transformer_id`` or
        ``<pretrained_model_name_or_path>/transformer`` (a local diffusers directory); a path that does not resolve RAISES -- never random weights."""
        from .. import wire
        from .model import MI355XWanTransformer3DModel, WanTransformerConfig

        cfg = self.transformer_config
        if state_dict is None:
            directory = wire.resolve_transformer_dir(self.pretrained_model_name_or_path, self.transformer_id)
            disk = wire.load_transformer_config(directory)
            if disk:
                cfg = WanTransformerConfig.from_dict(disk)
            state_dict = wire.load_transformer_state_dict(directory)
        cfg = cfg or WanTransformerConfig()
        self.transformer_config = cfg
        transformer = MI355XWanTransformer3DModel(cfg, device=device)
        transformer.load_diffusers_state_dict(state_dict)
        from ..ltx_video.specification import FlowMatchSigmas  # the reference builds the same FlowMatchEulerDiscreteScheduler for LTX and Wan

        return {"transformer": transformer, "scheduler": FlowMatchSigmas()}

    @staticmethod
    def _collate(data):
        out = {}
        for k in data[0]:
            if k in IGNORE_KEYS_FOR_COLLATION:
                out[k] = data[0][k]
                continue
            vals = [d[k] for d in data]
            out[k] = torch.cat(vals) if torch.is_tensor(vals[0]) else vals
        return out

    def collate_conditions(self, data):
        return self._collate(data)

    def collate_latents(self, data):
        return self._collate(data)

    def forward(self, transformer, condition_model_conditions: Dict[str, torch.Tensor], latent_model_conditions: Dict[str, torch.Tensor],
                sigmas: torch.Tensor, scheduler=None, generator: Optional[torch.Generator] = None, compute_posterior: bool = True,
                noise: Optional[torch.Tensor] = None, posterior_noise: Optional[torch.Tensor] = None, **kwargs):
        """base_specification.py:433-493 -> (pred, target, sigmas).  ``compute_posterior`` is accepted and ignored exactly like the reference does (:446):
        "latents" are always the stored moments [B, 2C, F, H, W], with "latents_mean" / "latents_std" (= 1 / std) next to them."""
        if latent_model_conditions.get("latent_condition") is not None or condition_model_conditions.get("encoder_hidden_states_image") is not None:
            raise NotImplementedError("the image-to-video conditioning is not part of this path")
        latents = latent_model_conditions.pop("latents")
        mean, std = latent_model_conditions.pop("latents_mean"), latent_model_conditions.pop("latents_std")
        return MI355XWanSpecOps.forward(self, transformer, latents, condition_model_conditions["encoder_hidden_states"], sigmas, mean, std,
                                        posterior_noise=posterior_noise, noise=noise, generator=generator)

    def _save_model(self, directory: str, transformer, transformer_state_dict: Optional[Dict[str, torch.Tensor]] = None, scheduler=None) -> None:
        """base_specification.py:554-568: ``<directory>/transformer`` = config.json + diffusion_pytorch_model.safetensors with the diffusers parameter
        names and shapes (the patch embedding back in its Conv3d shape), loadable by ``WanTransformer3DModel.from_pretrained``."""
        import dataclasses
        import json
        import os

        from safetensors.torch import save_file

        cfg = transformer.config
        if transformer_state_dict is None:
            transformer_state_dict = transformer.state_dict_views()
        sd = {k: v.detach().to("cpu").contiguous() for k, v in transformer_state_dict.items()}
        pt, ph, pw = cfg.patch_size
        sd["patch_embedding.weight"] = sd["patch_embedding.weight"].reshape(cfg.inner_dim, cfg.in_channels, pt, ph, pw).contiguous()
        out = os.path.join(directory, "transformer")
        os.makedirs(out, exist_ok=True)
        save_file(sd, os.path.join(out, "diffusion_pytorch_model.safetensors"), metadata={"format": "pt"})
        conf = dict(dataclasses.asdict(cfg), _class_name="WanTransformer3DModel", patch_size=list(cfg.patch_size))
        with open(os.path.join(out, "config.json"), "w") as f:
            json.dump(conf, f, indent=2)
        if scheduler is not None:
            os.makedirs(os.path.join(directory, "scheduler"), exist_ok=True)
            with open(os.path.join(directory, "scheduler", "scheduler_config.json"), "w") as f:
                json.dump({"_class_name": "FlowMatchEulerDiscreteScheduler", "num_train_timesteps": 1000, "shift": 1.0}, f, indent=2)


# The public class: these overrides on top of the reference's own WanModelSpecification when finetrainers is importable (prepare_conditions,
# prepare_latents, load_condition_models, load_latent_models, load_pipeline, validation are then inherited), on StandaloneModelSpecification otherwise
MI355XWanModelSpecification = as_drop_in(MI355XWanModelSpecification, "finetrainers.models.wan", "WanModelSpecification")
