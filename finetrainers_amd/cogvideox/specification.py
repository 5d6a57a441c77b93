"""CogVideoX on MI355X -- the first layer of the next model family (BASELINE config 3: CogVideoX-2b LoRA, 49x480x720,
DP = 8).  What exists: the spec-level arithmetic of ``CogVideoXModelSpecification.forward``
(finetrainers/models/cogvideox/base_specification.py:258-333) around the transformer call -- latent scaling, DDIM ``add_noise``,
``get_velocity`` (velocity -> x0), the ``1 / (1 - alphas_cumprod[t])`` loss weight (finetrainers/utils/diffusion.py:117-130) -- as gfx950
kernels behind the C ABI (``ftmi_ddim_add_noise`` / ``ftmi_ddim_get_velocity`` / ``ftmi_mse_loss``), and the joint text + video attention
of every CogVideoX block through the ``mi355x`` attention provider (``ftmi_attn_fwd`` / ``_bwd``; 226 + 17 550 tokens, 30 heads of 64).
``MI355XCogVideoXSpecOps.forward`` takes the transformer as a callable: the native DiT of ``model.py`` (CogVideoX-2b variant), or the
reference's diffusers model with the ``mi355x`` attention provider.  The oracle for all of it is ``oracle/cogvideox.py``.
"""

from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .. import ops
from ..utils.reference_base import as_drop_in, keep_or_default


from ..ltx_video.specification import IGNORE_KEYS_FOR_COLLATION  # modeling_utils.py: keys passed through from the first sample


class CogVideoXDDIMTables:
    """The two things the step needs from ``CogVideoXDDIMScheduler``: ``alphas_cumprod`` (fp32 [1000], built exactly as the scheduler's
    constructor does: scaled-linear betas, SNR shift, optional zero-terminal-SNR rescale) and ``config.num_train_timesteps``."""

    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.0120, snr_shift_scale: float = 3.0,
                 rescale_betas_zero_snr: bool = False):
        self.config = type("Cfg", (), {"num_train_timesteps": num_train_timesteps})()
        betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float64) ** 2
        ac = torch.cumprod(1.0 - betas, dim=0)
        ac = ac / (snr_shift_scale + (1 - snr_shift_scale) * ac)
        if rescale_betas_zero_snr:
            a = ac.sqrt()
            a0, aT = a[0].clone(), a[-1].clone()
            ac = ((a - aT) * (a0 / (a0 - aT))) ** 2
        self.alphas_cumprod = ac.float()

    def coefficients(self, timesteps: torch.Tensor, dtype: torch.dtype = torch.bfloat16) -> Tuple[torch.Tensor, torch.Tensor]:
        """(sqrt(alphas_cumprod[t]), sqrt(1 - alphas_cumprod[t])) as the scheduler computes them: the table is cast to the sample dtype FIRST,
        then indexed and square-rooted in that dtype.  Returned as fp32 tensors holding those bf16 values (what the kernels take)."""
        ac = self.alphas_cumprod.to(device=timesteps.device, dtype=dtype)
        return (ac[timesteps] ** 0.5).flatten().float().contiguous(), ((1 - ac[timesteps]) ** 0.5).flatten().float().contiguous()

    def loss_weights(self, timesteps: torch.Tensor) -> torch.Tensor:
        """utils/diffusion.py:125-128: 1 / (1 - alphas), alphas = scheduler_alphas[timesteps] in fp32 (trainer.py:466)."""
        return (1 / (1 - self.alphas_cumprod.to(timesteps.device)[timesteps])).float().contiguous()


class _GetVelocity(torch.autograd.Function):
    """``scheduler.get_velocity(sample=velocity, noise=noisy_latents, t)`` = sqrt(a) noisy - sqrt(1 - a) velocity, differentiable in the model
    output: d velocity = bf(-sqrt(1 - a) * d pred), through the same kernel."""

    @staticmethod
    def forward(ctx, velocity, noisy, sa, so):
        ctx.save_for_backward(sa, so)
        return ops.ddim_get_velocity(velocity, noisy, sa, so)

    @staticmethod
    def backward(ctx, dpred):
        sa, so = ctx.saved_tensors
        dpred = dpred.contiguous()
        return ops.ddim_get_velocity(dpred, torch.zeros_like(dpred), torch.zeros_like(sa), so), None, None, None


class MI355XCogVideoXSpecOps:
    """The arithmetic of ``CogVideoXModelSpecification.forward`` around the DiT call, on the GPU.  ``transformer`` is any callable with the
    reference's signature ``(hidden_states [B,F,C,H,W], encoder_hidden_states, timestep, image_rotary_emb, ofs, return_dict=False)``."""

    def __init__(self, scaling_factor: float = 1.15258426, invert_scale_latents: bool = False, scheduler: Optional[CogVideoXDDIMTables] = None):
        self.scaling_factor = 1.0 if invert_scale_latents else scaling_factor
        self.scheduler = scheduler or CogVideoXDDIMTables()

    @property
    def _resolution_dim_keys(self) -> Dict[str, Tuple[int, ...]]:
        return {"latents": (1, 3, 4)}  # base_specification.py:118-119 ([B, F, C, H, W])

    def noise_and_target(self, latents: torch.Tensor, sigmas: torch.Tensor, noise: Optional[torch.Tensor] = None,
                         generator: Optional[torch.Generator] = None):
        """:283-293: -> (noisy [B,F,C,H,W], target = scaled latents, timesteps [B] long)."""
        latents = latents.to(torch.bfloat16)
        timesteps = (sigmas.flatten() * 1000.0).long()
        if noise is None:
            noise = torch.zeros_like(latents).normal_(generator=generator)
        sa, so = self.scheduler.coefficients(timesteps)
        x0, noisy = ops.ddim_add_noise(latents, noise.to(torch.bfloat16), sa, so, self.scaling_factor)
        return noisy, x0, timesteps

    def forward(self, transformer: Callable, latents: torch.Tensor, encoder_hidden_states: torch.Tensor, sigmas: torch.Tensor,
                image_rotary_emb=None, noise: Optional[torch.Tensor] = None, generator: Optional[torch.Generator] = None):
        """-> (pred, target, sigmas) like the reference (:258-333)."""
        tcfg = getattr(transformer, "config", None)
        pt = getattr(tcfg, "patch_size_t", None)
        if pt is not None:  # :287-288 / :403-410 _pad_frames, as written there: F % pt == 0 still appends a full group of copies of the last frame
            extra = pt - latents.shape[1] % pt
            latents = torch.cat([latents, latents[:, -1:].expand(-1, extra, -1, -1, -1)], dim=1)
        noisy, target, timesteps = self.noise_and_target(latents, sigmas, noise, generator)
        if image_rotary_emb is None and getattr(tcfg, "use_rotary_positional_embeddings", False):  # base_specification.py:302-317
            from .model import rotary_tables

            image_rotary_emb = tuple(t.to(noisy.device) for t in rotary_tables(tcfg, noisy.shape[3], noisy.shape[4], noisy.shape[1]))
        ofs = None if getattr(tcfg, "ofs_embed_dim", None) is None else torch.full((noisy.shape[0],), 2.0, dtype=noisy.dtype, device=noisy.device)  # :296-300
        velocity = transformer(hidden_states=noisy, encoder_hidden_states=encoder_hidden_states, timestep=timesteps, image_rotary_emb=image_rotary_emb,
                               ofs=ofs, return_dict=False)[0]
        sa, so = self.scheduler.coefficients(timesteps)
        pred = _GetVelocity.apply(velocity.to(torch.bfloat16), noisy, sa, so)  # scheduler.get_velocity(velocity, noisy_latents, timesteps)
        return pred, target, sigmas

    def loss(self, pred: torch.Tensor, target: torch.Tensor, sigmas: torch.Tensor) -> torch.Tensor:
        """trainer.py:463-481 with the DDIM weights: device fp32 scalar (and d loss / d pred via ops.mse_loss when training)."""
        timesteps = (sigmas.flatten() * 1000.0).long()
        loss, _ = ops.mse_loss(pred.contiguous(), target.contiguous(), self.scheduler.loss_weights(timesteps), want_grad=False)
        return loss.reshape(())

    def loss_backward(self, pred: torch.Tensor, target: torch.Tensor, sigmas: torch.Tensor, grad_scale: float = 1.0) -> torch.Tensor:
        """Loss and ``loss.backward()`` in one: loss and d loss / d pred come out of one kernel and seed the backward of ``pred``'s graph (the DiT)."""
        timesteps = (sigmas.flatten() * 1000.0).long()
        loss, dpred = ops.mse_loss(pred.detach().contiguous(), target.contiguous(), self.scheduler.loss_weights(timesteps), want_grad=True,
                                   grad_scale=grad_scale)
        pred.backward(dpred)
        return loss.reshape(()) * grad_scale


class MI355XCogVideoXModelSpecification(MI355XCogVideoXSpecOps):
    """Mirror of ``CogVideoXModelSpecification`` (finetrainers/models/cogvideox/base_specification.py:80-420) for the SFT hot path: same constructor
    keywords, ``_resolution_dim_keys``, ``load_diffusion_models``, ``collate_*``, ``forward`` with the reference's signature, ``_save_lora_weights``.
    Text encoder, VAE, pipeline and validation stay with the reference."""

    def __init__(self, pretrained_model_name_or_path: Optional[str] = "THUDM/CogVideoX-5b", tokenizer_id: Optional[str] = None,
                 text_encoder_id: Optional[str] = None, transformer_id: Optional[str] = None, vae_id: Optional[str] = None,
                 text_encoder_dtype: torch.dtype = torch.bfloat16, transformer_dtype: torch.dtype = torch.bfloat16, vae_dtype: torch.dtype = torch.bfloat16,
                 revision: Optional[str] = None, cache_dir: Optional[str] = None, condition_model_processors: Optional[list] = None,
                 latent_model_processors: Optional[list] = None, transformer_config=None, vae_scaling_factor: float = 1.15258426,
                 invert_scale_latents: bool = False, **kwargs) -> None:
        if transformer_dtype != torch.bfloat16:
            raise ValueError("the MI355X backend computes in bf16 (fp32 accumulation); transformer_dtype must be torch.bfloat16")
        super().__init__(scaling_factor=vae_scaling_factor, invert_scale_latents=invert_scale_latents)
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.tokenizer_id, self.text_encoder_id, self.transformer_id, self.vae_id = tokenizer_id, text_encoder_id, transformer_id, vae_id
        self.text_encoder_dtype, self.transformer_dtype, self.vae_dtype = text_encoder_dtype, transformer_dtype, vae_dtype
        self.revision, self.cache_dir = revision, cache_dir
        self.condition_model_processors = keep_or_default(self, "condition_model_processors", condition_model_processors, [])
        self.latent_model_processors = keep_or_default(self, "latent_model_processors", latent_model_processors, [])
        self.transformer_config = transformer_config

    def load_diffusion_models(self, state_dict: Optional[Dict[str, torch.Tensor]] = None, device: Optional[torch.device] = None) -> Dict[str, object]:
        """-> {"transformer", "scheduler"} (base_specification.py:150-170).  With no ``state_dict`` the frozen weights come from ``transformer_id`` or
        ``<pretrained_model_name_or_path>/transformer`` (a local diffusers directory); a path that does not resolve RAISES -- never random weights."""
        from .. import wire
        from .model import CogVideoXTransformerConfig, MI355XCogVideoXTransformer3DModel

        cfg = self.transformer_config
        if state_dict is None:
            directory = wire.resolve_transformer_dir(self.pretrained_model_name_or_path, self.transformer_id)
            disk = wire.load_transformer_config(directory)
            if disk:
                fields = CogVideoXTransformerConfig.__dataclass_fields__
                cfg = CogVideoXTransformerConfig(**{k: disk[k] for k in fields if k in disk})
            state_dict = wire.load_transformer_state_dict(directory)
        cfg = cfg or CogVideoXTransformerConfig()
        self.transformer_config = cfg
        transformer = MI355XCogVideoXTransformer3DModel(cfg, device=device)
        transformer.load_diffusers_state_dict(state_dict)
        return {"transformer": transformer, "scheduler": self.scheduler}

    @staticmethod
    def _collate(data):
        out = {}
        for k in data[0]:
            if k in IGNORE_KEYS_FOR_COLLATION:
                out[k] = data[0][k]
                continue
            vals = [d[k] for d in data]
            out[k] = torch.cat([v if v.dim() > 0 else v[None] for v in vals], dim=0) if torch.is_tensor(vals[0]) else vals  # non-tensors: the list
        return out

    def collate_conditions(self, data):
        return self._collate(data)

    def collate_latents(self, data):
        return self._collate(data)

    def forward(self, transformer, condition_model_conditions: Dict[str, torch.Tensor], latent_model_conditions: Dict[str, torch.Tensor],
                sigmas: torch.Tensor, scheduler=None, generator: Optional[torch.Generator] = None, compute_posterior: bool = True,
                noise: Optional[torch.Tensor] = None, posterior_noise: Optional[torch.Tensor] = None, **kwargs):
        """base_specification.py:258-333: -> (pred, target, sigmas).  ``compute_posterior = False``: "latents" are the VAE posterior's moments
        [B, F, 2C, H, W] (what --enable_precomputation stores) and are sampled here (``DiagonalGaussianDistribution(..., _dim=2).sample``)."""
        latents = latent_model_conditions.pop("latents")
        if not compute_posterior:
            B, F_, C2, H, W = latents.shape
            mom = latents.to(torch.bfloat16).reshape(B * F_, C2, H, W)  # per (sample, frame): (mean | logvar) along the channel axis
            if posterior_noise is None:
                posterior_noise = torch.randn((B * F_, C2 // 2, H, W), generator=generator, device=mom.device, dtype=torch.bfloat16)
            latents = ops.posterior_sample(mom, posterior_noise.reshape(B * F_, C2 // 2, H, W).to(mom)).view(B, F_, C2 // 2, H, W)
        return MI355XCogVideoXSpecOps.forward(self, transformer, latents, condition_model_conditions["encoder_hidden_states"], sigmas, noise=noise,
                                              generator=generator)

    def _save_lora_weights(self, directory: str, transformer_state_dict: Optional[Dict[str, torch.Tensor]] = None, scheduler=None,
                           metadata: Optional[Dict[str, str]] = None, *args, **kwargs) -> None:
        """base_specification.py:366-384: ``pytorch_lora_weights.safetensors`` (``transformer.``-prefixed peft keys + metadata) and the scheduler config."""
        import json
        import os

        from .. import wire

        if transformer_state_dict is not None:
            wire.save_lora_weights(directory, transformer_state_dict, metadata)
        if scheduler is not None:
            os.makedirs(os.path.join(directory, "scheduler"), exist_ok=True)
            with open(os.path.join(directory, "scheduler", "scheduler_config.json"), "w") as f:
                json.dump({"_class_name": "CogVideoXDDIMScheduler", "num_train_timesteps": scheduler.config.num_train_timesteps, "beta_start": 0.00085,
                           "beta_end": 0.012, "beta_schedule": "scaled_linear", "snr_shift_scale": 3.0, "prediction_type": "v_prediction"}, f, indent=2)


# The public class: these overrides on top of the reference's own CogVideoXModelSpecification when finetrainers is importable (prepare_conditions,
# prepare_latents, load_condition_models, load_latent_models, load_pipeline, validation are then inherited), on StandaloneModelSpecification otherwise
MI355XCogVideoXModelSpecification = as_drop_in(MI355XCogVideoXModelSpecification, "finetrainers.models.cogvideox", "CogVideoXModelSpecification")
