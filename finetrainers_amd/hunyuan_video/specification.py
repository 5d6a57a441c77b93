"""The arithmetic of ``HunyuanVideoModelSpecification.forward`` around the DiT call (finetrainers/models/hunyuan_video/base_specification.py:294-330) and the
SFT loss (trainer/sft_trainer/trainer.py:463-481), restated in oracle/hunyuan.py ``spec_forward`` (pinned to the reference's own ``forward`` by the
``hunyuan.spec_*`` golden fixtures): posterior draw (when the batch carries stored moments) or given latents, ``latents * vae.scaling_factor``, flow-match
mix, integer timesteps, ``guidance * 1000``, the DiT call with the condition dict, target ``noise - latents``.  The scaling and the mix are torch
elementwise ops on the (small) latent tensors with the reference's rounding points; the posterior draw and the loss are library kernels."""

from __future__ import annotations

from typing import Callable, Dict, Optional, Tuple

import torch

from .. import ops

bf16 = torch.bfloat16


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
