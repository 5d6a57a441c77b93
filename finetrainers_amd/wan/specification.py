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
