"""Host-side sigma sampling / loss weighting of the SFT step -- same names, arguments and results as
finetrainers/utils/diffusion.py:38-130 (flow-match branch; ``compute_loss_weighting_for_sd3`` restated from
diffusers.training_utils).  A handful of scalars per step: this is host logic, not a kernel."""

from __future__ import annotations

import math
from typing import Optional

import torch


def default_flow_shift(sigmas: torch.Tensor, shift: float = 1.0) -> torch.Tensor:
    return (sigmas * shift) / (1 + (shift - 1) * sigmas)


def compute_density_for_timestep_sampling(weighting_scheme: str, batch_size: int, logit_mean: float = None, logit_std: float = None,
                                          mode_scale: float = None, device: torch.device = torch.device("cpu"),
                                          generator: Optional[torch.Generator] = None) -> torch.Tensor:
    if weighting_scheme == "logit_normal":
        u = torch.normal(mean=logit_mean, std=logit_std, size=(batch_size,), device=device, generator=generator)
        u = torch.nn.functional.sigmoid(u)
    elif weighting_scheme == "mode":
        u = torch.rand(size=(batch_size,), device=device, generator=generator)
        u = 1 - u - mode_scale * (torch.cos(math.pi * u / 2) ** 2 - 1 + u)
    else:
        u = torch.rand(size=(batch_size,), device=device, generator=generator)
    return u


def prepare_sigmas(scheduler, sigmas: torch.Tensor, batch_size: int, num_train_timesteps: int, flow_weighting_scheme: str = "none",
                   flow_logit_mean: float = 0.0, flow_logit_std: float = 1.0, flow_mode_scale: float = 1.29,
                   device: torch.device = torch.device("cpu"), generator: Optional[torch.Generator] = None) -> torch.Tensor:
    weights = compute_density_for_timestep_sampling(
        weighting_scheme=flow_weighting_scheme, batch_size=batch_size, logit_mean=flow_logit_mean, logit_std=flow_logit_std,
        mode_scale=flow_mode_scale, device=device, generator=generator,
    )
    indices = (weights * num_train_timesteps).long()
    return sigmas[indices]


def compute_loss_weighting_for_sd3(weighting_scheme: str, sigmas: torch.Tensor) -> torch.Tensor:
    if weighting_scheme == "sigma_sqrt":
        return (sigmas**-2.0).float()
    if weighting_scheme == "cosmap":
        bot = 1 - 2 * sigmas + 2 * sigmas**2
        return 2 / (math.pi * bot)
    return torch.ones_like(sigmas)


def prepare_loss_weights(scheduler, alphas: Optional[torch.Tensor] = None, sigmas: Optional[torch.Tensor] = None,
                         flow_weighting_scheme: str = "none") -> torch.Tensor:
    return compute_loss_weighting_for_sd3(sigmas=sigmas, weighting_scheme=flow_weighting_scheme)
