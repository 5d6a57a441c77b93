"""One HunyuanVideo LoRA SFT optimisation step (reference loop: finetrainers/trainer/sft_trainer/trainer.py:430-503 with the HunyuanVideo specification):
noising -> DiT forward -> MSE -> backward -> LoRA-gradient average over the data-parallel ranks -> global-norm clip -> AdamW, the last two fused over ONE flat
fp32 buffer that the blocks' adapter Parameters are views of.  First cut: one all-reduce of the flat gradient after the backward (315 MB at rank 64)."""

from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from .model import MI355XHunyuanVideoTransformer3DModel
from .specification import MI355XHunyuanVideoSpecOps


class MI355XHunyuanVideoSFTStep:
    def __init__(self, transformer: MI355XHunyuanVideoTransformer3DModel, spec: Optional[MI355XHunyuanVideoSpecOps] = None, lr: float = 2e-5, betas=(0.9, 0.95),
                 eps: float = 1e-8, weight_decay: float = 1e-4, max_grad_norm: float = 1.0, guidance: float = 1.0, parallel=None,
                 generator: Optional[torch.Generator] = None, lr_scheduler=None):
        self.params = transformer.lora_parameters()
        if not self.params:
            raise ValueError("attach a LoRA adapter first (transformer.add_adapter)")
        self.transformer, self.spec = transformer, spec or MI355XHunyuanVideoSpecOps()
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm, self.guidance = lr, betas, eps, weight_decay, max_grad_norm, guidance
        self.parallel, self.generator, self.lr_scheduler = parallel, generator, lr_scheduler
        dev = transformer.device
        self.flat = torch.cat([p.detach().reshape(-1) for p in self.params]).contiguous()
        off = 0
        for p in self.params:  # the adapters now live in one buffer: one fused clip + AdamW launch covers all 200 of them
            n = p.numel()
            p.data = self.flat[off:off + n].view(p.shape)
            off += n
        if parallel is not None and parallel.active:
            parallel.broadcast_(self.flat, src=0)
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self._scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        self.step_count = 0

    def step(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], sigmas: torch.Tensor, compute_posterior: bool = True,
             posterior_noise: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        pred, target, _ = self.spec.forward(self.transformer, latents, dict(conditions), sigmas, guidance=self.guidance, compute_posterior=compute_posterior,
                                            posterior_noise=posterior_noise, noise=noise, generator=self.generator)
        loss = self.spec.loss_backward(pred, target)
        gflat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in self.params])
        if self.parallel is not None and self.parallel.active:
            self.parallel.all_reduce_mean_(gflat)
        self.step_count += 1
        lr = self.lr if self.lr_scheduler is None else self.lr_scheduler.current_lr()
        gn = torch.empty(1, dtype=torch.float32, device=gflat.device)
        ops.clip_adamw_step(self.flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm,
                            scratch=self._scratch, grad_norm_out=gn)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        for p in self.params:
            p.grad = None
        return {"loss": loss.detach(), "grad_norm": gn}
