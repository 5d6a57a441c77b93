"""The SFT optimisation step of finetrainers, MI355X-native.

``MI355XSFTStep.step`` restates the body of ``SFTTrainer._train`` (finetrainers/trainer/sft_trainer/trainer.py:
430-528) for the LoRA path: sigma sampling -> ``ModelSpecification.forward`` -> weighted MSE -> backward ->
(DP: all-reduce of the flat LoRA gradient over RCCL/xGMI) -> global-norm clip -> AdamW.  Differences that are
design, not drift:
  * loss + d(loss)/d(pred) is one kernel, clip + AdamW is one kernel over the flat fp32 LoRA buffer;
  * no host synchronisation inside the step: loss / grad-norm stay on the device (the reference does five
    ``.item()`` calls per step, trainer.py:483,506 and parallel/utils.py:11); ``step`` returns device scalars.
``sft_loss`` is the drop-in for trainer.py:473-480 when the reference's own loop is kept (a torch scalar whose
``backward()`` feeds the DiT backward).
"""

from __future__ import annotations

from typing import Any, Dict, Optional

import torch

from . import ops
from .utils import diffusion as diffusion_utils


class _MSELossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target, weight, grad_scale):
        loss, dpred = ops.mse_loss(pred.contiguous(), target.contiguous(), weight, want_grad=True, grad_scale=grad_scale)
        ctx.save_for_backward(dpred)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        # g is 1.0 for loss.backward(); keep generality without a host sync
        return dpred * g.to(dpred.dtype), None, None, None


def sft_loss(pred: torch.Tensor, target: torch.Tensor, sigmas: torch.Tensor, flow_weighting_scheme: str = "none",
             gradient_accumulation_steps: int = 1) -> torch.Tensor:
    """trainer.py:463-480: ``weights * (pred.float() - target.float())**2`` averaged over all non-batch dims then the
    batch, divided by the accumulation steps.  ``sigmas`` as returned by the spec ([B,S,1], constant per sample)."""
    per_sample_sigma = sigmas.reshape(sigmas.shape[0], -1)[:, 0].float()
    weights = diffusion_utils.compute_loss_weighting_for_sd3(flow_weighting_scheme, per_sample_sigma).float().contiguous()
    scale = 1.0 / gradient_accumulation_steps
    loss = _MSELossFunction.apply(pred, target, weights, scale)
    return loss * scale if gradient_accumulation_steps > 1 else loss


class MI355XSFTStep:
    """One LoRA SFT optimisation step on one rank (one process per GPU).  ``parallel`` is a
    ``finetrainers_amd.parallel.DataParallelBackend`` (or None for a single GPU)."""

    def __init__(self, transformer, specification, lr: float = 5e-5, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4,
                 max_grad_norm: float = 1.0, flow_weighting_scheme: str = "none", flow_logit_mean: float = 0.0, flow_logit_std: float = 1.0,
                 flow_mode_scale: float = 1.29, parallel=None, generator: Optional[torch.Generator] = None):
        if transformer.lora_A is None:
            raise ValueError("attach a LoRA adapter first (transformer.add_adapter)")
        self.transformer = transformer
        self.spec = specification
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.scheme = flow_weighting_scheme
        self.flow_logit_mean, self.flow_logit_std, self.flow_mode_scale = flow_logit_mean, flow_logit_std, flow_mode_scale
        self.parallel = parallel
        self.generator = generator
        dev = transformer.device
        # flat fp32 optimiser state matching transformer.lora_flat = [A | B]
        self.n_a, self.n_b = transformer.lora_A.numel(), transformer.lora_B.numel()
        self.exp_avg = torch.zeros(self.n_a + self.n_b, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self._scratch = torch.zeros(2, dtype=torch.float32, device=dev)
        self.step_count = 0
        from .ltx_video.specification import FlowMatchSigmas

        self.scheduler = FlowMatchSigmas()
        self.scheduler_sigmas = self.scheduler.sigmas.to(dev)

    def sample_sigmas(self, batch_size: int) -> torch.Tensor:
        """trainer.py:436-448."""
        return diffusion_utils.prepare_sigmas(
            scheduler=self.scheduler, sigmas=self.scheduler_sigmas, batch_size=batch_size, num_train_timesteps=1000,
            flow_weighting_scheme=self.scheme, flow_logit_mean=self.flow_logit_mean, flow_logit_std=self.flow_logit_std,
            flow_mode_scale=self.flow_mode_scale, device=self.scheduler_sigmas.device, generator=self.generator,
        )

    def step(self, condition_model_conditions: Dict[str, Any], latent_model_conditions: Dict[str, Any], sigmas: Optional[torch.Tensor] = None,
             **spec_kwargs) -> Dict[str, torch.Tensor]:
        tr = self.transformer
        latents = latent_model_conditions["latents"]
        B = latents.shape[0]
        if sigmas is None:
            sigmas = self.sample_sigmas(B)
        # 3. forward (trainer.py:452-461)
        pred, target, sig = self.spec.forward(
            transformer=tr, condition_model_conditions=dict(condition_model_conditions), latent_model_conditions=dict(latent_model_conditions),
            sigmas=sigmas, generator=self.generator, compute_posterior=True, **spec_kwargs,
        )
        # 4. loss + backward (trainer.py:463-481)
        loss = sft_loss(pred, target, sig, self.scheme)
        loss.backward()
        gflat = self._flat_grad(tr.lora_A.grad, tr.lora_B.grad)
        # DP: average the LoRA gradients across ranks (the reference's DDP does this inside backward, ptd.py:462-463)
        if self.parallel is not None and self.parallel.world_size > 1:
            self.parallel.all_reduce_mean_(gflat)
        # 5-6. clip (utils/torch.py:99-161) + AdamW (optimizer.py:117-125), fused over the flat buffer
        self.step_count += 1
        grad_norm = self._clip_adamw(gflat)
        tr.lora_A.grad = None  # optimizer.zero_grad(set_to_none=True) (trainer.py:503)
        tr.lora_B.grad = None
        return {"loss": loss.detach(), "grad_norm": grad_norm}

    def _flat_grad(self, ga: torch.Tensor, gb: torch.Tensor) -> torch.Tensor:
        gflat = self.transformer._grad_flat
        if gflat is not None and ga.data_ptr() == gflat.data_ptr() and gb.data_ptr() == gflat.data_ptr() + 4 * self.n_a:
            return gflat  # autograd kept our buffer (the usual case: .grad was None)
        return torch.cat([ga.reshape(-1), gb.reshape(-1)])

    def _clip_adamw(self, gflat: torch.Tensor) -> torch.Tensor:
        tr = self.transformer
        gn = torch.empty(1, dtype=torch.float32, device=gflat.device)
        ops.clip_adamw_step(tr.lora_flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr, self.betas, self.eps,
                            self.weight_decay, self.max_grad_norm, scratch=self._scratch, grad_norm_out=gn)
        tr._lora_versions = None  # parameters changed in place by the library: refresh the bf16 working copies next forward
        return gn
