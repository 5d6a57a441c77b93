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
    """loss = scale * mean_b mean w_b (pred - target)^2 with d loss / d pred from the same kernel.  ``scale`` (1 / accumulation
    steps) is applied exactly ONCE: the kernel folds it into the saved gradient, the returned loss is multiplied here."""

    @staticmethod
    def forward(ctx, pred, target, weight, scale):
        loss, dpred = ops.mse_loss(pred.contiguous(), target.contiguous(), weight, want_grad=True, grad_scale=scale)
        ctx.save_for_backward(dpred)
        loss = loss.reshape(())
        return loss * scale if scale != 1.0 else loss

    @staticmethod
    def backward(ctx, g):
        (dpred,) = ctx.saved_tensors
        return dpred * g.to(dpred.dtype), None, None, None  # g = 1 for loss.backward(); no host sync either way


def sft_loss(pred: torch.Tensor, target: torch.Tensor, sigmas: torch.Tensor, flow_weighting_scheme: str = "none",
             gradient_accumulation_steps: int = 1) -> torch.Tensor:
    """trainer.py:463-480: ``weights * (pred.float() - target.float())**2`` averaged over all non-batch dims then the
    batch, divided by the accumulation steps.  ``sigmas`` as returned by the spec ([B,S,1], constant per sample)."""
    per_sample_sigma = sigmas.reshape(sigmas.shape[0], -1)[:, 0].float()
    weights = diffusion_utils.compute_loss_weighting_for_sd3(flow_weighting_scheme, per_sample_sigma).float().contiguous()
    return _MSELossFunction.apply(pred, target, weights, 1.0 / gradient_accumulation_steps)


class MI355XSFTStep:
    """One LoRA SFT optimisation step on one rank (one process per GPU).  ``parallel`` is a
    ``finetrainers_amd.parallel.DataParallelBackend`` (or None for a single GPU).

    Data parallelism (reference: ``apply_ddp`` -> ``replicate``, parallel/ptd.py:462-463): on construction the LoRA parameters are
    broadcast from rank 0 (DDP broadcasts module state the same way), every step each rank runs its own samples, and the LoRA
    gradients are averaged bucket by bucket WHILE the backward is still running (``GradBucketReducer``).  With
    ``gradient_accumulation_steps`` > 1 only the last micro-step of a group applies AdamW; every micro-step exchanges and clips, as the
    reference loop does (trainer.py:479-503), unless ``no_sync_accumulation``."""

    def __init__(self, transformer, specification, lr: float = 5e-5, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4,
                 max_grad_norm: float = 1.0, flow_weighting_scheme: str = "none", flow_logit_mean: float = 0.0, flow_logit_std: float = 1.0,
                 flow_mode_scale: float = 1.29, parallel=None, generator: Optional[torch.Generator] = None,
                 gradient_accumulation_steps: int = 1, grad_bucket_blocks: int = 7, lr_scheduler=None, compute_posterior: bool = True, no_sync_accumulation: bool = False):
        if transformer.lora_A is None:
            raise ValueError("attach a LoRA adapter first (transformer.add_adapter)")
        if gradient_accumulation_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        self.transformer = transformer
        self.spec = specification
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.scheme = flow_weighting_scheme
        self.flow_logit_mean, self.flow_logit_std, self.flow_mode_scale = flow_logit_mean, flow_logit_std, flow_mode_scale
        self.parallel = parallel
        self.generator = generator
        self.gradient_accumulation_steps = gradient_accumulation_steps
        # ``finetrainers_amd.utils.lr_schedule.LRSchedule`` (or anything with current_lr() / step()): its rate is read for every optimiser step
        # and it is stepped right after, as trainer.py:500-503 does with the LambdaLR
        self.lr_scheduler = lr_scheduler
        # False = the batches carry the VAE posterior's moments (what --enable_precomputation stores, trainer.py:374) and every step samples them
        self.compute_posterior = compute_posterior
        # Data parallelism x gradient accumulation: the reference wraps the model with replicate() and never enters no_sync (trainer.py:479-503,
        # "TODO revisit no_sync"), so EVERY backward all-reduces the accumulated .grad and the per-backward clip sees the same averaged sums on
        # all ranks.  That is the default here too; True skips the exchange (and the clip) on the non-stepping micro-steps -- equal results
        # whenever no partial sum exceeds max_grad_norm, half the traffic.
        self.no_sync_accumulation = no_sync_accumulation
        self._micro_step = 0
        dev = transformer.device
        transformer._assert_flat_aliasing()
        # flat fp32 optimiser state matching transformer.lora_flat = [A | B]
        self.n_a, self.n_b = transformer._lora_A_full.numel(), transformer._lora_B_full.numel()  # storage size (rank padded to a multiple of 64)
        self.exp_avg = torch.zeros(self.n_a + self.n_b, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self._scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.reducer = None
        # MI355XParallelBackend.apply_ddp() installs the exchange ON THE MODEL (hook + finish, like DDP's reducer): a step that is handed such a model
        # leaves those hooks where they are -- the backward then returns with averaged gradients -- and must not bring a second exchange of its own
        self._model_owns_exchange = getattr(transformer, "_grad_bucket_hook", None) is not None and getattr(transformer, "_grad_bucket_finish", None) is not None
        if self._model_owns_exchange and parallel is not None and parallel.active:
            raise ValueError("MI355XSFTStep(parallel=...) on a model that apply_ddp() already wired for the gradient exchange: the gradients would be averaged "
                             "twice -- pass parallel=None (the model's own hooks do the exchange) or do not call apply_ddp")
        if parallel is not None and parallel.active:
            from .parallel import GradBucketReducer

            # replicas must start from the same adapter: add_adapter draws A from each process's own RNG
            parallel.broadcast_(transformer.lora_flat, src=0)
            transformer._lora_versions = None
            transformer.grad_bucket_blocks = grad_bucket_blocks
            self.reducer = GradBucketReducer(parallel)
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
             **spec_kwargs) -> Dict[str, Optional[torch.Tensor]]:
        """One micro-step.  Returns device scalars ``loss`` (already divided by the accumulation steps, as trainer.py:479 logs it) and
        ``grad_norm`` (None on the non-final micro-steps of an accumulation group, where no optimiser step happens)."""
        tr = self.transformer
        tr._assert_flat_aliasing()
        latents = latent_model_conditions["latents"]
        B = latents.shape[0]
        if sigmas is None:
            sigmas = self.sample_sigmas(B)
        gas = self.gradient_accumulation_steps
        self._micro_step += 1
        sync = self._micro_step % gas == 0  # last micro-step of the group: exchange gradients, clip, step (trainer.py:498)
        # 3. forward (trainer.py:452-461)
        pred, target, sig = self.spec.forward(
            transformer=tr, condition_model_conditions=dict(condition_model_conditions), latent_model_conditions=dict(latent_model_conditions),
            sigmas=sigmas, generator=self.generator, compute_posterior=self.compute_posterior, **spec_kwargs,
        )
        # 4. loss + backward (trainer.py:463-481): loss and d loss / d pred come out of one kernel and the DiT backward is seeded with
        # that gradient directly (what loss.backward() would hand it, without the unit-seed multiply)
        per_sample_sigma = sig.reshape(B, -1)[:, 0].float()
        weights = diffusion_utils.compute_loss_weighting_for_sd3(self.scheme, per_sample_sigma).float().contiguous()
        loss, dpred = ops.mse_loss(pred.detach().contiguous(), target.contiguous(), weights, want_grad=True, grad_scale=1.0 / gas)
        loss = loss.reshape(()) / gas if gas > 1 else loss.reshape(())
        exchange = self.reducer is not None and (sync or not self.no_sync_accumulation)
        prev_hook, prev_fin = tr._grad_bucket_hook, tr._grad_bucket_finish
        # Who owns the exchange is decided PER STEP, not at construction: apply_ddp() may be called on the model after this object was built (a step
        # that then blanked the model's hooks for its backward would leave the replicas to diverge silently).
        model_owns = prev_hook is not None and prev_fin is not None
        if model_owns and self.reducer is not None:
            raise RuntimeError("MI355XSFTStep has its own gradient exchange (parallel=...) and the model now carries apply_ddp()'s hooks as well: the gradients "
                               "would be averaged twice -- build the step with parallel=None or do not call apply_ddp on this model")
        self._model_owns_exchange = model_owns
        if not model_owns:  # this step's own reducer (or none): install for the duration of the backward, then put back what was there
            tr._grad_bucket_hook = self.reducer.bucket_ready if exchange else None
            tr._grad_bucket_finish = None
        try:
            pred.backward(dpred)  # DP: buckets of finished blocks are all-reduced (AVG) on RCCL's stream while this still runs
        except BaseException:
            if exchange:
                self.reducer.abort()  # buckets issued before the failure: drained and dropped, never carried into the next step
            raise
        finally:
            tr._grad_bucket_hook, tr._grad_bucket_finish = prev_hook, prev_fin
        if not sync:
            # the reference clips after every backward (trainer.py:487-492), i.e. also the partial sums of an accumulation window -- in place
            # on .grad (under DP on the all-reduced sums, see no_sync_accumulation)
            if exchange:
                self.reducer.finish()
            if self.max_grad_norm and self.max_grad_norm > 0 and (self.reducer is None or exchange):
                ops.clip_grad_norm_(self._flat_grad(tr.lora_A.grad, tr.lora_B.grad), self.max_grad_norm, scratch=self._scratch)
            return {"loss": loss.detach(), "grad_norm": None}
        gflat = self._flat_grad(tr.lora_A.grad, tr.lora_B.grad)
        if self.reducer is not None:
            self.reducer.finish()
        # 5-6. clip (utils/torch.py:99-161) + AdamW (optimizer.py:117-125), fused over the flat buffer
        self.step_count += 1
        grad_norm = self._clip_adamw(gflat)
        tr.lora_A.grad = None  # optimizer.zero_grad(set_to_none=True) (trainer.py:503)
        tr.lora_B.grad = None
        return {"loss": loss.detach(), "grad_norm": grad_norm}

    # ---- resume (reference: utils/state_checkpoint.py saves optimizer + scheduler state next to the model's) ------------------------------
    def state_dict(self) -> Dict[str, Any]:
        """Optimiser-side state of the fused step: AdamW moments laid out like ``transformer.lora_flat`` ([A | B], rank-padded storage),
        the step counters and the schedule clock.  The LoRA parameters themselves travel in ``transformer.state_dict()``."""
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "micro_step": self._micro_step,
                "lr_scheduler": None if self.lr_scheduler is None else self.lr_scheduler.state_dict(),
                "hyper": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay, "max_grad_norm": self.max_grad_norm}}

    @torch.no_grad()
    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        if sd["exp_avg"].numel() != self.exp_avg.numel():
            raise ValueError(f"optimizer state holds {sd['exp_avg'].numel()} values, the attached adapter needs {self.exp_avg.numel()}")
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.step_count, self._micro_step = int(sd["step"]), int(sd.get("micro_step", 0))
        if self.lr_scheduler is not None and sd.get("lr_scheduler") is not None:
            self.lr_scheduler.load_state_dict(sd["lr_scheduler"])

    def _flat_grad(self, ga: torch.Tensor, gb: torch.Tensor) -> torch.Tensor:
        gflat = self.transformer._grad_flat
        if gflat is not None and ga.data_ptr() == gflat.data_ptr() and gb.data_ptr() == gflat.data_ptr() + 4 * self.n_a:
            return gflat  # .grad IS the flat buffer the kernels wrote (the usual case)
        if self.reducer is not None:
            raise RuntimeError("data-parallel step: lora_A.grad / lora_B.grad are not the backend's flat gradient buffer (foreign .grad tensors "
                               "were installed); the bucketed exchange would have missed them")
        tr = self.transformer  # foreign .grad tensors (single GPU only): lay them out like the (rank-padded) parameter storage
        flat = torch.zeros(self.n_a + self.n_b, dtype=torch.float32, device=ga.device)
        flat[:self.n_a].view_as(tr._lora_A_full)[:, :, :tr.lora_rank, :].copy_(ga)
        flat[self.n_a:].view_as(tr._lora_B_full)[:, :, :, :tr.lora_rank].copy_(gb)
        return flat

    def _clip_adamw(self, gflat: torch.Tensor) -> torch.Tensor:
        tr = self.transformer
        gn = torch.empty(1, dtype=torch.float32, device=gflat.device)
        lr = self.lr if self.lr_scheduler is None else self.lr_scheduler.current_lr()
        ops.clip_adamw_step(tr.lora_flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, lr, self.betas, self.eps,
                            self.weight_decay, self.max_grad_norm, scratch=self._scratch, grad_norm_out=gn)
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        tr._lora_versions = None  # parameters changed in place by the library: refresh the bf16 working copies next forward
        return gn
