"""One CogVideoX LoRA SFT optimisation step (reference loop: finetrainers/trainer/sft_trainer/trainer.py:430-503 with the CogVideoX
specification): sigma draw -> DDIM noising -> DiT forward -> velocity -> x0 -> 1 / (1 - alphas_cumprod) weighted MSE -> backward -> LoRA-gradient
average over the data-parallel ranks -> global-norm clip -> AdamW, the last two fused over the model-wide flat LoRA buffer.  First cut: the
gradients of a block are all-reduced (AVG, asynchronously on RCCL's stream) as soon as that block's backward has produced them, while the earlier
blocks still compute; the step waits for the outstanding collectives before the flat gradient is assembled."""

from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from ..utils import diffusion as diffusion_utils
from .model import MI355XCogVideoXTransformer3DModel
from .specification import MI355XCogVideoXSpecOps


class MI355XCogVideoXSFTStep:
    def __init__(self, transformer: MI355XCogVideoXTransformer3DModel, spec: Optional[MI355XCogVideoXSpecOps] = None, lr: float = 5e-5,
                 betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 1e-4, max_grad_norm: float = 1.0, parallel=None,
                 generator: Optional[torch.Generator] = None, lr_scheduler=None, grad_bucket_blocks: int = 5):
        if transformer.lora_flat is None:
            raise ValueError("attach a LoRA adapter first (transformer.add_adapter)")
        self.transformer, self.spec = transformer, spec or MI355XCogVideoXSpecOps()
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.parallel, self.generator, self.lr_scheduler = parallel, generator, lr_scheduler
        dev = transformer.device
        self.exp_avg = torch.zeros_like(transformer.lora_flat)
        self.exp_avg_sq = torch.zeros_like(transformer.lora_flat)
        self._scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        self.step_count = 0
        self.grad_bucket_blocks = grad_bucket_blocks  # native block stack: blocks per all-reduce bucket (2 x 5 x 4 x r x D fp32 = 39 MB at rank 128)
        if parallel is not None and parallel.active:
            parallel.broadcast_(transformer.lora_flat, src=0)  # replicas start from rank 0's adapter (DDP does the same)
        # utils/diffusion.py:75-81: the DDIM "sigma" table is scheduler.timesteps / num_train_timesteps, timesteps = 999 ... 0
        n = self.spec.scheduler.config.num_train_timesteps
        self.num_train_timesteps = n
        self.sigma_table = (torch.arange(n - 1, -1, -1, device=dev).float() / float(n))

    def sample_sigmas(self, batch_size: int) -> torch.Tensor:
        """utils/diffusion.py:107-114: uniform draw, index into the table."""
        u = torch.rand((batch_size,), device=self.transformer.device, generator=self.generator)
        return self.sigma_table[(u * self.num_train_timesteps).long()]

    def step(self, latents: torch.Tensor, encoder_hidden_states: torch.Tensor, sigmas: Optional[torch.Tensor] = None,
             noise: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        tr = self.transformer
        if sigmas is None:
            sigmas = self.sample_sigmas(latents.shape[0])
        pred, target, _ = MI355XCogVideoXSpecOps.forward(self.spec, tr, latents, encoder_hidden_states, sigmas, noise=noise, generator=self.generator)
        dp = self.parallel is not None and self.parallel.active
        pending = []

        def exchange(ga, gb):  # called from inside the backward, block by block (last block first): every rank issues the same sequence
            for t in (ga, gb):
                h = self.parallel.all_reduce_mean_async(t)
                if h is not None:
                    pending.append(h)

        for blk in tr.transformer_blocks:  # per-block Python composition: one exchange per block
            blk._grad_hook = exchange if dp else None
        # native block stack: the backward runs in ranges of grad_bucket_blocks blocks, each range's slice of the flat gradient is one bucket
        tr._grad_bucket_hook = (lambda lo, hi, ga, gb: exchange(ga, gb)) if dp else None
        tr.grad_bucket_blocks = self.grad_bucket_blocks
        try:
            loss = self.spec.loss_backward(pred, target, sigmas)
        except BaseException:
            # a backward that raised after issuing some buckets: every rank issued the same collectives, so they complete -- wait for them and
            # drop the handles (the next step must not race RCCL's stream on the gradient buffer, nor re-divide a tensor): GradBucketReducer.abort
            for work, _ in pending:
                try:
                    work.wait()
                except Exception:
                    pass
            pending.clear()
            raise
        finally:
            tr._grad_bucket_hook = None
            for blk in tr.transformer_blocks:
                blk._grad_hook = None
        for work, div in pending:  # device-side wait on RCCL; gloo: host wait + divide
            work.wait()
            if div is not None:
                div.div_(self.parallel.world_size)
        self.buckets_issued = len(pending) // 2
        gflat = tr.flat_lora_grad()
        self.step_count += 1
        lr = self.lr if self.lr_scheduler is None else self.lr_scheduler.current_lr()
        gn = torch.empty(1, dtype=torch.float32, device=gflat.device)
        ops.clip_adamw_step(tr.lora_flat, gflat, self.exp_avg, self.exp_avg_sq, self.step_count, lr, self.betas, self.eps, self.weight_decay,
                            self.max_grad_norm, scratch=self._scratch, grad_norm_out=gn)
        tr._lora_versions = None  # parameters changed in place by the library: refresh the bf16 working copies next forward
        if self.lr_scheduler is not None:
            self.lr_scheduler.step()
        for blk in tr.transformer_blocks:
            blk.lora_A.grad = blk.lora_B.grad = None
        return {"loss": loss.detach(), "grad_norm": gn}
