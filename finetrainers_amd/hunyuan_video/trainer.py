"""One HunyuanVideo LoRA SFT optimisation step (reference loop: finetrainers/trainer/sft_trainer/trainer.py:430-503 with the HunyuanVideo specification):
noising -> DiT forward -> MSE -> backward -> LoRA-gradient average over the data-parallel ranks -> global-norm clip -> AdamW, the last two fused over ONE flat
fp32 buffer that the blocks' adapter Parameters are views of.  The gradients live in a second flat buffer of the same layout: the blocks' backward adds into
its views in place (no concatenation), and under data parallelism every ``grad_bucket_blocks`` finished blocks -- a contiguous slice, the backward walks the
buffer from its end -- are all-reduced (AVG) asynchronously on RCCL's stream while the earlier blocks still compute (315 MB at rank 64 in 8 buckets)."""

from __future__ import annotations

from typing import Dict, Optional

import torch

from .. import ops
from .model import MI355XHunyuanVideoTransformer3DModel
from .specification import MI355XHunyuanVideoSpecOps


class MI355XHunyuanVideoSFTStep:
    def __init__(self, transformer: MI355XHunyuanVideoTransformer3DModel, spec: Optional[MI355XHunyuanVideoSpecOps] = None, lr: float = 2e-5, betas=(0.9, 0.95),
                 eps: float = 1e-8, weight_decay: float = 1e-4, max_grad_norm: float = 1.0, guidance: float = 1.0, parallel=None,
                 generator: Optional[torch.Generator] = None, lr_scheduler=None, grad_bucket_blocks: int = 8):
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
        # gradient storage: one flat buffer laid out like the parameters; every block writes its two views
        self.gflat = torch.zeros_like(self.flat)
        self.grad_bucket_blocks = max(1, int(grad_bucket_blocks))
        self._spans = {}
        off = 0
        for blk in list(transformer.transformer_blocks) + list(transformer.single_transformer_blocks):
            na, nb = blk.lora_A.numel(), blk.lora_B.numel()
            blk._grad_a_view = self.gflat[off:off + na].view(blk.lora_A.shape)
            blk._grad_b_view = self.gflat[off + na:off + na + nb].view(blk.lora_B.shape)
            self._spans[id(blk)] = (off, off + na + nb)
            off += na + nb
        assert off == self.flat.numel()
        self.buckets_issued = 0
        self.exp_avg, self.exp_avg_sq = torch.zeros_like(self.flat), torch.zeros_like(self.flat)
        self._scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        self.step_count = 0

    def step(self, latents: torch.Tensor, conditions: Dict[str, torch.Tensor], sigmas: torch.Tensor, compute_posterior: bool = True,
             posterior_noise: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        pred, target, _ = self.spec.forward(self.transformer, latents, dict(conditions), sigmas, guidance=self.guidance, compute_posterior=compute_posterior,
                                            posterior_noise=posterior_noise, noise=noise, generator=self.generator)
        dp = self.parallel is not None and self.parallel.active
        gflat = self.gflat
        gflat.zero_()
        pending, ready = [], []
        tr = self.transformer
        blocks = list(tr.transformer_blocks) + list(tr.single_transformer_blocks)

        def flush():
            lo, hi = min(s_[0] for s_ in ready), max(s_[1] for s_ in ready)
            assert hi - lo == sum(s_[1] - s_[0] for s_ in ready), "finished blocks must form one contiguous slice of the flat gradient"
            h = self.parallel.all_reduce_mean_async(gflat[lo:hi])
            if h is not None:
                pending.append(h)
            ready.clear()

        def done(blk):  # called from inside the backward when a block's gradient is complete (last block first): every rank issues the same sequence
            ready.append(self._spans[id(blk)])
            if len(ready) >= self.grad_bucket_blocks or self._spans[id(blk)][0] == 0:
                flush()

        for blk in blocks:
            blk._grad_hook = done if dp else None
        try:
            loss = self.spec.loss_backward(pred, target)
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
            for blk in blocks:
                blk._grad_hook = None
        if dp and ready:
            flush()
        for work, div in pending:  # device-side wait on RCCL; gloo: host wait + divide
            work.wait()
            if div is not None:
                div.div_(self.parallel.world_size)
        self.buckets_issued = len(pending)
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
