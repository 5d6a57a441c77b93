"""One Wan-T2V FULL fine-tune optimisation step (BASELINE config 4; reference loop: finetrainers/trainer/sft_trainer/trainer.py:430-503 with
``--training_type full-finetune`` and FSDP-2, :171-181): posterior sample + flow-match noising -> DiT forward -> MSE -> backward producing every parameter
gradient -> fp32 reduce-scatter of each unit's gradient to its owners -> global-norm clip over the shards -> AdamW on the bf16 shards.

Parameters are sharded over the data-parallel ranks unit by unit (wan/fsdp.py): the blocks gather their parameters right before they compute, the next
block's all-gather and the previous block's reduce-scatter run on RCCL's stream meanwhile.  On one GPU the same code runs with whole "shards" and no
collectives."""

from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.distributed as dist

from .. import ops
from .fsdp import ParameterSharder
from .model import MI355XWanTransformer3DModel
from .specification import MI355XWanSpecOps

bf16 = torch.bfloat16


class MI355XWanFullFinetuneStep:
    def __init__(self, transformer: MI355XWanTransformer3DModel, spec: Optional[MI355XWanSpecOps] = None, lr: float = 1e-5, betas=(0.9, 0.95),
                 eps: float = 1e-8, weight_decay: float = 1e-4, max_grad_norm: float = 1.0, parallel=None, generator: Optional[torch.Generator] = None,
                 lr_scheduler=None, gradient_accumulation_steps: int = 1):
        if gradient_accumulation_steps < 1:
            raise ValueError("gradient_accumulation_steps must be >= 1")
        self.gradient_accumulation_steps, self._micro_step = gradient_accumulation_steps, 0
        self.transformer, self.spec = transformer, spec or MI355XWanSpecOps()
        self.lr, self.betas, self.eps, self.weight_decay, self.max_grad_norm = lr, betas, eps, weight_decay, max_grad_norm
        self.parallel, self.generator, self.lr_scheduler = parallel, generator, lr_scheduler
        active = parallel is not None and parallel.active
        world, rank = (parallel.world_size, parallel.rank) if active else (1, 0)
        backend = parallel.backend if active else "none"
        tr = transformer
        if active:  # replicas start from rank 0's weights (the reference loads the same checkpoint on every rank)
            parallel.broadcast_(tr.root.data, src=0)
            for blk in tr.blocks:
                parallel.broadcast_(blk.flat.data, src=0)
        names = ["root"] + [f"blocks.{i}" for i in range(len(tr.blocks))]
        self.sharder = ParameterSharder([tr.root.data] + [b.flat.data for b in tr.blocks], names, world, rank, backend, force_collectives=active and world == 1)
        self._reduce_norm = active
        # from here on each rank keeps only its shards: the modules' parameters ARE the shards (what an optimiser / checkpoint writer of this rank sees)
        tr.root.data = self.sharder.units[0].shard
        for i, blk in enumerate(tr.blocks):
            blk.flat.data = self.sharder.units[i + 1].shard
            blk._pre_forward = self._pre_forward
            blk._pre_backward = self._pre_backward
            blk._grad_hook = self._post_backward
        self._index = {id(b): i + 1 for i, b in enumerate(tr.blocks)}
        self.exp_avg = [torch.zeros_like(u.shard) for u in self.sharder.units]
        self.exp_avg_sq = [torch.zeros_like(u.shard) for u in self.sharder.units]
        dev = tr.device
        self._scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, dtype=torch.float32, device=dev)
        self._total = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_count = 0

    # ---- hooks driven by the blocks -----------------------------------------------------------------------------------------------------------
    def _pre_forward(self, blk) -> None:
        i = self._index[id(blk)]
        blk._param_src = self.sharder.acquire(i)
        self.sharder.prefetch(i + 1)

    def _pre_backward(self, blk) -> None:
        i = self._index[id(blk)]
        blk._param_src = self.sharder.acquire(i)
        if i > 1:
            self.sharder.prefetch(i - 1)
        blk.grad_flat = self.sharder.grad_buffer(i)
        blk.mark_updated()  # the gathered copy is new: rebuild the transposed weights of the input-gradient GEMMs

    def _post_backward(self, blk) -> None:
        self.sharder.scatter_grad(self._index[id(blk)])
        blk.grad_flat = None
        # the transposed weights of this block's input-gradient GEMMs are a full-size copy of its parameters: kept alive they would add
        # up to one unsharded bf16 model per rank, which is what the sharding exists to avoid
        blk._transposed = None

    # ---- the step -----------------------------------------------------------------------------------------------------------------------------
    def step(self, moments: torch.Tensor, encoder_hidden_states: torch.Tensor, latents_mean: torch.Tensor, latents_std: torch.Tensor,
             sigmas: torch.Tensor, posterior_noise: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        tr, sh = self.transformer, self.sharder
        tr._root_src = sh.acquire(0)
        tr.root_grad = sh.grad_buffer(0)
        pred, target, _ = self.spec.forward(tr, moments, encoder_hidden_states, sigmas, latents_mean, latents_std, posterior_noise=posterior_noise, noise=noise,
                                            generator=self.generator)
        gas = self.gradient_accumulation_steps
        self._micro_step += 1
        sync = self._micro_step % gas == 0  # last micro-step of the window: the optimiser steps (trainer.py:498)
        loss = self.spec.loss_backward(pred, target, grad_scale=1.0 / gas)
        sh.scatter_grad(0)
        tr.root_grad = None
        sh.finish_gradients()
        # global gradient norm over all shards of all ranks (utils/torch.py:99-161 on DTensor shards): per-unit sums of squares, one all-reduce
        self._total.zero_()
        for u in sh.units:
            self._total += ops.grad_sumsq(u.shard_grad, self._scratch)
        if self._reduce_norm:
            dist.all_reduce(self._total, op=dist.ReduceOp.SUM)
        gn = torch.empty(1, dtype=torch.float32, device=tr.device)
        if sync:
            self.step_count += 1
            lr = self.lr if self.lr_scheduler is None else self.lr_scheduler.current_lr()
            for u, m, v in zip(sh.units, self.exp_avg, self.exp_avg_sq):
                ops.adamw_bf16_step(u.shard, u.shard_grad, m, v, self.step_count, lr, self.betas, self.eps, self.weight_decay, sumsq=self._total,
                                    max_norm=self.max_grad_norm, grad_norm_out=gn)
            if self.lr_scheduler is not None:
                self.lr_scheduler.step()
            sh.zero_shard_grads()
        else:  # the reference clips after every backward, also where no optimiser step follows: the accumulated shard gradients are scaled in place
            for u in sh.units:
                ops.clip_by_sumsq_(u.shard_grad, self._total, self.max_grad_norm, grad_norm_out=gn)
        sh.release_all()
        tr._root_src = None
        for blk in tr.blocks:
            blk._param_src = None
        return {"loss": loss.detach(), "grad_norm": gn}

    def state_dict(self) -> Dict[str, object]:
        """This rank's optimiser state (moments of its shards) and counters; the parameters themselves are the modules' shards."""
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step": self.step_count, "micro_step": self._micro_step,
                "lr_scheduler": None if self.lr_scheduler is None else self.lr_scheduler.state_dict()}

    def load_state_dict(self, sd: Dict[str, object]) -> None:
        for dst, src in zip(self.exp_avg, sd["exp_avg"]):
            dst.copy_(src)
        for dst, src in zip(self.exp_avg_sq, sd["exp_avg_sq"]):
            dst.copy_(src)
        self.step_count, self._micro_step = int(sd["step"]), int(sd.get("micro_step", 0))
        if self.lr_scheduler is not None and sd.get("lr_scheduler") is not None:
            self.lr_scheduler.load_state_dict(sd["lr_scheduler"])

    @torch.no_grad()
    def gathered_state_dict(self) -> Dict[str, torch.Tensor]:
        """{diffusers parameter name: full bf16 tensor} assembled from all ranks (every rank must call it: it all-gathers unit by unit) -- what
        ``MI355XWanModelSpecification._save_model(directory, transformer, transformer_state_dict = ...)`` writes (trainer.py:293-305 gathers the FSDP state
        dict the same way before saving).  After sharding, the modules' own parameters are only this rank's slices."""
        full = self.gathered_parameters()
        tr = self.transformer
        out = dict(tr.root_layout.named_views(full["root"]))
        for i, blk in enumerate(tr.blocks):
            out.update({f"blocks.{i}.{k}": v for k, v in blk.layout.named_views(full[f"blocks.{i}"]).items()})
        return out

    @torch.no_grad()
    def gathered_parameters(self) -> Dict[str, torch.Tensor]:
        """{unit name: full bf16 parameters} assembled from all ranks (checkpointing, tests)."""
        out = {}
        sh = self.sharder
        for i, u in enumerate(sh.units):
            out[u.name] = sh.acquire(i).clone()
            sh.release_all()
        return out
