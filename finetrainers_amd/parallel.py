"""Data-parallel backend for the MI355X SFT step: one process per GPU, RCCL over xGMI.

Restates the slice of finetrainers/parallel the DP path uses (``BaseParallelBackend`` parallel/base.py:9-115;
PTD backend parallel/ptd.py:41-279: ``init_process_group("nccl")``, ``replicate(bucket_cap_mb=100)`` DDP,
``split_dataset_by_node``; scalar reductions parallel/utils.py:6-19) -- retargeted, not translated:

  * the trainable state is ONE flat fp32 LoRA-gradient buffer (234.9 MB at r=64) laid out [A | B] with the layer axis leading, so
    the gradients of a contiguous block range are two contiguous slices.  The DiT backward runs in block ranges
    (``ftmi_ltx_backward_range``) and ``GradBucketReducer`` all-reduces (AVG) the two slices of each finished range right away,
    asynchronously on RCCL's stream, while the earlier blocks still compute -- the role of the c10d reducer's 100 MB buckets
    without parameter hooks (default 7 blocks per bucket: 4 buckets of 2 x 29.4 MB; only the last one is exposed).  Averaging is
    folded into the collective (``ReduceOp.AVG`` on RCCL, SUM + scale on gloo);
  * the three logging scalars (grad-norm mean, loss mean, loss max: trainer.py:512-518) are reduced in ONE small
    collective and stay on the device -- no ``.item()`` on the critical path;
  * ``NCCL_P2P_DISABLE`` from the reference's example scripts is never set: it would force RCCL off xGMI.
CP / TP / PP / FSDP are out of scope for this path.
"""

from __future__ import annotations

import datetime
import os
from typing import Dict, Optional

import torch
import torch.distributed as dist


class DataParallelBackend:
    def __init__(self, backend: Optional[str] = None, timeout_s: int = 300, device: Optional[torch.device] = None,
                 exercise_collectives: bool = False):
        """``exercise_collectives``: create the process group and run every collective even at world size 1 (a one-rank RCCL
        communicator) -- lets a single-GPU box execute the exact code path of the multi-GPU step."""
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world_size = int(os.environ.get("WORLD_SIZE", "1"))
        use_gpu = torch.cuda.is_available() and (backend is None or backend == "nccl")
        self.backend = backend or ("nccl" if use_gpu else "gloo")
        if device is None:
            device = torch.device("cuda", self.local_rank) if use_gpu else torch.device("cpu")
        self.device = device
        if use_gpu:
            torch.cuda.set_device(self.device)
        self._owns_pg = False
        self.exercise_collectives = bool(exercise_collectives)
        if (self.world_size > 1 or self.exercise_collectives) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
            os.environ.pop("NCCL_P2P_DISABLE", None)  # never inherit the reference scripts' setting (keeps xGMI on)
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world_size,
                                    timeout=datetime.timedelta(seconds=timeout_s))
            self._owns_pg = True

    # ---- properties mirroring BaseParallelBackend -------------------------------------------------------------
    @property
    def active(self) -> bool:
        """True when gradients have to be exchanged (more than one rank, or a one-rank group kept on purpose)."""
        return self.world_size > 1 or self.exercise_collectives

    @property
    def is_main_process(self) -> bool:
        return self.rank == 0

    @property
    def data_replication_enabled(self) -> bool:
        return self.world_size > 1

    @property
    def _dp_degree(self) -> int:
        return self.world_size

    # ---- collectives ---------------------------------------------------------------------------------------------
    @torch.no_grad()
    def all_reduce_mean_(self, flat: torch.Tensor) -> torch.Tensor:
        """In-place average of the flat gradient buffer over all ranks (DDP's gradient all-reduce)."""
        if not self.active:
            return flat
        if self.backend == "nccl":
            dist.all_reduce(flat, op=dist.ReduceOp.AVG)
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM)
            flat.div_(self.world_size)
        return flat

    @torch.no_grad()
    def all_reduce_mean_async(self, t: torch.Tensor):
        """Start averaging ``t`` over all ranks and return a handle for ``GradBucketReducer.finish``.  On RCCL the collective runs on
        the process group's own stream (ordered after the work already queued on the current stream), so it overlaps whatever the
        caller launches next; nothing blocks the host."""
        if not self.active:
            return None
        if self.backend == "nccl":
            return (dist.all_reduce(t, op=dist.ReduceOp.AVG, async_op=True), None)
        return (dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True), t)

    @torch.no_grad()
    def reduce_step_metrics(self, loss: torch.Tensor, grad_norm: torch.Tensor) -> Dict[str, torch.Tensor]:
        """trainer.py:512-518 (dist_mean(grad_norm), dist_mean(loss), dist_max(loss)) in one collective, no host sync:
        all-gather of [loss, grad_norm] per rank, then mean / max on the device."""
        pair = torch.stack([loss.reshape(()).float(), grad_norm.reshape(()).float()])
        if self.world_size == 1:
            return {"global_avg_loss": pair[0], "global_max_loss": pair[0], "grad_norm": pair[1]}
        if self.backend == "nccl":
            gathered = torch.empty(self.world_size, 2, dtype=torch.float32, device=pair.device)
            dist.all_gather_into_tensor(gathered, pair)
        else:  # gloo has no all-gather for device tensors: sum of one-hot rows (works on CPU and on a GPU alike)
            gathered = torch.zeros(self.world_size, 2, dtype=torch.float32, device=pair.device)
            gathered[self.rank] = pair
            dist.all_reduce(gathered, op=dist.ReduceOp.SUM)
        return {"global_avg_loss": gathered[:, 0].mean(), "global_max_loss": gathered[:, 0].max(), "grad_norm": gathered[:, 1].mean()}

    @torch.no_grad()
    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.active:
            dist.broadcast(t, src=src)
        return t

    def shard_indices(self, n: int):
        """Which of n samples this rank draws (``split_dataset_by_node`` semantics: rank-strided), ptd.py:136-143."""
        return range(self.rank, n, self.world_size)

    def wait_for_everyone(self) -> None:
        if self.world_size > 1:
            if self.backend == "nccl":
                dist.barrier(device_ids=[self.device.index])  # explicit device: no "guessing device" warning, no wrong-GPU barrier
            else:
                dist.barrier()

    def destroy(self) -> None:
        if self._owns_pg and dist.is_initialized():
            dist.destroy_process_group()
            self._owns_pg = False


class GradBucketReducer:
    """Bucketed, overlapped gradient exchange of the DP step -- the role of DDP's reducer (``replicate(bucket_cap_mb=100)``,
    finetrainers/parallel/ptd.py:462-463) without parameter hooks: the DiT backward runs in block ranges
    (``ftmi_ltx_backward_range``) and reports each range as soon as its LoRA gradients are final; the slices of the flat fp32
    gradient buffer are all-reduced (AVG) right away on RCCL's stream while the remaining blocks compute.  At r = 64 a bucket of 7
    blocks is 2 x 29.4 MB, i.e. 4 buckets per step (DDP's 100 MB cap would give 3); per-link xGMI time of the whole 235 MB exchange is
    ~3 ms against a ~60 ms step, and only the last bucket (the first blocks) is exposed.  Every rank issues the same collectives in
    the same order by construction (the bucket schedule is a function of L alone)."""

    def __init__(self, backend: DataParallelBackend):
        self.backend = backend
        self._pending = []
        self.buckets_issued = 0

    def bucket_ready(self, l_lo: int, l_hi: int, grad_a: torch.Tensor, grad_b: torch.Tensor) -> None:
        """Hook signature of ``MI355XLTXVideoTransformer3DModel._grad_bucket_hook``."""
        for t in (grad_a, grad_b):
            h = self.backend.all_reduce_mean_async(t)
            if h is not None:
                self._pending.append(h)
        self.buckets_issued += 1

    def finish(self) -> None:
        """Make the current stream wait for every outstanding bucket (device-side wait on RCCL; gloo: host wait + divide)."""
        for work, div in self._pending:
            work.wait()
            if div is not None:
                div.div_(self.backend.world_size)
        self._pending.clear()

    def abort(self) -> None:
        """A backward that raised after issuing some buckets: wait for the collectives already in flight (every rank issued them, so they
        complete) and forget them -- the next step must neither wait on stale handles nor re-divide their tensors."""
        for work, _ in self._pending:
            try:
                work.wait()
            except Exception:
                pass
        self._pending.clear()
