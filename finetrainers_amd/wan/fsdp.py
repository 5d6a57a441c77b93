"""Parameter sharding for the Wan full fine-tune (BASELINE config 4: "FSDP-2 param shard across 8 x MI355X").

What the reference does (finetrainers/parallel/ptd.py:466-499, trainer/sft_trainer/trainer.py:171-181): ``fully_shard`` on every transformer block and
then on the model, ``MixedPrecisionPolicy(param_dtype = bf16, reduce_dtype = fp32)``: each rank owns 1 / W of every unit's parameters (and their
optimiser state); a unit is all-gathered (bf16) right before its forward and again before its backward, with the next unit prefetched; after a unit's
backward its gradient is reduce-scattered (fp32, averaged) to the owning ranks; blocks are resharded after their forward except the last.

The MI355X design keeps that data flow but is built for one process per GPU over RCCL / xGMI and flat buffers:
  * a unit = one flat bf16 parameter buffer (wan/block.py, wan/model.py), padded to W x k elements; the local shard is the contiguous slice
    [rank k, (rank + 1) k) -- ONE all-gather and ONE reduce-scatter per unit and direction (93 MB bf16 / 186 MB fp32 per Wan-1.3B block), no per-tensor
    collectives, no DTensor bookkeeping;
  * two full-size parameter buffers rotate: while block i computes out of one, block i +- 1 is gathered into the other on RCCL's stream (the
    process group stream waits for the kernels already queued on the compute stream that still read the buffer, so reuse needs no host sync);
  * two full-size fp32 gradient buffers rotate the same way: block i's backward accumulates into one while block i + 1's reduce-scatter drains the other;
  * the root unit (embedders, output projection: 2 % of the parameters) is gathered once per step.
gloo (tests: CPU, or two ranks on one GPU) runs the same schedule synchronously with host staging.
Status: the schedule is tested at world size 2 over gloo and on a one-rank RCCL communicator; the overlap described above is the intended behaviour of
asynchronous collectives on the process group's stream and has not been traced on a multi-GPU node yet."""

from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

bf16 = torch.bfloat16


class ShardedUnit:
    def __init__(self, name: str, full_params: torch.Tensor, world: int, rank: int):
        self.name, self.numel = name, full_params.numel()
        self.k = ((self.numel + world - 1) // world + 63) // 64 * 64  # shard length, 128-byte granules
        self.padded = self.k * world
        dev = full_params.device
        shard = torch.zeros(self.k, dtype=full_params.dtype, device=dev)
        lo, hi = rank * self.k, min((rank + 1) * self.k, self.numel)
        if hi > lo:
            shard[: hi - lo].copy_(full_params[lo:hi])
        self.shard = shard                                     # the parameters this rank owns (and optimises)
        self.shard_grad = torch.zeros(self.k, dtype=torch.float32, device=dev)
        self.rs_out = torch.zeros(self.k, dtype=torch.float32, device=dev)
        self.full: Optional[torch.Tensor] = None               # the gathered parameters while the unit is unsharded
        self.full_grad: Optional[torch.Tensor] = None          # the fp32 gradient buffer its backward accumulates into
        self.gather_work = None
        self.rs_work = None
        self.rs_pending = False


class ParameterSharder:
    """Schedules all-gathers / reduce-scatters of ``units`` (root first, then the blocks in forward order)."""

    def __init__(self, unit_params: List[torch.Tensor], names: List[str], world: int, rank: int, backend: str, n_block_buffers: int = 2,
                 force_collectives: bool = False):
        self.world, self.rank, self.backend = world, rank, backend
        self._local = world == 1 and not force_collectives  # force_collectives: run the gathers / scatters even on a one-rank communicator (RCCL smoke)
        self.units = [ShardedUnit(n, p, world, rank) for n, p in zip(names, unit_params)]
        dev = unit_params[0].device
        dtype = unit_params[0].dtype
        blocks = self.units[1:]
        big = max((u.padded for u in blocks), default=0)
        self._param_bufs = [torch.zeros(big, dtype=dtype, device=dev) for _ in range(n_block_buffers if blocks else 0)]
        self._grad_bufs = [torch.zeros(big, dtype=torch.float32, device=dev) for _ in range(n_block_buffers if blocks else 0)]
        self._grad_buf_user: List[Optional[ShardedUnit]] = [None] * len(self._grad_bufs)
        self._param_buf_user: List[Optional[ShardedUnit]] = [None] * len(self._param_bufs)
        root = self.units[0]
        self._root_param = torch.zeros(root.padded, dtype=dtype, device=dev)
        self._root_grad = torch.zeros(root.padded, dtype=torch.float32, device=dev)
        self.gathers_issued = self.scatters_issued = 0
        self._rs_order: Dict[int, int] = {}
        self._rs_counter = 0

    # ---- collectives ------------------------------------------------------------------------------------------------------------------------
    def _all_gather(self, full: torch.Tensor, shard: torch.Tensor):
        self.gathers_issued += 1
        if self._local:
            full[: shard.numel()].copy_(shard)
            return None
        if self.backend == "nccl":
            return dist.all_gather_into_tensor(full, shard, async_op=True)
        host = torch.empty(full.numel() * full.element_size(), dtype=torch.uint8)  # gloo: bytes through the host, synchronous (tests)
        dist.all_gather_into_tensor(host, shard.detach().cpu().contiguous().view(torch.uint8))
        full.copy_(host.view(full.dtype))
        return None

    def _reduce_scatter_mean(self, out: torch.Tensor, full: torch.Tensor):
        self.scatters_issued += 1
        if self._local:
            out.copy_(full[: out.numel()])
            return None
        if self.backend == "nccl":
            return dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.AVG, async_op=True)
        host_out = torch.empty(out.numel(), dtype=torch.float32)
        dist.reduce_scatter_tensor(host_out, full.detach().cpu().contiguous(), op=dist.ReduceOp.SUM)
        out.copy_(host_out.div_(self.world))
        return None

    # ---- parameters -------------------------------------------------------------------------------------------------------------------------
    def _buffer_for(self, u: ShardedUnit) -> torch.Tensor:
        if u is self.units[0]:
            return self._root_param
        for i, user in enumerate(self._param_buf_user):  # already resident?
            if user is u:
                return self._param_bufs[i]
        # take the buffer whose user is furthest from u in schedule order (with two buffers: the one that is not u's neighbour in flight)
        idx = self.units.index(u)
        best, best_d = 0, -1
        for i, user in enumerate(self._param_buf_user):
            d = 1 << 30 if user is None else abs(self.units.index(user) - idx)
            if d > best_d:
                best, best_d = i, d
        old = self._param_buf_user[best]
        if old is not None:
            old.full, old.gather_work = None, None
        self._param_buf_user[best] = u
        return self._param_bufs[best]

    def prefetch(self, i: int) -> None:
        """Start gathering unit i (no-op when it is resident or in flight, or out of range)."""
        if i < 0 or i >= len(self.units):
            return
        u = self.units[i]
        if u.full is not None:
            return
        if self._local:  # nothing to gather: compute straight out of the (whole) shard
            u.full = u.shard
            return
        buf = self._buffer_for(u)
        u.full = buf[: u.padded]
        u.gather_work = self._all_gather(u.full, u.shard)

    def acquire(self, i: int) -> torch.Tensor:
        """The gathered parameters of unit i, valid for the kernels launched after this call."""
        self.prefetch(i)
        u = self.units[i]
        if u.gather_work is not None:
            u.gather_work.wait()  # device-side: the compute stream waits for RCCL's stream
            u.gather_work = None
        return u.full[: u.numel]

    def release_all(self) -> None:
        """Reshard everything (end of a step: the shards were updated, every gathered copy is stale)."""
        for u in self.units:
            u.full, u.gather_work = None, None
        self._param_buf_user = [None] * len(self._param_bufs)

    # ---- gradients --------------------------------------------------------------------------------------------------------------------------
    def grad_buffer(self, i: int) -> torch.Tensor:
        """A zeroed fp32 buffer for unit i's backward to accumulate into (the same one while the unit's backward is in progress)."""
        u = self.units[i]
        if u.full_grad is not None:
            return u.full_grad[: u.numel]
        if i == 0:
            buf = self._root_grad
        else:
            free = [b for b in range(len(self._grad_bufs)) if self._grad_buf_user[b] is None]
            # no free buffer: take the one whose reduce-scatter was issued first (it has had the whole of the last block's backward to drain)
            j = free[0] if free else min(range(len(self._grad_bufs)), key=lambda b: self._rs_order.get(id(self._grad_buf_user[b]), 0))
            old = self._grad_buf_user[j]
            if old is not None:
                self._finish_scatter(old)
            self._grad_buf_user[j] = u
            buf = self._grad_bufs[j]
        u.full_grad = buf[: u.padded]
        u.full_grad.zero_()
        return u.full_grad[: u.numel]

    def scatter_grad(self, i: int) -> None:
        """Unit i's gradient is final: start its fp32 averaged reduce-scatter into the owner's shard."""
        u = self.units[i]
        if u.full_grad is None:
            raise RuntimeError(f"unit {u.name}: no gradient buffer was handed out")
        u.rs_work = self._reduce_scatter_mean(u.rs_out, u.full_grad)
        u.rs_pending = True
        self._rs_counter += 1
        self._rs_order[id(u)] = self._rs_counter

    def _finish_scatter(self, u: ShardedUnit) -> None:
        if not u.rs_pending:
            if u.full_grad is not None:
                raise RuntimeError(f"unit {u.name}: gradient buffer reclaimed before its reduce-scatter was issued")
            return
        if u.rs_work is not None:
            u.rs_work.wait()
            u.rs_work = None
        u.shard_grad.add_(u.rs_out)  # += : micro-batches of an accumulation window add up on the owner
        u.rs_pending = False
        u.full_grad = None
        for b, user in enumerate(self._grad_buf_user):
            if user is u:
                self._grad_buf_user[b] = None

    def finish_gradients(self) -> None:
        """Wait for every outstanding reduce-scatter; afterwards ``shard_grad`` of every unit holds the averaged gradient of the owned slice."""
        for u in self.units:
            if u.rs_pending:
                self._finish_scatter(u)

    def zero_shard_grads(self) -> None:
        for u in self.units:
            u.shard_grad.zero_()
