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


# ROCr reads HSA_ENABLE_IPC_MODE_LEGACY at hsa_init and RCCL reads NCCL_* when its first communicator is created: both have to be in the environment
# BEFORE this process touches the GPU, so they are set at import (a launcher that already exported them wins).  `describe()` reports what is in the
# environment AND whether HIP was already initialised when this module was imported (in which case the HSA setting came too late to matter here).
_HIP_WAS_INITIALISED_AT_IMPORT = bool(getattr(torch.cuda, "is_initialized", lambda: False)())
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
# Defaults for the 29-59 MB gradient buckets on 8 fully connected xGMI peers (a ring is per-link bound, 7 links x ~153 GB/s): RCCL's
# tuner is left in charge (it picks among ring / tree / direct by message size and topology) unless FTMI_RCCL_ALGO / FTMI_RCCL_PROTO pin a choice; the
# protocol default for multi-MB messages is Simple (LL / LL128 trade bandwidth for latency below ~1 MB) and is pinned so that a tuner table tuned for
# small messages cannot pick LL128 for a 59 MB bucket.  What ran is recorded by `describe()` in every bench line.
# The Simple default is scoped to multi-rank jobs (WORLD_SIZE > 1 in the launcher's environment): a single-rank process -- whose only collectives are a few
# bytes of metrics -- keeps RCCL's own choice, and an explicit FTMI_RCCL_* always wins.
_MULTI_RANK_LAUNCH = int(os.environ.get("WORLD_SIZE", "1") or "1") > 1
for _var, _dst, _dflt in (("FTMI_RCCL_ALGO", "NCCL_ALGO", None), ("FTMI_RCCL_PROTO", "NCCL_PROTO", "Simple")):
    if os.environ.get(_var):
        os.environ[_dst] = os.environ[_var]
    elif _dflt is not None and _MULTI_RANK_LAUNCH:
        os.environ.setdefault(_dst, _dflt)


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
        self._rank_devices = None
        self.exercise_collectives = bool(exercise_collectives)
        if (self.world_size > 1 or self.exercise_collectives) and not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.pop("NCCL_P2P_DISABLE", None)  # never inherit the reference scripts' setting (keeps xGMI on)
            # (HSA_ENABLE_IPC_MODE_LEGACY and the NCCL_ALGO / NCCL_PROTO pins are set at module import, before the GPU is touched)
            dist.init_process_group(backend=self.backend, rank=self.rank, world_size=self.world_size,
                                    timeout=datetime.timedelta(seconds=timeout_s))
            self._owns_pg = True
        # (which device every rank sits on is NOT gathered here: a collective inside a constructor, with its failure swallowed on one rank, can leave the
        #  other ranks waiting -- callers that want the record in describe() call gather_rank_devices() explicitly, on every rank: bench.py does)

    def describe(self) -> Dict[str, object]:
        """What the exchange runs on, for the bench line and the multi-GPU tests' logs."""
        ver = None
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version()) if self.backend == "nccl" else None
        except Exception:
            pass
        out = {"backend": self.backend, "world_size": self.world_size, "rccl_version": ver, "algo": os.environ.get("NCCL_ALGO", "auto (RCCL tuner)"),
               "proto": os.environ.get("NCCL_PROTO", "auto (RCCL tuner)"), "p2p_disabled": os.environ.get("NCCL_P2P_DISABLE") == "1",
               "ipc_mode_legacy": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
               "hsa_env_set_before_hip_init": not _HIP_WAS_INITIALISED_AT_IMPORT}
        # what the COMMUNICATOR itself reports (not the environment): its size and, gathered once, the device every rank sits on -- the driver can
        # verify "N ranks on N distinct GPUs" from the bench line
        if dist.is_initialized():
            out["group_size"] = dist.get_world_size()
            out["group_rank"] = dist.get_rank()
            if self._rank_devices is not None:  # gathered ONCE, collectively, by gather_rank_devices() (describe() itself is rank-local)
                out["rank_devices"] = self._rank_devices
                out["distinct_devices"] = len({(d.get("device"), d.get("pci_bus_id"), d.get("uuid")) for d in self._rank_devices})
        return out

    def gather_rank_devices(self) -> None:
        """COLLECTIVE, opt-in: every rank of the group must call it (once, after construction).  Records which device each rank of the communicator sits on,
        for describe().  Errors propagate -- a rank that failed here must not let the others run on."""
        if not dist.is_initialized():
            return
        mine = {"rank": dist.get_rank(), "local_rank": self.local_rank, "device": str(self.device)}
        if self.device.type == "cuda":
            pr = torch.cuda.get_device_properties(self.device)
            mine.update(device_name=pr.name, device_index=self.device.index, pci_bus_id=getattr(pr, "pci_bus_id", None), uuid=str(getattr(pr, "uuid", "")))
        got = [None] * dist.get_world_size()
        dist.all_gather_object(got, mine)
        self._rank_devices = got

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

    # ---- the in-library exchange (include/ftmi355.h: ftmi_allreduce_*) ----------------------------------------------------------------------------
    def native_exchange(self):
        """COLLECTIVE, opt-in (``FTMI_NATIVE_ALLREDUCE=1`` or an explicit call on every rank): a communicator owned by libftmi355.so -- RCCL looked up
        with dlopen, its own communication stream, event hand-over with the compute stream -- instead of ``torch.distributed``'s.  Rank 0 obtains the
        rendezvous id, the process group carries its 128 bytes to the other ranks.  Returns the ``ftmi_exchange`` handle (cached); GradBucketReducer
        then issues its buckets through ``ftmi_allreduce_bucket`` / ``ftmi_allreduce_wait``.  GPU ranks only."""
        if getattr(self, "_native_ex", None) is not None:
            return self._native_ex
        if self.device.type != "cuda":
            raise RuntimeError("the in-library exchange runs on RCCL: GPU ranks only (CPU ranks use gloo through torch.distributed)")
        import ctypes

        from . import _lib

        lib = _lib.load()
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            _lib.check(lib.ftmi_allreduce_unique_id(ident), "ftmi_allreduce_unique_id")
        if self.world_size > 1:
            box = [ident.raw if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            ident = ctypes.create_string_buffer(box[0], 128)
        ex = ctypes.c_void_p()
        torch.cuda.set_device(self.device)
        _lib.check(lib.ftmi_allreduce_init(ident, self.rank, self.world_size, ctypes.byref(ex)), "ftmi_allreduce_init")
        self._native_ex = ex
        return ex

    def destroy(self) -> None:
        if getattr(self, "_native_ex", None) is not None:
            from . import _lib

            _lib.load().ftmi_allreduce_destroy(self._native_ex)
            self._native_ex = None
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

    def __init__(self, backend: DataParallelBackend, native: Optional[bool] = None):
        self.backend = backend
        self._pending = []
        self.buckets_issued = 0
        # native: the buckets go through the library's own communicator (ftmi_allreduce_bucket / _wait: RCCL on the library's communication stream) instead
        # of torch.distributed's process group.  Default: FTMI_NATIVE_ALLREDUCE=1 on GPU ranks of an RCCL job; every rank must make the same choice.
        if native is None:
            native = os.environ.get("FTMI_NATIVE_ALLREDUCE", "0") not in ("", "0") and backend.active and backend.backend == "nccl" and backend.device.type == "cuda"
        self.native = bool(native)
        self._ex = backend.native_exchange() if self.native else None
        self._native_pending = False
        # measure_exposed: bracket finish() with events on the compute stream -- the time the step actually WAITS for the exchange (what the
        # backward did not cover); read with exposed_ms().  Off by default (two event records per step).
        self.measure_exposed = False
        self._exposed = []

    def bucket_ready(self, l_lo: int, l_hi: int, grad_a: torch.Tensor, grad_b: torch.Tensor) -> None:
        """Hook signature of ``MI355XLTXVideoTransformer3DModel._grad_bucket_hook``."""
        if self.native:
            from . import _lib

            lib, st = _lib.load(), torch.cuda.current_stream().cuda_stream
            for t in (grad_a, grad_b):
                if not (t.is_contiguous() and t.dtype == torch.float32):
                    raise ValueError("the in-library exchange reduces contiguous fp32 slices of the flat gradient buffer in place")
                _lib.check(lib.ftmi_allreduce_bucket(self._ex, t.data_ptr(), t.numel(), 1, st), "ftmi_allreduce_bucket")
            self._native_pending = True
            self.buckets_issued += 1
            return
        for t in (grad_a, grad_b):
            h = self.backend.all_reduce_mean_async(t)
            if h is not None:
                self._pending.append(h)
        self.buckets_issued += 1

    def finish(self) -> None:
        """Make the current stream wait for every outstanding bucket (device-side wait on RCCL; gloo: host wait + divide)."""
        ev = t_host = None
        if self.native:
            if self._native_pending:
                from . import _lib

                if self.measure_exposed:
                    ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                    ev[0].record()
                _lib.check(_lib.load().ftmi_allreduce_wait(self._ex, torch.cuda.current_stream().cuda_stream), "ftmi_allreduce_wait")
                self._native_pending = False
                if ev is not None:
                    ev[1].record()
                    self._exposed.append(ev)
            return
        if self.measure_exposed and self._pending:
            if torch.cuda.is_available() and self.backend.device.type == "cuda":
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            else:  # CPU ranks (gloo): the wait is a host wait
                import time

                t_host = time.perf_counter()
        for work, div in self._pending:
            work.wait()
            if div is not None:
                div.div_(self.backend.world_size)
        self._pending.clear()
        if ev is not None:
            ev[1].record()
            self._exposed.append(ev)
        elif t_host is not None:
            import time

            self._exposed.append((time.perf_counter() - t_host) * 1e3)

    def exposed_ms(self) -> Optional[float]:
        """Mean time per step the compute stream spent waiting in finish() (call after a synchronize); None if nothing was measured."""
        if not self._exposed:
            return None
        t = [e if isinstance(e, float) else e[0].elapsed_time(e[1]) for e in self._exposed]
        self._exposed.clear()
        return sum(t) / len(t)

    def abort(self) -> None:
        """A backward that raised after issuing some buckets: wait for the collectives already in flight (every rank issued them, so they
        complete) and forget them -- the next step must neither wait on stale handles nor re-divide their tensors."""
        if self.native and self._native_pending:
            from . import _lib

            _lib.load().ftmi_allreduce_wait(self._ex, torch.cuda.current_stream().cuda_stream)  # the buckets already on the wire complete (every rank issued them)
            self._native_pending = False
        for work, _ in self._pending:
            try:
                work.wait()
            except Exception:
                pass
        self._pending.clear()


# ----------------------------------------------------------------------------------------------------------------------------------------------
# The reference's ``BaseParallelBackend`` surface (finetrainers/parallel/base.py:9-115), so that ``--parallel_backend mi355x`` selects this
# backend in the unmodified trainer (selection: parallel/__init__.py:17; use: trainer/sft_trainer/trainer.py:185-189, 210-228, 336-358, 512-528).
# ----------------------------------------------------------------------------------------------------------------------------------------------
from contextlib import contextmanager  # noqa: E402
from typing import Any, List  # noqa: E402

from .utils.reference_base import reference_class  # noqa: E402

_RefBaseParallelBackend = reference_class("finetrainers.parallel.base", "BaseParallelBackend")


class _StandaloneParallelBase:
    """``BaseParallelBackend`` (parallel/base.py:9-60) where the reference package is not installed: tracker plumbing only."""

    def __init__(self):
        self.tracker = None

    def initialize_trackers(self, trackers: List[str], experiment_name: str, config: Dict[str, Any], log_dir: str):
        self.tracker = _ListTracker() if self.is_main_process else _NullTracker()

    def log(self, metrics: Dict[str, Any], step: int) -> None:
        if self.is_main_process and self.tracker is not None:
            self.tracker.log(metrics, step)


class _NullTracker:
    def log(self, metrics, step):
        pass

    def finish(self):
        pass


class _ListTracker(_NullTracker):
    """Stand-in for the reference's trackers (wandb / none) where they are not installed: keeps what was logged."""

    def __init__(self):
        self.records = []

    def log(self, metrics, step):
        self.records.append((int(step), dict(metrics)))


class MI355XCheckpointer:
    """``PTDCheckpointer`` (parallel/ptd.py:296-420) for the MI355X step: same constructor keywords, ``save(step, force, _device=, _is_main_process=)``
    / ``load(step)``, same directory naming and purge rule; the files are the reference's DCP layout (``wire.save_training_state``)."""

    def __init__(self, dataloader=None, model_parts=None, optimizers=None, schedulers=None, states: Optional[Dict[str, Any]] = None,
                 checkpointing_steps: int = 500, checkpointing_limit: Optional[int] = None, output_dir: str = ".", enable: bool = True,
                 _callback_fn=None, _prefix: str = "finetrainers_step", sft_step=None) -> None:
        import pathlib

        self.dataloader, self.model_parts, self.states = dataloader, list(model_parts or []), dict(states or {})
        self.sft_step = sft_step if sft_step is not None else optimizers  # MI355XSFTStep (fused clip + AdamW state) or the trainer's own optimizer wrapper
        self.schedulers = schedulers
        self.checkpointing_steps, self.checkpointing_limit = checkpointing_steps, checkpointing_limit
        self.output_dir = pathlib.Path(output_dir)
        self.enable, self._callback_fn, self._prefix = enable, _callback_fn, _prefix

    def _dir(self, step: int):
        return self.output_dir / f"{self._prefix}_{step}"

    # ---- the unmodified SFTTrainer hands over ITS optimizer (trainer.py:309-320: `optimizers=self.optimizer`, an OptimizerWrapper over torch's AdamW) ----
    def _fused(self) -> bool:
        return self.sft_step is not None and hasattr(self.sft_step, "exp_avg")

    def _generic_states(self) -> Dict[str, Any]:
        """What PTDCheckpointer hands to torch.distributed.checkpoint (parallel/ptd.py:313-321) when the optimizer is torch's: Stateful wrappers around
        the model parts and the optimizer (the reference's OptimizerWrapper is used as it is), the dataloader if it is stateful, scheduler state."""
        from torch.distributed.checkpoint.state_dict import StateDictOptions, get_optimizer_state_dict, set_optimizer_state_dict
        from torch.distributed.checkpoint.stateful import Stateful

        parts = self.model_parts

        class _Model(Stateful):
            def state_dict(self_inner):
                return {k: v for m in parts for k, v in m.state_dict().items()}

            def load_state_dict(self_inner, sd):
                for m in parts:
                    m.load_state_dict(sd, strict=False)
                    if hasattr(m, "_lora_versions"):
                        m._lora_versions = None

        opt = self.sft_step

        class _Optim(Stateful):
            def _each(self_inner):
                opts = getattr(opt, "optimizers", None) or ([opt] if opt is not None else [])
                return list(zip(parts, opts))

            def state_dict(self_inner):
                o = StateDictOptions(flatten_optimizer_state_dict=True)
                return {k: v for m, oo in self_inner._each() for k, v in get_optimizer_state_dict(m, oo, options=o).items()}

            def load_state_dict(self_inner, sd):
                o = StateDictOptions(flatten_optimizer_state_dict=True)
                for m, oo in self_inner._each():
                    set_optimizer_state_dict(m, oo, optim_state_dict=sd, options=o)

        states = dict(self.states)
        states["model"] = _Model()
        if opt is not None:
            states["optimizer"] = opt if isinstance(opt, Stateful) else _Optim()
        if self.dataloader is not None and hasattr(self.dataloader, "state_dict") and hasattr(self.dataloader, "load_state_dict"):
            states["dataloader"] = self.dataloader
        if self.schedulers is not None and hasattr(self.schedulers, "get_lr_scheduler_state"):
            states.update(self.schedulers.get_lr_scheduler_state())
        return states

    def save(self, step: int = -1, force: bool = False, *, _device=None, _is_main_process: bool = True) -> Optional[str]:
        from . import wire

        if not self.enable or (not force and step % self.checkpointing_steps != 0):
            return None
        path = None
        try:
            if self._fused():
                ts = self.states.get("train_state")
                train_state = None if ts is None else {k: getattr(ts, k) for k in ("step", "observed_data_samples", "global_avg_losses", "global_max_losses", "log_steps") if hasattr(ts, k)}
                dl_state = self.dataloader.state_dict() if hasattr(self.dataloader, "state_dict") else None
                path = wire.save_training_state(str(self.output_dir), step, self.model_parts[0], self.sft_step, train_state=train_state, dataloader_state=dl_state,
                                                dp_rank=self._dp_rank())
            else:
                import torch.distributed.checkpoint as dcp

                path = str(self._dir(step))
                dcp.save(self._generic_states(), checkpoint_id=path)
            self._purge()
        finally:
            # the trained adapters are written whatever happened to the training-state files (the trainer's final save(force=True) is the only place
            # the model hook runs: trainer.py:563)
            if self._callback_fn is not None and _is_main_process and self.model_parts:
                self._callback_fn({k: v.detach().cpu() for k, v in self.model_parts[0].state_dict().items()})
        return path

    @staticmethod
    def _dp_rank() -> int:
        return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0

    def load(self, step: int = -1) -> bool:
        from . import wire

        if not self.enable or not self.output_dir.exists():
            return False
        if step == -1:
            found = sorted(self.output_dir.glob(f"{self._prefix}_*"), key=lambda x: int(x.name.split("_")[-1]))
            if not found:
                return False
            step = int(found[-1].name.split("_")[-1])
        if not self._dir(step).exists():
            return False
        if not self._fused():
            import torch.distributed.checkpoint as dcp
            from torch.distributed.checkpoint.stateful import Stateful

            states = self._generic_states()
            if step == 0:  # ptd.py:361-362: optimizers / schedulers do not exist yet at step 0
                states = {"model": states["model"]}
            keep = {k: v for k, v in states.items() if isinstance(v, Stateful)}
            dcp.load(states, checkpoint_id=str(self._dir(step)))
            states.update(keep)
            for k, v in states.items():  # plain entries (train_state scalars ...) were updated in the copy: hand them back
                if k in self.states and not isinstance(v, Stateful):
                    self.states[k] = v
            return True
        got = wire.load_training_state(str(self._dir(step)), self.model_parts[0], self.sft_step, dp_rank=self._dp_rank())
        ts = self.states.get("train_state")
        if ts is not None:
            for k, v in got.items():
                if hasattr(ts, k):
                    setattr(ts, k, v)
        return True

    def _purge(self) -> None:
        import shutil

        if not self.checkpointing_limit or self.checkpointing_limit <= 0:
            return
        found = sorted(self.output_dir.glob(f"{self._prefix}_*"), key=lambda x: int(x.name.split("_")[-1]), reverse=True)
        for stale in found[self.checkpointing_limit:]:
            shutil.rmtree(stale, ignore_errors=True)


class MI355XParallelBackend(_RefBaseParallelBackend if _RefBaseParallelBackend is not None else _StandaloneParallelBase):
    """``PytorchDTensorParallelBackend``'s constructor and methods (parallel/ptd.py:41-279) for the path this backend implements: batch-sharded
    data parallelism of the LoRA step over RCCL (one process per GPU).  Degrees this path does not implement are refused at construction."""

    def __init__(self, world_size: int, pp_degree: int = 1, dp_degree: int = 1, dp_shards: int = -1, cp_degree: int = 1, tp_degree: int = 1,
                 backend: str = "nccl", timeout: int = 180, logging_dir: Optional[str] = None, output_dir: Optional[str] = None,
                 gradient_accumulation_steps: Optional[int] = None, exercise_collectives: bool = False) -> None:
        super().__init__()
        import pathlib

        dp_shards = 1 if dp_shards in (-1, None) else dp_shards
        for name, degree in (("pp_degree", pp_degree), ("cp_degree", cp_degree), ("tp_degree", tp_degree), ("dp_shards", dp_shards)):
            if degree != 1:
                raise NotImplementedError(f"MI355XParallelBackend: {name}={degree} -- this backend implements replicated data parallelism of the LoRA step "
                                          "(SURVEY 8(e)); sharded full fine-tuning is finetrainers_amd.wan.sharding, CP / TP / PP are out of scope")
        if dp_degree != world_size:
            raise ValueError(f"World size {world_size} must equal dp_degree ({dp_degree}) for this backend")
        use_gpu = backend == "nccl" and torch.cuda.is_available()
        self._dp = DataParallelBackend(backend=backend if use_gpu or backend != "nccl" else "gloo", timeout_s=timeout, exercise_collectives=exercise_collectives)
        if self._dp.world_size != world_size:
            raise ValueError(f"WORLD_SIZE in the environment is {self._dp.world_size}, the backend was built for {world_size}")
        self._world_size, self._degree = world_size, dp_degree
        self._output_dir = pathlib.Path(output_dir) if output_dir is not None else None
        self._logging_dir = self._output_dir / logging_dir if output_dir is not None and logging_dir is not None else None
        self._gas = gradient_accumulation_steps
        self._mesh = None
        self.reducer: Optional[GradBucketReducer] = None

    # ---- model / data / optimizer preparation ------------------------------------------------------------------------------------------------
    def enable_determinism(self, seed: int) -> None:
        """parallel/ptd.py:93-95 + utils/torch.py enable_determinism: every rank the same seed (replicated weights, different data)."""
        import random

        random.seed(seed)
        torch.manual_seed(seed)
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(seed)

    def apply_ddp(self, model, device_mesh=None, grad_bucket_blocks: int = 7):
        """``replicate(model, bucket_cap_mb=100)`` (parallel/ptd.py:462-463) for the MI355X transformer: broadcast the adapters from rank 0 and
        install the bucketed, overlapped gradient exchange on the block-range backward (``GradBucketReducer``)."""
        if not hasattr(model, "lora_flat") or model.lora_A is None:
            raise ValueError("apply_ddp: attach a LoRA adapter first (only the adapters are replicated and exchanged)")
        if self._dp.active:
            self._dp.broadcast_(model.lora_flat, src=0)
            model._lora_versions = None
            model.grad_bucket_blocks = grad_bucket_blocks
            self.reducer = GradBucketReducer(self._dp)
            model._grad_bucket_hook = self.reducer.bucket_ready
            model._grad_bucket_finish = self.reducer.finish
        return model

    def apply_fsdp2(self, *args, **kwargs):
        raise NotImplementedError("MI355XParallelBackend: parameter sharding is finetrainers_amd.wan.sharding (Wan full fine-tune); the LoRA step replicates")

    def apply_context_parallel(self, *args, **kwargs):
        raise NotImplementedError("MI355XParallelBackend: context parallelism is out of scope (SURVEY 2a)")

    def prepare_model(self, model):
        return model

    def prepare_dataset(self, dataset):
        """parallel/ptd.py:136-143: rank-strided split of an iterable dataset (``split_dataset_by_node``)."""
        if self._degree == 1:
            return dataset
        if hasattr(dataset, "_data"):
            import datasets.distributed

            dataset._data = datasets.distributed.split_dataset_by_node(dataset._data, self.rank, self._world_size)
            return dataset
        raise TypeError("prepare_dataset: expected the reference's iterable dataset wrapper (an object with `_data`)")

    def prepare_dataloader(self, dataset, batch_size: int, num_workers: int, pin_memory: bool = False):
        """parallel/ptd.py:145-155.  The reference's DPDataLoader is a torchdata StatefulDataLoader; where torchdata is not installed a plain
        DataLoader with the same batching serves the step (no mid-epoch resume state)."""
        try:
            from finetrainers.data import DPDataLoader  # type: ignore

            return DPDataLoader(self.rank if self._degree > 1 else 0, dataset, batch_size=batch_size, num_workers=num_workers)
        except Exception:
            return torch.utils.data.DataLoader(dataset, batch_size=batch_size, num_workers=num_workers, pin_memory=pin_memory)

    def prepare_optimizer(self, optimizer, lr_scheduler):
        return optimizer, lr_scheduler

    def get_mesh(self, name: Optional[str] = None):
        """parallel/ptd.py:161-209 for a one-dimensional replicate mesh.  The reference trainer INDEXES the mesh it gets back --
        ``get_mesh()["dp_cp"]`` (trainer.py:512), ``get_mesh()["dp"]`` (trainer.py:595) --, and PTD makes those names exist by flattening the data
        dimensions onto the mesh (ptd.py:200-205): the same two flattened names are created here."""
        if self._degree == 1 or not dist.is_initialized():
            return None
        if self._mesh is None:
            dev = "cuda" if self._dp.backend == "nccl" else "cpu"
            mesh = torch.distributed.device_mesh.init_device_mesh(dev, mesh_shape=[self._degree], mesh_dim_names=["dp_replicate"])
            mesh[("dp_replicate",)]._flatten(mesh_dim_name="dp")     # data_replication_enabled: dp = dp_cp = (dp_replicate,)
            mesh[("dp_replicate",)]._flatten(mesh_dim_name="dp_cp")
            self._mesh = mesh
        if name is None:
            return self._mesh
        try:
            return self._mesh[name]
        except (KeyError, RuntimeError):
            return None if self._mesh.ndim == 0 else self._mesh

    def get_checkpointer(self, *args, **kwargs) -> MI355XCheckpointer:
        return MI355XCheckpointer(*args, **kwargs)

    # ---- the step's collectives (what DDP's reducer and parallel/utils.py:6-19 do) -----------------------------------------------------------
    def reduce_step_metrics(self, loss, grad_norm):
        return self._dp.reduce_step_metrics(loss, grad_norm)

    @property
    def data_parallel(self) -> DataParallelBackend:
        return self._dp

    def wait_for_everyone(self):
        return self._dp.wait_for_everyone()

    @contextmanager
    def main_process_first(self):
        if self.is_main_process:
            yield
            self.wait_for_everyone()
        else:
            self.wait_for_everyone()
            yield

    def destroy(self):
        if self.is_main_process and getattr(self, "tracker", None) is not None and hasattr(self.tracker, "finish"):
            self.tracker.finish()
        return self._dp.destroy()

    # ---- properties (parallel/ptd.py:214-279) ------------------------------------------------------------------------------------------------
    world_size = property(lambda self: self._world_size)
    rank = property(lambda self: self._dp.rank)
    local_rank = property(lambda self: self._dp.local_rank)
    is_main_process = property(lambda self: self._dp.rank == 0)
    is_local_main_process = property(lambda self: self._dp.local_rank == 0)
    device = property(lambda self: self._dp.device)
    pipeline_parallel_enabled = property(lambda self: False)
    data_parallel_enabled = property(lambda self: self._degree > 1)
    data_replication_enabled = property(lambda self: self._degree > 1)
    data_sharding_enabled = property(lambda self: False)
    context_parallel_enabled = property(lambda self: False)
    tensor_parallel_enabled = property(lambda self: False)
    _dp_degree = property(lambda self: self._degree)


def register_into_finetrainers() -> bool:
    """Make ``--parallel_backend mi355x`` resolve to ``MI355XParallelBackend`` in the reference (finetrainers/parallel/__init__.py:11-23): needs the
    enum member ``ParallelBackendEnum.MI355X = "mi355x"``; wraps ``get_parallel_backend_cls``.  False when the reference is not installed."""
    try:
        import finetrainers.parallel as ref  # type: ignore
    except Exception:
        return False
    member = getattr(ref.ParallelBackendEnum, "MI355X", None)
    if member is None:
        raise RuntimeError('finetrainers.parallel.ParallelBackendEnum lacks MI355X = "mi355x" (INTEGRATION.md)')
    original = ref.get_parallel_backend_cls

    def get_parallel_backend_cls(backend):
        return MI355XParallelBackend if backend == member else original(backend)

    ref.get_parallel_backend_cls = get_parallel_backend_cls
    return True
