"""Wire formats on either side of the MI355X SFT step: what feeds the step and what it emits, byte-compatible
with the reference so that artefacts move freely between the two.

  * precomputed samples ....... ``<dir>/finetrainers-precomputed-data/{latent|condition}-{index}.pt`` -- one ``torch.save``d dict per
                                sample, keys per processor ``output_names`` (finetrainers/data/precomputation.py:13,413-420;
                                LTX keys: finetrainers/models/ltx_video/base_specification.py:123-128).  ``PrecomputedSampleFeeder`` is the
                                MI355X-side reader: a background thread loads and collates the next batches into pinned host memory and
                                copies them to HBM on a side stream, so the ~65 ms step never waits for the disk or the PCIe copy.
  * LoRA checkpoint ........... ``pytorch_lora_weights.safetensors`` with ``transformer.``-prefixed peft keys and the ``lora_config`` JSON
                                + ``format: pt`` metadata (finetrainers/trainer/sft_trainer/trainer.py:283-298,
                                finetrainers/utils/serialization.py:6-10); the reader mirrors
                                finetrainers/patches/dependencies/diffusers/peft.py:31-58.
  * base weights .............. a diffusers model directory ``<root>/transformer/diffusion_pytorch_model*.safetensors`` (+ index json),
                                read with ``safetensors`` alone (what ``LTXVideoTransformer3DModel.from_pretrained(root,
                                subfolder="transformer")`` reads, finetrainers/models/ltx_video/base_specification.py:173-190).
Host-side I/O only: nothing here computes.
"""

from __future__ import annotations

import glob
import json
import os
import queue
import threading
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple

import torch

PRECOMPUTED_DATA_DIR = "finetrainers-precomputed-data"  # finetrainers/data/precomputation.py:13
LORA_WEIGHT_NAME = "pytorch_lora_weights.safetensors"   # diffusers' LORA_WEIGHT_NAME_SAFE (what save_lora_weights writes)
TRANSFORMER_WEIGHT_NAME = "diffusion_pytorch_model.safetensors"


# --------------------------------------------------------------------------------------------------------------------------
# precomputed samples
# --------------------------------------------------------------------------------------------------------------------------
def save_precomputed_item(item: Dict[str, Any], index: int, directory: str, data_type: str) -> str:
    """finetrainers/data/precomputation.py:413-415."""
    os.makedirs(directory, exist_ok=True)
    path = os.path.join(directory, f"{data_type}-{index}.pt")
    torch.save(item, path)
    return path


def load_precomputed_item(index: int, directory: str, data_type: str, map_location=None) -> Dict[str, Any]:
    """finetrainers/data/precomputation.py:418-420."""
    return torch.load(os.path.join(directory, f"{data_type}-{index}.pt"), map_location=map_location, weights_only=True)


class PrecomputedSampleFeeder:
    """Endless iterator of ``(condition_batch, latent_batch)`` over a reference precomputation directory.

    Index assignment follows ``PrecomputedOnceDataIterable`` (precomputation.py:350-384): rank r owns the contiguous range
    ``[r * n_per_rank, (r+1) * n_per_rank)`` with ``n_per_rank = max(1, n_items // world_size)`` and cycles through it.  Samples are
    grouped into batches of equal latent resolution (``ResolutionSampler`` semantics, data/sampler.py:6-58, keyed by the spec's
    ``_resolution_dim_keys``) and collated with the spec's ``collate_conditions`` / ``collate_latents``.
    MI355X-side design: the disk read, the collation and the host->device copy run ``prefetch`` batches ahead on a worker thread
    (pinned staging buffers, its own HIP stream); ``__next__`` only waits on an event."""

    MAX_PINNED_RINGS = 16  # distinct (tensor name, shape, dtype) staging rings kept page-locked at a time

    def __init__(self, save_dir: str, rank: int, world_size: int, batch_size: int, collate_conditions: Callable, collate_latents: Callable,
                 resolution_dim_keys: Optional[Dict[str, Tuple[int, ...]]] = None, device: Optional[torch.device] = None, prefetch: int = 2):
        self.dir = save_dir if os.path.basename(os.path.normpath(save_dir)) == PRECOMPUTED_DATA_DIR else os.path.join(save_dir, PRECOMPUTED_DATA_DIR)
        n_lat = len(glob.glob(os.path.join(self.dir, "latent-*.pt")))
        n_cond = len(glob.glob(os.path.join(self.dir, "condition-*.pt")))
        if n_lat == 0 or n_lat != n_cond:
            raise ValueError(f"{self.dir}: expected matching latent-*.pt / condition-*.pt files, found {n_lat} / {n_cond}")
        if n_lat <= rank:
            raise ValueError(f"Precomputed data directory does not contain enough items (required {rank + 1}, found {n_lat}).")
        self.rank, self.world_size, self.batch_size = rank, world_size, batch_size
        self.n_per_rank = max(1, n_lat // world_size)
        self.collate_conditions, self.collate_latents = collate_conditions, collate_latents
        self.dim_keys = resolution_dim_keys or {"latents": (2, 3, 4)}
        self.device = device
        self._use_gpu = device is not None and device.type == "cuda"
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, prefetch))
        self._pinned: Dict[Any, Dict[str, Any]] = {}  # persistent pinned staging buffers (worker thread only)
        self._last_slot = None
        self._stop = threading.Event()
        self._err: Optional[BaseException] = None
        self._thread = threading.Thread(target=self._worker, name="ftmi-feeder", daemon=True)
        self._thread.start()

    def __len__(self) -> int:
        return self.n_per_rank

    # ---- worker side ---------------------------------------------------------------------------------------------------
    def _sample_stream(self) -> Iterator[Tuple[Dict[str, Any], Dict[str, Any]]]:
        i = 0
        while True:
            index = self.rank * self.n_per_rank + i
            yield load_precomputed_item(index, self.dir, "condition", "cpu"), load_precomputed_item(index, self.dir, "latent", "cpu")
            i = (i + 1) % self.n_per_rank

    def _pinned_slot(self, key: str, like: torch.Tensor) -> torch.Tensor:
        """A persistent pinned staging buffer for tensor ``key`` of this shape, from a small ring: allocating pinned memory per batch
        (``Tensor.pin_memory()`` = hipHostMalloc + hipHostFree) synchronises with the device and cost the consumer ~10 % of a 36 ms step.  A slot
        is rewritten only after the host saw its last host-to-device copy complete."""
        ring_key = (key, tuple(like.shape), like.dtype)
        ring = self._pinned.pop(ring_key, None) or {"slots": [], "events": [], "next": 0}
        self._pinned[ring_key] = ring  # most recently used last (dicts keep insertion order)
        while len(self._pinned) > self.MAX_PINNED_RINGS:
            # multi-resolution / bucketed datasets, varying text lengths: every distinct shape has its own ring, so page-locked host memory would
            # grow without bound -- drop the least recently used ring once the copies out of it have completed
            old = self._pinned.pop(next(iter(self._pinned)))
            for ev in old["events"]:
                if ev is not None:
                    ev.synchronize()
        n = self._q.maxsize + 2
        if len(ring["slots"]) < n:
            ring["slots"].append(torch.empty(like.shape, dtype=like.dtype, pin_memory=True))
            ring["events"].append(None)
            i = len(ring["slots"]) - 1
        else:
            i = ring["next"]
            ring["next"] = (i + 1) % n
            if ring["events"][i] is not None:
                ring["events"][i].synchronize()  # (long done: the copy was issued n batches ago)
        self._last_slot = (ring, i)
        return ring["slots"][i]

    def _to_device(self, d: Dict[str, Any], stream) -> Dict[str, Any]:
        out = {}
        for k, v in d.items():
            if torch.is_tensor(v) and self._use_gpu:
                host = self._pinned_slot(k, v)
                host.copy_(v)
                ring, i = self._last_slot
                with torch.cuda.stream(stream):
                    out[k] = host.to(self.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(stream)
                ring["events"][i] = ev
            else:
                out[k] = v
        return out

    def _worker(self) -> None:
        try:
            stream = torch.cuda.Stream(device=self.device) if self._use_gpu else None
            buckets: Dict[Tuple[int, ...], List[Tuple[Dict[str, Any], Dict[str, Any]]]] = {}
            key_name = next(iter(self.dim_keys))
            for cond, lat in self._sample_stream():
                if self._stop.is_set():
                    return
                dims = tuple(lat[key_name].size(x) for x in self.dim_keys[key_name])
                bucket = buckets.setdefault(dims, [])
                bucket.append((cond, lat))
                if len(bucket) < self.batch_size:
                    continue
                conds, lats = zip(*buckets.pop(dims))
                cb, lb = self.collate_conditions(list(conds)), self.collate_latents(list(lats))
                cb, lb = self._to_device(cb, stream), self._to_device(lb, stream)
                ev = None
                if self._use_gpu:
                    ev = torch.cuda.Event()
                    ev.record(stream)
                while not self._stop.is_set():
                    try:
                        self._q.put((cb, lb, ev), timeout=0.1)
                        break
                    except queue.Full:
                        continue
        except BaseException as e:  # surfaced to the consumer: the step must not hang on a dead feeder
            self._err = e
            self._q.put(None)

    # ---- consumer side -------------------------------------------------------------------------------------------------
    def __iter__(self) -> "PrecomputedSampleFeeder":
        return self

    def __next__(self) -> Tuple[Dict[str, Any], Dict[str, Any]]:
        item = self._q.get()
        if item is None:
            raise RuntimeError("precomputed-sample feeder failed") from self._err
        cb, lb, ev = item
        if ev is not None:
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)  # device-side wait only: the copy was issued batches ago
            # The tensors were allocated on the worker's side stream.  The step that consumes them is asynchronous on the host, so the
            # caller may drop the dicts while forward / backward kernels that read them (the text states up to the end of the backward) are
            # still queued; without this the caching allocator would hand the blocks back to the side stream's pool at once and the
            # worker's next prefetch copy could overwrite them under those kernels.
            for d in (cb, lb):
                for v in d.values():
                    if torch.is_tensor(v) and v.is_cuda:
                        v.record_stream(cur)
        return cb, lb

    def close(self) -> None:
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._thread.join(timeout=5)


# --------------------------------------------------------------------------------------------------------------------------
# LoRA checkpoint
# --------------------------------------------------------------------------------------------------------------------------
def lora_config_metadata(rank: int, lora_alpha: float, target_modules) -> Dict[str, str]:
    """The metadata block of finetrainers/trainer/sft_trainer/trainer.py:285-291."""
    cfg = {"r": rank, "lora_alpha": lora_alpha, "init_lora_weights": True, "target_modules": target_modules}
    return {"lora_config": json.dumps(cfg, indent=4)}


def save_lora_weights(directory: str, transformer_state_dict: Dict[str, torch.Tensor], metadata: Optional[Dict[str, str]] = None) -> str:
    """``LTXPipeline.save_lora_weights(directory, state_dict, save_function=partial(safetensors_torch_save_function, metadata=...),
    safe_serialization=True)`` (ltx_video/base_specification.py:379-395): peft keys gain the ``transformer.`` prefix, metadata gains
    ``format: pt`` (utils/serialization.py:6-10)."""
    from safetensors.torch import save_file

    os.makedirs(directory, exist_ok=True)
    weights = {}
    for k, v in transformer_state_dict.items():
        if "lora_" not in k:
            raise ValueError(f"{k}: not a LoRA tensor (pass get_peft_model_state_dict / lora_state_dict output)")
        key = k.replace(".default.", ".")
        weights[key if key.startswith("transformer.") else f"transformer.{key}"] = v.detach().to("cpu").contiguous()
    md = dict(metadata or {})
    md["format"] = "pt"
    path = os.path.join(directory, LORA_WEIGHT_NAME)
    save_file(weights, path, md)
    return path


def load_lora_weights(path: str) -> Tuple[Dict[str, torch.Tensor], Dict[str, Any]]:
    """-> (peft-keyed state dict without the ``transformer.`` prefix, lora_config dict); mirror of
    finetrainers/patches/dependencies/diffusers/peft.py:31-58 up to the point where peft injects the adapter."""
    from safetensors import safe_open

    if os.path.isdir(path):
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if len(files) == 0:
            raise ValueError(f"No .safetensors files found in {path}.")
        path = files[0]
    sd = {}
    with safe_open(path, framework="pt") as f:
        md = f.metadata() or {}
        for k in f.keys():
            if k.startswith("transformer."):
                sd[k[len("transformer."):]] = f.get_tensor(k)
    if "lora_config" not in md:
        raise ValueError(f"{path}: no lora_config metadata (not written by finetrainers)")
    return sd, json.loads(md["lora_config"])


# --------------------------------------------------------------------------------------------------------------------------
# base weights
# --------------------------------------------------------------------------------------------------------------------------
def resolve_transformer_dir(pretrained_model_name_or_path: Optional[str], transformer_id: Optional[str] = None) -> str:
    """Local directory holding the transformer's safetensors: ``transformer_id`` itself or ``<root>/transformer`` (the two
    ``from_pretrained`` forms of base_specification.py:176-188).  Hub ids are not resolvable here (no network): raise."""
    for cand in ([transformer_id] if transformer_id else []) + ([os.path.join(pretrained_model_name_or_path, "transformer"), pretrained_model_name_or_path]
                                                                if pretrained_model_name_or_path else []):
        if cand and os.path.isdir(cand) and (glob.glob(os.path.join(cand, "*.safetensors")) or glob.glob(os.path.join(cand, "*.safetensors.index.json"))):
            return cand
    raise FileNotFoundError(
        f"no local diffusers transformer weights under {transformer_id or pretrained_model_name_or_path!r} "
        "(expected <root>/transformer/diffusion_pytorch_model*.safetensors); the MI355X backend never trains on random weights implicitly -- "
        "pass a local snapshot directory, or call load_diffusion_models(random_init_seed=...) explicitly for synthetic benchmarks")


def load_transformer_state_dict(directory: str) -> Dict[str, torch.Tensor]:
    """All tensors of a (possibly sharded) diffusers safetensors checkpoint directory."""
    from safetensors.torch import load_file

    index = glob.glob(os.path.join(directory, "*.safetensors.index.json"))
    if index:
        with open(index[0]) as f:
            files = sorted(set(json.load(f)["weight_map"].values()))
    else:
        files = [os.path.basename(p) for p in sorted(glob.glob(os.path.join(directory, "*.safetensors")))]
    sd: Dict[str, torch.Tensor] = {}
    for fn in files:
        sd.update(load_file(os.path.join(directory, fn)))
    return sd


def load_transformer_config(directory: str) -> Dict[str, Any]:
    p = os.path.join(directory, "config.json")
    if not os.path.exists(p):
        return {}
    with open(p) as f:
        return json.load(f)


# --------------------------------------------------------------------------------------------------------------------------
# training-state checkpoint in the reference's torch.distributed.checkpoint (DCP) layout
# --------------------------------------------------------------------------------------------------------------------------
# finetrainers/parallel/ptd.py:296-352: PTDCheckpointer saves  torch.distributed.checkpoint.save(states, checkpoint_id=<output_dir>/finetrainers_step_<N>)
# with  states = {"train_state": TrainState, "model": ModelWrapper([transformer]), "optimizer": OptimizerWrapper, "dataloader": ..., "lr_scheduler":
# LambdaLR}.  DCP stores the nested state dicts of these Stateful objects:
#   model.<fqn>                                   get_model_state_dict(peft-wrapped transformer): base weights as <module>.base_layer.weight,
#                                                 adapters as <module>.lora_A.default.weight (frozen weights included)
#   optimizer.state.<fqn>.{step, exp_avg, exp_avg_sq}   and   optimizer.param_groups.<fqn>.<hyper-parameter>
#                                                 get_optimizer_state_dict(..., flatten_optimizer_state_dict=True) (optimizer.py:48-52)
#   lr_scheduler.{base_lrs, last_epoch, _step_count, ...}      LambdaLR.state_dict()
#   train_state.{step, observed_data_samples, global_avg_losses, global_max_losses, log_steps}     (state.py:23-38; the lists as torch.save bytes)
# The functions below produce / consume exactly that dictionary, so a run can be resumed across the two implementations: the MI355X step
# keeps ONE flat fp32 buffer per moment ([A | B], rank-padded), which is cut into the per-parameter tensors the reference's optimizer holds.
DCP_PREFIX = "finetrainers_step"
_ADAMW_GROUP_DEFAULTS = {"amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": False,
                         "decoupled_weight_decay": True}


def _lora_param_views(lora_rank: int, a_full: torch.Tensor, b_full: torch.Tensor, paths: List[str]):
    """(fqn, view) of every adapter tensor inside rank-padded [L, 8, r_pad, D] / [L, 8, D, r_pad] storage, in ``paths`` order."""
    L, n = a_full.shape[0], a_full.shape[1]
    assert len(paths) == L * n
    for idx, path in enumerate(paths):
        l, i = divmod(idx, n)
        yield f"{path}.lora_A.default.weight", a_full[l, i, :lora_rank, :]
        yield f"{path}.lora_B.default.weight", b_full[l, i, :, :lora_rank]


def training_state_dict(model_state_dict: Dict[str, torch.Tensor], lora_paths: List[str], lora_rank: int, exp_avg_a: torch.Tensor,
                        exp_avg_b: torch.Tensor, exp_avg_sq_a: torch.Tensor, exp_avg_sq_b: torch.Tensor, step_count: int, hyper: Dict[str, Any],
                        lr_scheduler_state: Optional[Dict[str, Any]] = None, train_state: Optional[Dict[str, Any]] = None,
                        dataloader_state: Optional[Dict[str, Any]] = None, dp_rank: int = 0) -> Dict[str, Any]:
    """The nested dictionary the reference hands to ``torch.distributed.checkpoint.save``.  ``exp_avg_a`` ... are the moments in the
    transformer's (rank-padded) parameter storage shapes; ``hyper``: lr, betas, eps, weight_decay; ``train_state``: step,
    observed_data_samples, global_avg_losses, global_max_losses, log_steps."""
    import io

    opt: Dict[str, Any] = {}
    m1 = dict(_lora_param_views(lora_rank, exp_avg_a, exp_avg_b, lora_paths))
    m2 = dict(_lora_param_views(lora_rank, exp_avg_sq_a, exp_avg_sq_b, lora_paths))
    for fqn in m1:
        opt[f"state.{fqn}.step"] = torch.tensor(float(step_count), dtype=torch.float32)
        opt[f"state.{fqn}.exp_avg"] = m1[fqn]
        opt[f"state.{fqn}.exp_avg_sq"] = m2[fqn]
    # The reference builds its optimizer over ALL model.parameters() (finetrainers/optimizer.py:36-38), frozen base weights included, so
    # get_optimizer_state_dict(flatten=True) carries param_groups.<fqn>.* for every parameter of the model (and state.* only for the trained
    # ones).  A checkpoint without those keys would make the reference's strict dcp.load raise "Missing key in checkpoint state_dict".
    for fqn in list(model_state_dict) + [f for f in m1 if f not in model_state_dict]:
        grp = {"lr": float(hyper["lr"]), "betas": tuple(float(b) for b in hyper["betas"]), "eps": float(hyper["eps"]), "weight_decay": float(hyper["weight_decay"]),
               **_ADAMW_GROUP_DEFAULTS, "initial_lr": float(hyper.get("initial_lr", hyper["lr"]))}
        for k, v in grp.items():
            opt[f"param_groups.{fqn}.{k}"] = v
    out: Dict[str, Any] = {"model": dict(model_state_dict), "optimizer": opt}
    if lr_scheduler_state is not None:
        out["lr_scheduler"] = dict(lr_scheduler_state)
    ts = dict(train_state or {})

    def blob(x):
        b = io.BytesIO()
        torch.save(list(x), b)
        return b

    out["train_state"] = {"step": torch.tensor(int(ts.get("step", step_count)), dtype=torch.int32),
                          "observed_data_samples": torch.tensor(int(ts.get("observed_data_samples", 0)), dtype=torch.int32),
                          "global_avg_losses": blob(ts.get("global_avg_losses", [])), "global_max_losses": blob(ts.get("global_max_losses", [])),
                          "log_steps": blob(ts.get("log_steps", []))}
    # PTDCheckpointer.states always holds the DPDataLoader, whose state_dict() is {"dp_rank_<r>": pickle.dumps(<StatefulDataLoader state>)}
    # (finetrainers/data/dataloader.py:27-40): the entry is always written -- an empty inner state (= "start the stream over", the only
    # state a precomputed-sample feeder has between epochs) unless the caller hands one over.
    import pickle

    out["dataloader"] = dict(dataloader_state) if dataloader_state is not None else {f"dp_rank_{int(dp_rank)}": pickle.dumps({})}
    return out


def lambda_lr_state(base_lr: float, last_epoch: int, current_lr: float) -> Dict[str, Any]:
    """``torch.optim.lr_scheduler.LambdaLR.state_dict()`` of the reference's scheduler after ``last_epoch`` steps (the lambda itself is not
    pickled: ``lr_lambdas: [None]``)."""
    return {"base_lrs": [float(base_lr)], "last_epoch": int(last_epoch), "_step_count": int(last_epoch) + 1, "_is_initial": False,
            "_get_lr_called_within_step": False, "_last_lr": [float(current_lr)], "lr_lambdas": [None]}


def save_training_state(output_dir: str, step: int, transformer, sft_step, train_state: Optional[Dict[str, Any]] = None,
                        dataloader_state: Optional[Dict[str, Any]] = None, dp_rank: int = 0) -> str:
    """``PTDCheckpointer.save`` for the MI355X step: writes ``<output_dir>/finetrainers_step_<step>/`` (DCP ``.metadata`` + ``*.distcp``).
    ``transformer``: MI355XLTXVideoTransformer3DModel with an adapter; ``sft_step``: MI355XSFTStep."""
    import torch.distributed.checkpoint as dcp

    tr = transformer
    n_a = tr._lora_A_full.numel()
    sched = getattr(sft_step, "lr_scheduler", None)
    lr_now = sft_step.lr if sched is None else sched.current_lr()
    sd = training_state_dict(
        {k: v.detach().cpu() for k, v in tr.state_dict().items()}, [p for p, _, _ in tr.lora_views()], tr.lora_rank,
        sft_step.exp_avg[:n_a].view_as(tr._lora_A_full).cpu(), sft_step.exp_avg[n_a:].view_as(tr._lora_B_full).cpu(),
        sft_step.exp_avg_sq[:n_a].view_as(tr._lora_A_full).cpu(), sft_step.exp_avg_sq[n_a:].view_as(tr._lora_B_full).cpu(),
        sft_step.step_count, {"lr": lr_now, "betas": sft_step.betas, "eps": sft_step.eps, "weight_decay": sft_step.weight_decay, "initial_lr": sft_step.lr},
        lr_scheduler_state=None if sched is None else lambda_lr_state(sched.base_lr, sched.last_epoch, lr_now),
        train_state=train_state, dataloader_state=dataloader_state, dp_rank=dp_rank)  # every data-parallel rank writes ITS dataloader entry (dp_rank_<r>)
    path = os.path.join(output_dir, f"{DCP_PREFIX}_{step}")
    dcp.save(sd, checkpoint_id=path)
    return path


def prune_template_to_checkpoint(tmpl: Dict[str, Any], checkpoint_dir: str, optional_prefixes) -> List[str]:
    """Remove from a nested DCP load template every leaf under one of ``optional_prefixes`` (dotted key paths) that the checkpoint's ``.metadata``
    does not list; returns the removed keys.  Everything else stays strict (a missing adapter or moment must still raise)."""
    import torch.distributed.checkpoint as dcp

    have = set(dcp.FileSystemReader(checkpoint_dir).read_metadata().state_dict_metadata.keys())
    removed: List[str] = []

    def walk(d, prefix):
        for k in list(d.keys()):
            key = f"{prefix}{k}"
            if isinstance(d[k], dict):
                walk(d[k], key + ".")
                if not d[k] and any((key + ".").startswith(p) or p.startswith(key + ".") for p in optional_prefixes):
                    del d[k]
            elif key not in have and any(key.startswith(p) for p in optional_prefixes):
                removed.append(key)
                del d[k]

    walk(tmpl, "")
    return removed


@torch.no_grad()
def load_training_state(checkpoint_dir: str, transformer, sft_step, dp_rank: int = 0) -> Dict[str, Any]:
    """``PTDCheckpointer.load``: reads a DCP directory written by either implementation into the MI355X transformer + step (base weights
    only if the checkpoint's differ in name set -- they are frozen --, adapters, both moments, the step counter, the schedule clock).
    Returns the ``train_state`` scalars."""
    import torch.distributed.checkpoint as dcp

    tr = transformer
    n_a = tr._lora_A_full.numel()
    # DCP loads INTO a template of the right structure: build it from fresh CPU tensors of the current shapes
    tmpl = training_state_dict(
        {k: torch.empty_like(v, device="cpu") for k, v in tr.state_dict().items()}, [p for p, _, _ in tr.lora_views()], tr.lora_rank,
        torch.zeros_like(tr._lora_A_full, device="cpu"), torch.zeros_like(tr._lora_B_full, device="cpu"),
        torch.zeros_like(tr._lora_A_full, device="cpu"), torch.zeros_like(tr._lora_B_full, device="cpu"),
        0, {"lr": sft_step.lr, "betas": sft_step.betas, "eps": sft_step.eps, "weight_decay": sft_step.weight_decay},
        lr_scheduler_state=lambda_lr_state(sft_step.lr, 0, sft_step.lr) if getattr(sft_step, "lr_scheduler", None) is not None else None, dp_rank=dp_rank)
    tmpl.pop("train_state")
    tmpl["train_state"] = {"step": torch.zeros((), dtype=torch.int32), "observed_data_samples": torch.zeros((), dtype=torch.int32)}
    # Load what the checkpoint HOLDS: a strict dcp.load raises on every template key the file does not have, and checkpoints differ in their
    # OPTIONAL parts -- `optimizer.param_groups.<fqn>.*` of the frozen base weights (written since round 4, absent before), the dataloader entry
    # (one per data-parallel rank: dp_rank_<r>), the scheduler.  Those parts of the template are pruned to the checkpoint's own key set first.
    prune_template_to_checkpoint(tmpl, checkpoint_dir, optional_prefixes=("optimizer.param_groups.", "dataloader.", "lr_scheduler."))
    dcp.load(tmpl, checkpoint_id=checkpoint_dir)
    tr.load_state_dict(tmpl["model"], strict=False)  # base weights are frozen: a checkpoint may carry only the adapters
    opt = tmpl["optimizer"]
    ea, eb = torch.zeros_like(tr._lora_A_full, device="cpu"), torch.zeros_like(tr._lora_B_full, device="cpu")
    qa, qb = torch.zeros_like(ea), torch.zeros_like(eb)
    steps = set()
    for (fqn, va), (_, vq) in zip(_lora_param_views(tr.lora_rank, ea, eb, [p for p, _, _ in tr.lora_views()]),
                                  _lora_param_views(tr.lora_rank, qa, qb, [p for p, _, _ in tr.lora_views()])):
        va.copy_(opt[f"state.{fqn}.exp_avg"])
        vq.copy_(opt[f"state.{fqn}.exp_avg_sq"])
        steps.add(int(opt[f"state.{fqn}.step"].item()))
    if len(steps) != 1:
        raise ValueError(f"optimizer state holds different step counts per parameter: {sorted(steps)}")
    sft_step.exp_avg[:n_a].copy_(ea.flatten())
    sft_step.exp_avg[n_a:].copy_(eb.flatten())
    sft_step.exp_avg_sq[:n_a].copy_(qa.flatten())
    sft_step.exp_avg_sq[n_a:].copy_(qb.flatten())
    sft_step.step_count = steps.pop()
    sched = getattr(sft_step, "lr_scheduler", None)
    if sched is not None and "lr_scheduler" in tmpl:
        sched.load_state_dict({"last_epoch": int(tmpl["lr_scheduler"]["last_epoch"]), "base_lrs": list(tmpl["lr_scheduler"]["base_lrs"])})
    tr._lora_versions = None  # parameters changed: refresh the bf16 working copies at the next forward
    return {"step": int(tmpl["train_state"]["step"]), "observed_data_samples": int(tmpl["train_state"]["observed_data_samples"])}
