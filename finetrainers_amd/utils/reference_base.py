"""How the MI355X plugin classes attach to the reference's own class hierarchy.

The drop-in boundary is the reference's ``ModelSpecification`` / ``BaseParallelBackend`` interfaces.  When ``finetrainers`` is
importable the MI355X classes are built as SUBCLASSES of the reference classes: only what the denoiser hot path replaces is overridden
(``load_diffusion_models``, ``forward``, ``_save_lora_weights`` / ``_save_model``), everything else -- ``prepare_conditions``,
``prepare_latents``, ``load_condition_models``, ``load_latent_models``, ``load_pipeline``, ``validation``, ``apply_tensor_parallel`` -- is the
reference's own code, so the unmodified ``SFTTrainer`` (trainer/sft_trainer/trainer.py:380-383, 834-835, 877, 896) finds every method it calls.

Where ``finetrainers`` is not installed (the build container, the GPU box: no diffusers) the same classes stand on
``StandaloneModelSpecification``: a restatement of the GENERIC half of ``ModelSpecification`` (finetrainers/models/modeling_utils.py:26-246:
constructor attributes, the processor loops of ``prepare_conditions`` / ``prepare_latents``, collation) whose model-specific loaders raise
the reference base's own ``NotImplementedError`` -- text encoders, VAE and pipelines are outside the MI355X hot path.
"""

from __future__ import annotations

import importlib
from typing import Any, Dict, List, Optional, Tuple

import torch

# finetrainers/models/modeling_utils.py:22
IGNORE_KEYS_FOR_COLLATION = {"height", "width", "num_frames", "frame_rate", "rope_interpolation_scale", "return_dict", "attention_kwargs",
                             "cross_attention_kwargs", "joint_attention_kwargs", "latents_mean", "latents_std"}

# keyword arguments of the reference constructors (modeling_utils.py:33-52; train.py:48-66 passes them by name)
REFERENCE_CTOR_KEYS = ("pretrained_model_name_or_path", "tokenizer_id", "tokenizer_2_id", "tokenizer_3_id", "text_encoder_id", "text_encoder_2_id",
                       "text_encoder_3_id", "transformer_id", "vae_id", "text_encoder_dtype", "text_encoder_2_dtype", "text_encoder_3_dtype",
                       "transformer_dtype", "vae_dtype", "revision", "cache_dir", "condition_model_processors", "latent_model_processors")


def reference_class(module: str, name: str) -> Optional[type]:
    """The reference class ``module.name`` if the reference package (and what it imports: diffusers, peft, torchdata) is installed, else None."""
    try:
        return getattr(importlib.import_module(module), name)
    except Exception:
        return None


class StandaloneModelSpecification:
    """The generic half of ``ModelSpecification`` (modeling_utils.py:26-246) for machines without the reference package."""

    def __init__(self, pretrained_model_name_or_path: Optional[str] = None, tokenizer_id: Optional[str] = None, tokenizer_2_id: Optional[str] = None,
                 tokenizer_3_id: Optional[str] = None, text_encoder_id: Optional[str] = None, text_encoder_2_id: Optional[str] = None,
                 text_encoder_3_id: Optional[str] = None, transformer_id: Optional[str] = None, vae_id: Optional[str] = None,
                 text_encoder_dtype: torch.dtype = torch.bfloat16, text_encoder_2_dtype: torch.dtype = torch.bfloat16,
                 text_encoder_3_dtype: torch.dtype = torch.bfloat16, transformer_dtype: torch.dtype = torch.bfloat16, vae_dtype: torch.dtype = torch.bfloat16,
                 revision: Optional[str] = None, cache_dir: Optional[str] = None, condition_model_processors: Optional[list] = None,
                 latent_model_processors: Optional[list] = None, **kwargs) -> None:
        self.pretrained_model_name_or_path = pretrained_model_name_or_path
        self.tokenizer_id, self.tokenizer_2_id, self.tokenizer_3_id = tokenizer_id, tokenizer_2_id, tokenizer_3_id
        self.text_encoder_id, self.text_encoder_2_id, self.text_encoder_3_id = text_encoder_id, text_encoder_2_id, text_encoder_3_id
        self.transformer_id, self.vae_id = transformer_id, vae_id
        self.text_encoder_dtype, self.text_encoder_2_dtype, self.text_encoder_3_dtype = text_encoder_dtype, text_encoder_2_dtype, text_encoder_3_dtype
        self.transformer_dtype, self.vae_dtype = transformer_dtype, vae_dtype
        self.revision, self.cache_dir = revision, cache_dir
        self.condition_model_processors = condition_model_processors or []
        self.latent_model_processors = latent_model_processors or []
        self.transformer_config = None  # the reference reads both from the hub (modeling_utils.py:248-300); here load_diffusion_models fills it
        self.vae_config = None

    def _trainer_init(self, *args, **kwargs):
        pass

    def _not_here(self, what: str):
        return NotImplementedError(f"ModelSpecification::{what} is not implemented for {self.__class__.__name__}: it belongs to the reference's "
                                   "model-specific specification (text encoders / VAE / pipeline are outside the MI355X hot path); install "
                                   "finetrainers and this class inherits it")

    def load_condition_models(self) -> Dict[str, torch.nn.Module]:
        raise self._not_here("load_condition_models")

    def load_latent_models(self) -> Dict[str, torch.nn.Module]:
        raise self._not_here("load_latent_models")

    def load_pipeline(self, *args, **kwargs):
        raise self._not_here("load_pipeline")

    def validation(self, *args, **kwargs):
        raise self._not_here("validation")

    def apply_tensor_parallel(self, *args, **kwargs) -> None:
        raise self._not_here("apply_tensor_parallel")

    def _run_processors(self, processors, kwargs: Dict[str, Any]) -> Dict[str, Any]:
        for processor in processors:  # modeling_utils.py:113-147
            kwargs.update(processor(**kwargs))
        return kwargs

    def prepare_conditions(self, processors: Optional[list] = None, **kwargs) -> Dict[str, Any]:
        return self._run_processors(self.condition_model_processors if processors is None else processors, kwargs)

    def prepare_latents(self, processors: Optional[list] = None, **kwargs) -> Dict[str, Any]:
        return self._run_processors(self.latent_model_processors if processors is None else processors, kwargs)

    @staticmethod
    def _collate(data: List[Dict[str, Any]]) -> Dict[str, Any]:
        """modeling_utils.py:156-181."""
        out: Dict[str, Any] = {}
        for key in list(data[0].keys()):
            if key in IGNORE_KEYS_FOR_COLLATION:
                out[key] = data[0][key]
                continue
            vals = [d[key] for d in data]
            if isinstance(vals[0], torch.Tensor):
                vals = torch.cat(vals)
            out[key] = vals
        return out

    def collate_conditions(self, data: List[Dict[str, Any]]) -> Dict[str, Any]:
        return self._collate(data)

    def collate_latents(self, data: List[Dict[str, Any]]) -> Dict[str, Any]:
        return self._collate(data)


def as_drop_in(cls: type, ref_module: str, ref_name: str, base_override: Optional[type] = None) -> type:
    """Rebuild ``cls`` (the MI355X overrides) on top of the reference class ``ref_module.ref_name`` -- or of ``StandaloneModelSpecification`` when
    the reference is not installed.  The base's constructor runs first with the reference's own keyword arguments (so inherited methods find
    every attribute they read, default processors included), then the MI355X constructor.  ``base_override``: tests inject a base class."""
    ref = base_override if base_override is not None else reference_class(ref_module, ref_name)
    base = ref if ref is not None else StandaloneModelSpecification

    def __init__(self, *args, **kwargs):
        if args:  # the reference's first positional parameter
            kwargs = dict(kwargs, pretrained_model_name_or_path=args[0])
            args = args[1:]
        if args:
            raise TypeError(f"{cls.__name__} takes keyword arguments (as train.py passes them)")
        base.__init__(self, **{k: kwargs[k] for k in REFERENCE_CTOR_KEYS if k in kwargs})
        cls.__init__(self, **kwargs)

    return type(cls.__name__, (cls, base), {"__init__": __init__, "__module__": cls.__module__, "__qualname__": cls.__qualname__, "__doc__": cls.__doc__,
                                            "IS_REFERENCE_SUBCLASS": ref is not None, "MI355X_OVERRIDES": cls})


def keep_or_default(self, name: str, value, default):
    """Constructor helper of the override classes: an explicit argument wins, otherwise what the base constructor already set stays."""
    if value is not None:
        return value
    return getattr(self, name, None) if getattr(self, name, None) else default
