"""Generate the golden fixtures under tests/golden/ by EXECUTING the reference's own
functions (read from /root/reference at generation time; nothing is copied into this repo).

The reference package cannot be imported here (``import finetrainers`` needs diffusers /
peft / torchdata, absent from the image), so the individual functions that make up the hot
path are pulled out of their source files with ``ast`` and compiled against a namespace that
stubs only the third-party *names* they mention (``Transformer2DModelOutput``,
``is_torch_version``, ``FlowMatchEulerDiscreteScheduler`` ...).  Sub-modules that live in
diffusers (blocks, rope, time-embed) are supplied by the oracle restatement -- so the
fixtures pin: the patched model-level forward, RoPE apply, RMSNorm patch, latent
normalise / pack / noising / target / timesteps (spec.forward), flow-match functional,
sigma sampling and grad clipping, i.e. everything on the path that is in /root/reference.

Run (in the build container only; /root/reference does not exist on the GPU box):
    python -m oracle.make_golden
"""

from __future__ import annotations

import ast
import math
import os
import random
import sys
import types
from typing import Any, Dict, List, Optional, Tuple, Union

import torch
import torch.distributed.tensor  # noqa: F401  (utils/torch.py:147 names torch.distributed.tensor.DTensor)
from safetensors.torch import save_file

REF = os.environ.get("FTMI_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ltx  # noqa: E402


def _find(node: ast.AST, name: str, cls: Optional[str] = None) -> ast.FunctionDef:
    scope = node
    if cls is not None:
        for n in ast.walk(node):
            if isinstance(n, ast.ClassDef) and n.name == cls:
                scope = n
                break
        else:
            raise KeyError(cls)
    for n in ast.walk(scope):
        if isinstance(n, ast.FunctionDef) and n.name == name:
            return n
    raise KeyError(name)


def extract(relpath: str, name: str, ns: Dict[str, Any], cls: Optional[str] = None):
    """Compile function ``name`` (optionally a method of ``cls`` or a nested def) from a reference file."""
    path = os.path.join(REF, relpath)
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    fn = _find(tree, name, cls)
    fn.decorator_list = []  # staticmethod / torch.no_grad handled by the caller
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    code = compile(mod, path, "exec")
    ns = dict(ns)
    exec(code, ns)
    f = ns[name]
    f.__globals__.update(ns)
    return f


TYPING_NS = dict(Any=Any, Dict=Dict, List=List, Optional=Optional, Tuple=Tuple, Union=Union, torch=torch, math=math)


class _Out:  # stands in for diffusers' Transformer2DModelOutput
    def __init__(self, sample):
        self.sample = sample


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    tensors: Dict[str, torch.Tensor] = {}

    # ---- functional/diffusion.py (pure torch: load the file itself) -------------------
    FF = types.ModuleType("ref_FF")
    with open(os.path.join(REF, "finetrainers/functional/diffusion.py")) as f:
        exec(compile(f.read(), "diffusion.py", "exec"), FF.__dict__)

    # ---- patched transformer forward + rope apply ------------------------------------
    patch_ns = dict(TYPING_NS, Transformer2DModelOutput=_Out, is_torch_version=lambda *a: True)
    ref_forward = extract("finetrainers/patches/models/ltx_video/patch.py", "_patched_LTXVideoTransformer3D_forward", patch_ns)
    ref_rope_apply = extract("finetrainers/patches/models/ltx_video/patch.py", "apply_rotary_emb", patch_ns)

    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 24, 64, generator=g).bfloat16()
    cos = torch.randn(2, 24, 64, generator=g)
    sin = torch.randn(2, 24, 64, generator=g)
    tensors["rope.x"], tensors["rope.cos"], tensors["rope.sin"] = x, cos, sin
    tensors["rope.out"] = ref_rope_apply(x, (cos, sin))

    # ---- RMSNorm patch -----------------------------------------------------------------
    rms_ns = dict(TYPING_NS, nn=torch.nn, is_torch_npu_available=lambda: False, is_torch_version=lambda *a: True)
    ref_rms = extract("finetrainers/patches/dependencies/diffusers/rms_norm.py", "_patched_rms_norm_forward", rms_ns)
    w = (1 + 0.1 * torch.randn(64, generator=g)).bfloat16()
    tensors["rms.x"], tensors["rms.w"] = x, w
    tensors["rms.out_affine"] = ref_rms(types.SimpleNamespace(weight=w, eps=1e-5, bias=None), x)
    tensors["rms.out_plain"] = ref_rms(types.SimpleNamespace(weight=None, eps=1e-6, bias=None), x)

    # ---- learning-rate multipliers (finetrainers/optimizer.py: the seven functions get_lr_scheduler chooses from) ------------------
    lr_ns = dict(TYPING_NS, Callable=__import__("typing").Callable)
    steps = list(range(0, 60))
    sched = {
        "constant": extract("finetrainers/optimizer.py", "get_constant_schedule", lr_ns)(),
        "constant_with_warmup": extract("finetrainers/optimizer.py", "get_constant_schedule_with_warmup", lr_ns)(7),
        "piecewise_constant": extract("finetrainers/optimizer.py", "get_piecewise_constant_schedule", lr_ns)("1:10,0.1:20,0.01:30,0.005"),
        "linear": extract("finetrainers/optimizer.py", "get_linear_schedule_with_warmup", lr_ns)(5, 50),
        "cosine": extract("finetrainers/optimizer.py", "get_cosine_schedule_with_warmup", lr_ns)(5, 50, 1),
        "cosine_half": extract("finetrainers/optimizer.py", "get_cosine_schedule_with_warmup", lr_ns)(5, 50, 0.5),
        "cosine_with_restarts": extract("finetrainers/optimizer.py", "get_cosine_with_hard_restarts_schedule_with_warmup", lr_ns)(5, 50, 3),
        "polynomial": extract("finetrainers/optimizer.py", "get_polynomial_decay_schedule_with_warmup", lr_ns)(5, 50, 5e-5, 1e-7, 2.0),
    }
    for k_, f_ in sched.items():
        tensors[f"lr.{k_}"] = torch.tensor([f_(t_) for t_ in steps], dtype=torch.float64)

    # ---- spec.forward driving the reference's patched transformer forward --------------
    spec_ns = dict(TYPING_NS, random=types.SimpleNamespace(random=lambda: 1.0), FF=FF,
                   DiagonalGaussianDistribution=None, LTXVideoTransformer3DModel=object)
    spec_fwd = extract("finetrainers/models/ltx_video/base_specification.py", "forward", spec_ns, cls="LTXVideoModelSpecification")
    norm_lat = extract("finetrainers/models/ltx_video/base_specification.py", "_normalize_latents", spec_ns, cls="LTXVideoModelSpecification")
    pack_lat = extract("finetrainers/models/ltx_video/base_specification.py", "_pack_latents", spec_ns, cls="LTXVideoModelSpecification")

    def run_spec(cfg: ltx.LTXConfig, tag: str, frames: int, height: int, width: int, first_frame: bool, seed: int,
                 rank: int, layers_note: str) -> None:
        model = ltx.build_model(cfg, seed=0, rank=rank, lora_b_std=0.02 if rank else None)
        model.gradient_checkpointing = False
        inp = ltx.synth_inputs(cfg, 1, frames, height, width, seed=seed, mask_lens=[cfg.text_seq_len // 4], sigmas=[0.7])
        inp.latents_mean = torch.randn(cfg.in_channels, generator=torch.Generator().manual_seed(5)) * 0.1
        inp.latents_std = 1.0 + 0.2 * torch.rand(cfg.in_channels, generator=torch.Generator().manual_seed(6))

        class _T(torch.nn.Module):  # the "transformer" the spec calls: reference forward over oracle sub-modules
            def __init__(self, m):
                super().__init__()
                self.m = m

            def forward(self, **kw):
                return ref_forward(self.m, **kw)

        spec_fwd.__globals__["random"] = types.SimpleNamespace(random=(lambda: 0.0) if first_frame else (lambda: 1.0))
        self_ns = types.SimpleNamespace(
            transformer_config=types.SimpleNamespace(patch_size=1, patch_size_t=1),
            _normalize_latents=norm_lat,
            _pack_latents=pack_lat,
        )
        sig5 = inp.sigmas.view(-1, 1, 1, 1, 1)
        gen = torch.Generator().manual_seed(seed + 100)
        torch.manual_seed(seed + 200)  # drives torch.rand_like(sigmas) of the first-frame branch
        with torch.no_grad():
            pred, target, sig = spec_fwd(
                self_ns,
                transformer=_T(model),
                condition_model_conditions={
                    "encoder_hidden_states": inp.encoder_hidden_states,
                    "encoder_attention_mask": inp.encoder_attention_mask,
                },
                latent_model_conditions={
                    "latents": inp.latents.clone(),
                    "num_frames": frames,
                    "height": height,
                    "width": width,
                    "latents_mean": inp.latents_mean,
                    "latents_std": inp.latents_std,
                },
                sigmas=sig5,
                generator=gen,
                compute_posterior=True,
            )
        tensors[f"{tag}.pred"] = pred.contiguous()
        tensors[f"{tag}.target"] = target.contiguous()
        tensors[f"{tag}.sigmas"] = sig.contiguous()
        tensors[f"{tag}.meta"] = torch.tensor([frames, height, width, int(first_frame), seed, rank], dtype=torch.int64)

    run_spec(ltx.LTXConfig.dummy(), "spec_dummy", 2, 4, 4, False, 3, 0, "dummy")
    run_spec(ltx.LTXConfig.dummy(), "spec_dummy_ff", 3, 4, 4, True, 4, 0, "dummy first-frame")
    run_spec(ltx.LTXConfig.production(num_layers=1), "spec_prod1", 2, 4, 4, False, 7, 64, "production dims, 1 layer")
    run_spec(ltx.LTXConfig.production(num_layers=1), "spec_prod1_ff", 2, 4, 4, True, 8, 64, "production dims, 1 layer, first-frame")

    # ---- pack / normalise on their own (B=1, where the reference's view() is valid) -----
    lat = torch.randn(1, 8, 3, 4, 6, generator=g).bfloat16()
    tensors["pack.in"] = lat
    tensors["pack.out"] = pack_lat(lat, 1, 1).contiguous()
    tensors["pack.out_p2"] = pack_lat(lat[:, :, :2], 2, 1).contiguous()
    mean, std = torch.randn(8, generator=g), 1 + torch.rand(8, generator=g)
    tensors["norm.mean"], tensors["norm.std"] = mean, std
    tensors["norm.out"] = norm_lat(lat, mean, std)

    # ---- flow match functional ----------------------------------------------------------
    x0 = torch.randn(2, 8, 3, 4, 4, generator=g).bfloat16()
    n = torch.randn(2, 8, 3, 4, 4, generator=g).bfloat16()
    t = torch.tensor([0.3, 0.9]).view(-1, 1, 1, 1, 1)
    tensors["fm.x0"], tensors["fm.n"], tensors["fm.t"] = x0, n, t
    tensors["fm.xt"] = FF.flow_match_xt(x0, n, t)
    tensors["fm.target"] = FF.flow_match_target(n, x0)

    # ---- sigma sampling -------------------------------------------------------------------
    class _FM:  # stands in for diffusers.FlowMatchEulerDiscreteScheduler (isinstance dispatch only)
        pass

    class _DDIM:
        pass

    diff_ns = dict(TYPING_NS, FlowMatchEulerDiscreteScheduler=_FM, CogVideoXDDIMScheduler=_DDIM,
                   compute_loss_weighting_for_sd3=None)
    ref_density = extract("finetrainers/utils/diffusion.py", "compute_density_for_timestep_sampling", diff_ns)
    diff_ns["compute_density_for_timestep_sampling"] = ref_density
    ref_prepare_sigmas = extract("finetrainers/utils/diffusion.py", "prepare_sigmas", diff_ns)
    table = ltx.scheduler_sigmas()
    for scheme in ("none", "logit_normal", "mode"):
        gen = torch.Generator().manual_seed(21)
        tensors[f"sigmas.{scheme}"] = ref_prepare_sigmas(
            scheduler=_FM(), sigmas=table, batch_size=16, num_train_timesteps=1000, flow_weighting_scheme=scheme,
            flow_logit_mean=0.0, flow_logit_std=1.0, flow_mode_scale=1.29, device=torch.device("cpu"), generator=gen,
        )

    # ---- CogVideoX: the reference's spec.forward / RoPE table / frame padding / DDIM loss weight and sigma sampling ----
    from oracle import cogvideox as cvx

    cvx_ns = dict(TYPING_NS, get_3d_rotary_pos_embed=cvx.get_3d_rotary_pos_embed, get_resize_crop_region_for_grid=cvx.get_resize_crop_region_for_grid,
                  DiagonalGaussianDistribution=None, CogVideoXTransformer3DModel=object, CogVideoXDDIMScheduler=cvx.CogVideoXDDIMScheduler)
    ref_cvx_rope = extract("finetrainers/models/cogvideox/utils.py", "prepare_rotary_positional_embeddings", cvx_ns)
    cvx_ns["prepare_rotary_positional_embeddings"] = ref_cvx_rope
    ref_cvx_fwd = extract("finetrainers/models/cogvideox/base_specification.py", "forward", cvx_ns, cls="CogVideoXModelSpecification")
    ref_cvx_pad = extract("finetrainers/models/cogvideox/base_specification.py", "_pad_frames", cvx_ns, cls="CogVideoXModelSpecification")
    for name, (h, w, f, pt) in {"v10": (48, 48, 3, None), "v15": (48, 64, 4, 2)}.items():
        c_, s_ = ref_cvx_rope(height=h, width=w, num_frames=f, vae_scale_factor_spatial=8, patch_size=2, patch_size_t=pt, attention_head_dim=16,
                              device=torch.device("cpu"), base_height=192, base_width=192)
        tensors[f"cvx.rope_{name}.cos"], tensors[f"cvx.rope_{name}.sin"] = c_, s_
    lat5 = torch.randn(1, 3, 4, 6, 6, generator=g).bfloat16()
    tensors["cvx.pad.in"] = lat5
    tensors["cvx.pad.out2"] = ref_cvx_pad(lat5, 2).contiguous()
    tensors["cvx.pad.out3"] = ref_cvx_pad(lat5, 3).contiguous()

    def run_cvx(tag: str, cfg, frames: int, hh: int, ww: int, seed: int, rank: int) -> None:
        model = cvx.build_model(cfg, seed=0, rank=rank, alpha=float(max(rank, 1)), lora_b_std=0.02 if rank else None)
        sched = cvx.CogVideoXDDIMScheduler()
        gg = torch.Generator().manual_seed(seed)
        lat = torch.randn(2, frames, cfg.in_channels, hh, ww, generator=gg).bfloat16()
        text = torch.randn(2, cfg.max_text_seq_length, cfg.text_embed_dim, generator=gg).bfloat16()
        sig = torch.tensor([0.25, 0.7])
        self_ns = types.SimpleNamespace(transformer_config=types.SimpleNamespace(**{k: getattr(cfg, k) for k in (
            "sample_height", "sample_width", "patch_size", "patch_size_t", "ofs_embed_dim", "attention_head_dim", "use_rotary_positional_embeddings")}),
            vae_config=types.SimpleNamespace(scaling_factor=1.15258426, invert_scale_latents=False), _pad_frames=ref_cvx_pad)
        with torch.no_grad():
            pred, target, sg = ref_cvx_fwd(self_ns, transformer=model, scheduler=sched, condition_model_conditions={"encoder_hidden_states": text},
                                           latent_model_conditions={"latents": lat.clone()}, sigmas=sig, generator=torch.Generator().manual_seed(seed + 100),
                                           compute_posterior=True)
        tensors[f"{tag}.pred"], tensors[f"{tag}.target"], tensors[f"{tag}.sigmas"] = pred.contiguous(), target.contiguous(), sg.contiguous()
        tensors[f"{tag}.meta"] = torch.tensor([frames, hh, ww, seed, rank], dtype=torch.int64)

    run_cvx("cvx.spec_dummy", cvx.CogVideoXConfig.dummy(), 3, 6, 6, 5, 0)  # the reference's own fixture config (rotary, 2 layers)
    cfg_2b1 = cvx.CogVideoXConfig(num_layers=1, sample_height=8, sample_width=12, sample_frames=9)  # 2b width (30 x 64), sincos positions, 1 layer
    run_cvx("cvx.spec_2b1", cfg_2b1, 3, 8, 12, 6, 64)
    ref_plw = extract("finetrainers/utils/diffusion.py", "prepare_loss_weights", dict(diff_ns, CogVideoXDDIMScheduler=cvx.CogVideoXDDIMScheduler))
    sch = cvx.CogVideoXDDIMScheduler()
    ts = torch.tensor([0, 250, 700, 999])
    tensors["cvx.loss_weights"] = ref_plw(scheduler=sch, alphas=sch.alphas_cumprod[ts], sigmas=None, flow_weighting_scheme="none")
    ddim_ns = dict(diff_ns, CogVideoXDDIMScheduler=cvx.CogVideoXDDIMScheduler)
    ddim_ns["compute_density_for_timestep_sampling"] = ref_density
    ref_ps_ddim = extract("finetrainers/utils/diffusion.py", "prepare_sigmas", ddim_ns)
    tensors["cvx.sigmas"] = ref_ps_ddim(scheduler=sch, sigmas=table, batch_size=16, num_train_timesteps=1000, device=torch.device("cpu"),
                                        generator=torch.Generator().manual_seed(33))

    # ---- grad clipping ----------------------------------------------------------------------
    from torch.utils._foreach_utils import (
        _device_has_foreach_support,
        _group_tensors_by_device_and_dtype,
        _has_foreach_support,
    )

    clip_ns = dict(TYPING_NS, dist=torch.distributed,
                   _device_has_foreach_support=_device_has_foreach_support,
                   _group_tensors_by_device_and_dtype=_group_tensors_by_device_and_dtype,
                   _has_foreach_support=_has_foreach_support)
    clip_ns["_get_total_norm"] = extract("finetrainers/utils/torch.py", "_get_total_norm", clip_ns)
    clip_ns["_clip_grads_with_norm_"] = extract("finetrainers/utils/torch.py", "_clip_grads_with_norm_", clip_ns)
    ref_clip = extract("finetrainers/utils/torch.py", "clip_grad_norm_", clip_ns)
    for tag, scale in (("clip_big", 3.0), ("clip_small", 1e-3)):
        ps = []
        for i, shp in enumerate([(64, 32), (32, 64), (7,), (128, 16)]):
            p = torch.nn.Parameter(torch.zeros(shp))
            p.grad = torch.randn(shp, generator=g) * scale
            tensors[f"{tag}.g{i}"] = p.grad.clone()
            ps.append(p)
        ps.append(torch.nn.Parameter(torch.zeros(3)))  # grad None => skipped (utils/torch.py:135)
        with torch.no_grad():
            total = ref_clip(ps, 1.0, foreach=True)
        tensors[f"{tag}.total_norm"] = total.reshape(1)
        for i, p in enumerate(ps[:-1]):
            tensors[f"{tag}.out{i}"] = p.grad.clone()

    # ---- Wan: the reference's spec forward, its DiagonalGaussianDistribution and its patched time/text embedding, executed -----------
    from oracle import wan

    MU = types.ModuleType("ref_models_utils")  # finetrainers/models/utils.py with diffusers' randn_tensor replaced by its torch equivalent
    MU.__dict__["__name__"] = "ref_models_utils"
    src = open(os.path.join(REF, "finetrainers/models/utils.py")).read().replace("from diffusers.utils.torch_utils import randn_tensor", "")
    MU.__dict__["randn_tensor"] = lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype)
    exec(compile(src, "models/utils.py", "exec"), MU.__dict__)
    wan_ns = dict(TYPING_NS, FF=FF, DiagonalGaussianDistribution=MU.DiagonalGaussianDistribution, WanTransformer3DModel=object)
    wan_fwd = extract("finetrainers/models/wan/base_specification.py", "forward", wan_ns, cls="WanModelSpecification")
    wan_norm = extract("finetrainers/models/wan/base_specification.py", "_normalize_latents", wan_ns, cls="WanModelSpecification")
    wcfg = wan.WanConfig.dummy()
    wmodel = wan.build_model(wcfg, seed=0, dtype=torch.bfloat16)
    gw = torch.Generator().manual_seed(21)
    B_, C_, F_, H_, W_ = 2, 16, 3, 8, 12
    mom = torch.randn(B_, 2 * C_, F_, H_, W_, generator=gw)
    mom[:, C_:] = mom[:, C_:] * 0.3 - 2.0
    mom = mom.bfloat16()
    lmean = (0.2 * torch.randn(C_, generator=gw)).float()
    lstd = (1.0 / (0.5 + torch.rand(C_, generator=gw))).float()  # the processors store the reciprocal
    wtext = torch.randn(B_, 7, wcfg.text_dim, generator=gw).bfloat16()
    wsig = torch.tensor([0.31, 0.84]).view(B_, 1, 1, 1, 1)
    self_stub = types.SimpleNamespace(_normalize_latents=wan_norm, transformer_config={})
    with torch.no_grad():
        pred_w, target_w, _ = wan_fwd(self_stub, wmodel, {"encoder_hidden_states": wtext},
                                      {"latents": mom.clone(), "latents_mean": lmean, "latents_std": lstd}, wsig, generator=torch.Generator().manual_seed(77))
    for k_, v_ in (("moments", mom), ("latents_mean", lmean), ("latents_std", lstd), ("text", wtext), ("sigmas", wsig.flatten()), ("pred", pred_w), ("target", target_w)):
        tensors[f"wan.spec.{k_}"] = v_
    ref_embed = extract("finetrainers/patches/models/wan/patch.py", "_patched_WanTimeTextImageEmbedding_forward", dict(TYPING_NS))
    with torch.no_grad():
        temb_w, tproj_w, text_w, _ = ref_embed(wmodel.condition_embedder, torch.tensor([310, 840]), wtext)
    tensors["wan.embed.temb"], tensors["wan.embed.timestep_proj"], tensors["wan.embed.text"] = temb_w, tproj_w, text_w

    # ---- HunyuanVideo: the reference's spec forward executed around a recording stub transformer -----------------------------
    # (the DiT itself is [upstream]; what is pinned here is everything the reference does around the call: posterior draw, VAE scaling factor,
    #  flow-match mix, integer timesteps, guidance * 1000, the keyword set handed to the transformer, the target)
    hy_ns = dict(TYPING_NS, FF=FF, DiagonalGaussianDistribution=MU.DiagonalGaussianDistribution, HunyuanVideoTransformer3DModel=object)
    hy_fwd = extract("finetrainers/models/hunyuan_video/base_specification.py", "forward", hy_ns, cls="HunyuanVideoModelSpecification")
    gh = torch.Generator().manual_seed(31)
    Bh, Ch, Fh, Hh, Wh = 2, 4, 3, 6, 8
    hy_mom = torch.randn(Bh, 2 * Ch, Fh, Hh, Wh, generator=gh)
    hy_mom[:, Ch:] = hy_mom[:, Ch:] * 0.3 - 2.0
    hy_mom = hy_mom.bfloat16()
    hy_cond = {"encoder_hidden_states": torch.randn(Bh, 5, 16, generator=gh).bfloat16(), "encoder_attention_mask": torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]]),
               "pooled_projections": torch.randn(Bh, 8, generator=gh).bfloat16()}
    hy_sig = torch.tensor([0.27, 0.66]).view(Bh, 1, 1, 1, 1)
    seen = {}

    def hy_stub(**kw):  # records what the reference hands over and returns a deterministic function of it
        seen.update({k_: (v_.clone() if torch.is_tensor(v_) else v_) for k_, v_ in kw.items()})
        return ((kw["hidden_states"].float() * 0.5 + kw["guidance"].view(-1, 1, 1, 1, 1).float() * 1e-4 + kw["timestep"].view(-1, 1, 1, 1, 1).float() * 1e-3).to(kw["hidden_states"].dtype),)

    hy_self = types.SimpleNamespace(vae_config=types.SimpleNamespace(scaling_factor=0.476986))
    for tag, cp in (("hunyuan.spec_moments", False), ("hunyuan.spec_latents", True)):
        lat_in = hy_mom.clone() if not cp else hy_mom[:, :Ch].clone()
        with torch.no_grad():
            pred_h, target_h, _ = hy_fwd(hy_self, hy_stub, dict(hy_cond), {"latents": lat_in}, hy_sig, guidance=6.0, generator=torch.Generator().manual_seed(91),
                                         compute_posterior=cp)
        assert set(seen) == {"hidden_states", "guidance", "encoder_hidden_states", "encoder_attention_mask", "pooled_projections", "timestep", "return_dict"}
        for k_, v_ in (("latents_in", lat_in), ("pred", pred_h), ("target", target_h), ("noisy", seen["hidden_states"]), ("guidance", seen["guidance"]),
                       ("timestep", seen["timestep"])):
            tensors[f"{tag}.{k_}"] = v_
    tensors["hunyuan.spec.sigmas"] = hy_sig.flatten()

    tensors = {k: v.detach().clone().contiguous() for k, v in tensors.items()}
    path = os.path.join(OUT, "reference_fixtures.safetensors")
    save_file(tensors, path, metadata={"generator": "oracle/make_golden.py", "reference": "a-r-r-o-w/finetrainers @ 2025-08-29"})
    print(f"wrote {path}: {len(tensors)} tensors, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
