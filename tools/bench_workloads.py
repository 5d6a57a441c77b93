"""The four workloads `bench.py --workload {ltx,cogvideox,wan,hunyuan}` can time (BASELINE.json configs[1], [2], [3], [4]): for each one a
builder that puts a random-init model of the named architecture, its step object and one synthetic batch of the named clip shape on the GPU
and returns the step closure + the static part of the JSON line, and a `cpu_baseline` that times the oracle (CPU restatement of the reference
path, kind "port") on a bounded sample of the same workload.  Bench infrastructure: the only place outside tests/ and __graft_entry__.smoke()
that imports oracle/, and only inside the cpu_baseline functions."""

from __future__ import annotations

import os
import time
from typing import Any, Callable, Dict

import torch

bf16 = torch.bfloat16
PEAK_BF16_TFLOPS = 2500.0


def _time_block(fn: Callable[[], None], warm: int = 1, timed: int = 1):
    ts = []
    for it in range(warm + timed):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sum(ts[warm:]) / timed, ts


# ------------------------------------------------------------------------------------------------------------------------------------------
# CogVideoX-2b LoRA (configs[2]): 49 x 480 x 720 -> latents [1, 13, 16, 60, 90], 226 text + 17 550 video tokens, 30 blocks, batch 1 per GPU
# ------------------------------------------------------------------------------------------------------------------------------------------
def build_cogvideox(args, par, dev) -> Dict[str, Any]:
    from finetrainers_amd.cogvideox import CogVideoXTransformerConfig, MI355XCogVideoXSFTStep, MI355XCogVideoXTransformer3DModel

    layers = args.layers if args.layers > 0 else 30
    cfg = CogVideoXTransformerConfig(num_layers=layers)
    model = MI355XCogVideoXTransformer3DModel(cfg, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    D = cfg.inner_dim

    def rnd(shape, fan_in):
        return (torch.randn(shape, generator=g, device=dev) / fan_in ** 0.5).to(bf16)

    def one(k, shp):
        if len(shp) == 2:
            return rnd(shp, shp[1])
        return torch.ones(shp, device=dev, dtype=bf16) if "norm" in k and k.endswith("weight") else 0.02 * rnd(shp, 1)

    sd = {k: one(k, getattr(model, name).shape) for k, name in model._KEYS.items()}
    sd["patch_embed.proj.weight"] = rnd((D, cfg.in_channels, 2, 2), 64)
    for i, blk in enumerate(model.transformer_blocks):
        for k, name in blk._KEYS.items():
            sd[f"transformer_blocks.{i}.{k}"] = one(k, getattr(blk, name).shape)
    model.load_diffusers_state_dict(sd)
    del sd
    model.add_adapter(r=args.rank, lora_alpha=float(args.rank))
    with torch.no_grad():
        n = model.lora_flat.numel() // 2
        model.lora_flat[n:].normal_(0, 0.01, generator=g)  # B != 0 so every gradient path carries data
    step = MI355XCogVideoXSFTStep(model, lr=5e-5, betas=(0.9, 0.99), generator=torch.Generator(device=dev).manual_seed(1 + par.rank),
                                  parallel=par if par.world_size > 1 else None)
    g.manual_seed(100 + par.rank)  # every rank its own clip
    lat = torch.randn((1, 13, 16, 60, 90), generator=g, device=dev).to(bf16)
    text = torch.randn((1, 226, 4096), generator=g, device=dev).to(bf16)
    N = 226 + 17550
    flop = layers * (2.0 * N * D * D * 12 * 2 + 4.0 * N * N * D * 3.5)  # linears forward + dgrad, attention forward + 2.5 x backward (LoRA / embed / head terms omitted)
    return {
        "one_step": lambda: step.step(lat, text),
        "samples_per_step": 1,
        "step_tflop": flop / 1e12,
        "metric": "train samples/sec (+ step ms) CogVideoX-2b LoRA 49x480x720 (BASELINE configs[2])",
        "data": "synthetic latents [1,13,16,60,90] + random text embeds [1,226,4096], random-init weights of the CogVideoX-2b DiT",
        "config": {"workload": f"CogVideoX-2b LoRA rank={args.rank} bf16 SFT step, 49x480x720 clip (226 text + 17550 video tokens), batch 1 per GPU, {layers} blocks (BASELINE configs[2])"
                               + ("" if layers == 30 else " -- REDUCED depth"),
                   "model": "CogVideoX-2b DiT: 30 blocks, width 1920, 30 x 64 heads, joint text + video attention", "seq_len": N,
                   "activation_checkpointing": False, "orchestration": "C block stack (ftmi_cog_blocks_forward / _backward)"},
        "layers": layers,
    }


def cpu_baseline_cogvideox(args, layers: int) -> Dict[str, Any]:
    from oracle import cogvideox as cvx

    ocfg = cvx.CogVideoXConfig(num_layers=1)
    oblk = cvx.build_model(ocfg, seed=0, rank=args.rank, alpha=float(args.rank), lora_b_std=0.02).transformer_blocks[0]
    D = ocfg.inner_dim
    g = torch.Generator().manual_seed(0)
    vid, txt, temb = torch.randn(1, 17550, D, generator=g).to(bf16), torch.randn(1, 226, D, generator=g).to(bf16), torch.randn(1, 512, generator=g).to(bf16)

    def fb():
        for p_ in oblk.parameters():
            p_.grad = None
        vr, tr_ = vid.clone().requires_grad_(True), txt.clone().requires_grad_(True)
        hv, ht = oblk(vr, tr_, temb)
        torch.autograd.backward([hv, ht], [torch.ones_like(hv), torch.ones_like(ht)])

    per_block, ts = _time_block(fb)
    return {"value": 1.0 / (per_block * layers), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle CogVideoXBlock forward + backward at the full 17776 tokens, 1 warm-up ({ts[0]:.1f} s) + 1 timed = {per_block:.1f} s, scaled x{layers} blocks "
                      f"(embed / head / optimiser are < 2 % of the step)"}


# ------------------------------------------------------------------------------------------------------------------------------------------
# Wan2.1-T2V-1.3B full fine-tune (configs[3]): 81 x 512 x 512 -> latents [1, 16, 21, 64, 64], 21 504 video + 512 text tokens, 30 blocks;
# N GPUs: parameters sharded per unit (the reference's FSDP-2 data flow), every rank its own clip
# ------------------------------------------------------------------------------------------------------------------------------------------
def build_wan(args, par, dev) -> Dict[str, Any]:
    from finetrainers_amd.wan import MI355XWanFullFinetuneStep, MI355XWanTransformer3DModel, WanTransformerConfig

    layers = args.layers if args.layers > 0 else 30
    cfg = WanTransformerConfig(num_layers=layers)
    model = MI355XWanTransformer3DModel(cfg, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    D = cfg.inner_dim
    with torch.no_grad():
        for name, v in model.state_dict_views().items():
            if name.endswith("weight") and v.dim() >= 2:
                v.copy_((torch.randn(v.shape, generator=g, device=dev) / v.shape[-1] ** 0.5).to(bf16))
            elif "norm" in name and name.endswith("weight"):
                v.fill_(1.0)
            elif "scale_shift_table" in name:
                v.copy_((torch.randn(v.shape, generator=g, device=dev) / D ** 0.5).to(bf16))
            else:
                v.copy_((0.02 * torch.randn(v.shape, generator=g, device=dev)).to(bf16))
    step = MI355XWanFullFinetuneStep(model, lr=1e-5, betas=(0.9, 0.95), weight_decay=1e-4, generator=torch.Generator(device=dev).manual_seed(1 + par.rank),
                                     parallel=par if par.world_size > 1 else None)
    g.manual_seed(100 + par.rank)
    B, C, F_, H, W, T = 1, 16, 21, 64, 64, 512
    moments = torch.randn((B, 2 * C, F_, H, W), generator=g, device=dev).to(bf16)
    moments[:, C:] = (moments[:, C:].float() * 0.3 - 2.0).to(bf16)
    text = torch.randn((B, T, cfg.text_dim), generator=g, device=dev).to(bf16)
    mean, std = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    sig = torch.tensor([0.6], device=dev)
    S, Fd = 21 * 32 * 32, cfg.ffn_dim
    lin = 2.0 * S * (6 * D * D + 2 * D * Fd) + 2.0 * T * 2 * D * D  # per block, forward: q|k|v, out, cross q, cross out, feed-forward; text k|v
    att = 4.0 * S * S * D + 4.0 * S * T * D                          # self + cross attention, forward
    flop = layers * (3.0 * lin + 3.5 * att)                          # linears: forward + input gradient + weight gradient; attention backward = 2.5 x forward
    W_ = par.world_size
    return {
        "one_step": lambda: step.step(moments, text, mean, std, sig),
        "samples_per_step": 1,
        "step_tflop": flop / 1e12,
        "metric": "train samples/sec (+ step ms) Wan-T2V-1.3B full fine-tune 81x512x512 (BASELINE configs[3])",
        "data": "synthetic posterior moments [1,32,21,64,64] + random text embeds [1,512,4096], random-init weights of the Wan2.1-T2V-1.3B DiT",
        "config": {"workload": f"Wan-T2V-1.3B full fine-tune bf16 step, 81x512x512 clip ({S} video + {T} text tokens), batch 1 per GPU, {layers} blocks (BASELINE configs[3])"
                               + ("" if layers == 30 else " -- REDUCED depth"),
                   "model": "Wan2.1-T2V-1.3B DiT: 30 blocks, width 1536, 12 x 128 heads, 1.42 B trainable bf16 parameters", "seq_len": S,
                   "parallelism_note": f"fsdp{W_}: parameters sharded per unit (bf16 all-gather / fp32 reduce-scatter)" if W_ > 1 else "one GPU: whole shards, no collective",
                   "activation_checkpointing": False, "orchestration": ("one C call per block and direction (ftmi_wan_block_forward / _backward)" if os.environ.get("FTMI_NATIVE_BLOCKS", "1") != "0" else "python, per kernel over the C ABI")},
        "layers": layers,
    }


def cpu_baseline_wan(args, layers: int) -> Dict[str, Any]:
    from oracle import wan

    ocfg = wan.WanConfig(num_layers=1)
    oblk = wan.WanTransformerBlock(ocfg).to(bf16)
    D, S, T = ocfg.num_attention_heads * ocfg.attention_head_dim, 21504, 512
    g = torch.Generator().manual_seed(0)
    xv, ev, tv = torch.randn(1, S, D, generator=g).to(bf16), torch.randn(1, T, D, generator=g).to(bf16), torch.randn(1, 6, D, generator=g).to(bf16)
    ang = torch.rand(S, 64, generator=g, dtype=torch.float64) * 6.283
    freqs = torch.polar(torch.ones_like(ang), ang).view(1, 1, S, 64)

    def fb():
        for p_ in oblk.parameters():
            p_.grad = None
        xr = xv.clone().requires_grad_(True)
        oblk(xr, ev, tv, freqs).backward(torch.ones_like(xv))

    per_block, ts = _time_block(fb)
    return {"value": 1.0 / (per_block * layers), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle WanTransformerBlock forward + backward (every parameter gradient) at the full {S} + {T} tokens, 1 warm-up ({ts[0]:.1f} s) + 1 timed = "
                      f"{per_block:.1f} s, scaled x{layers} blocks"}


# ------------------------------------------------------------------------------------------------------------------------------------------
# HunyuanVideo LoRA, fp8 weight storage (configs[4]): 61 x 544 x 960 -> latents [1, 16, 16, 68, 120], 32 640 video + 256 text tokens,
# 20 dual-stream + 40 single-stream blocks (12.8 B parameters), batch 1 per GPU
# ------------------------------------------------------------------------------------------------------------------------------------------
def build_hunyuan(args, par, dev) -> Dict[str, Any]:
    from finetrainers_amd import ops
    from finetrainers_amd.hunyuan_video import HunyuanVideoTransformerConfig, MI355XHunyuanVideoSFTStep, MI355XHunyuanVideoTransformer3DModel

    nd, ns = (20, 40) if args.layers <= 0 else (max(1, args.layers // 3), max(1, args.layers - args.layers // 3))
    cfg = HunyuanVideoTransformerConfig(num_layers=nd, num_single_layers=ns)
    model = MI355XHunyuanVideoTransformer3DModel(cfg, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    with torch.no_grad():
        def init(t, unit_scale):
            if t.dim() == 2:
                t.copy_((torch.randn(t.shape, generator=g, device=dev) / t.shape[1] ** 0.5).to(bf16))
            elif unit_scale:
                t.fill_(1.0)
            else:
                t.copy_((0.02 * torch.randn(t.shape, generator=g, device=dev)).to(bf16))

        for name, t in model.p.items():
            init(t, name.endswith("weight") and t.dim() == 1)  # the refiner's LayerNorm weights
        model.proj_out_w_t = ops.transpose_bf16(model.p["proj_out.weight"])
        for blk in list(model.transformer_blocks) + list(model.single_transformer_blocks):
            for name, buf in blk.named_buffers():
                if buf is not None and not name.endswith("_t") and name not in ("ones", "zeros"):
                    init(buf, name.startswith("norm_") and buf.dim() == 1)  # q / k RMSNorm weights
            for name in getattr(blk, "_TRANSPOSED", ("wq", "wk", "wv", "proj_mlp_w", "proj_out_w")):
                setattr(blk, name + "_t", ops.transpose_bf16(getattr(blk, name)))
    model.apply_layerwise_casting()
    model.add_adapter(r=args.rank, lora_alpha=float(args.rank))
    ckpt = bool(getattr(args, "gradient_checkpointing", False))
    if ckpt:
        model.apply_activation_checkpointing("full")
    with torch.no_grad():
        for p in model.lora_parameters()[1::2]:
            p.normal_(0, 0.01, generator=g)  # B != 0 so every gradient path carries data
    step = MI355XHunyuanVideoSFTStep(model, lr=2e-5, guidance=1.0, generator=torch.Generator(device=dev).manual_seed(1 + par.rank),
                                     parallel=par if par.world_size > 1 else None)
    g.manual_seed(100 + par.rank)
    B, C, F_, H, W, T = 1, 16, 16, 68, 120, 256
    lat = torch.randn((B, C, F_, H, W), generator=g, device=dev).to(bf16)
    mask = torch.ones(B, T, dtype=torch.long, device=dev)
    mask[:, 200:] = 0
    cond = {"encoder_hidden_states": torch.randn((B, T, cfg.text_embed_dim), generator=g, device=dev).to(bf16), "encoder_attention_mask": mask,
            "pooled_projections": torch.randn((B, cfg.pooled_projection_dim), generator=g, device=dev).to(bf16)}
    sig = torch.tensor([0.6], device=dev)
    S, D = 16 * 34 * 60, cfg.inner_dim
    N = S + T
    mlp = int(D * cfg.mlp_ratio)
    dual = 2.0 * S * (4 * D * D + 2 * D * mlp) * 2 + 2.0 * T * (4 * D * D + 2 * D * mlp) * 2 + 4.0 * N * N * D * 3.5
    single = 2.0 * N * (3 * D * D + D * mlp + (D + mlp) * D) * 2 + 4.0 * N * N * D * 3.5
    flop = nd * dual + ns * single  # linears forward + input gradient, attention forward + 2.5 x backward (LoRA / front / head terms omitted)
    full = (nd, ns) == (20, 40)
    return {
        "one_step": lambda: step.step(lat, cond, sig),
        "samples_per_step": 1,
        "step_tflop": flop / 1e12,
        "metric": "train samples/sec (+ step ms) HunyuanVideo LoRA fp8-weight-storage 61x544x960 (BASELINE configs[4])",
        "data": "synthetic latents [1,16,16,68,120] + random text embeds [1,256,4096] (200 real tokens) + pooled [1,768], random-init weights of the HunyuanVideo DiT rounded to "
                "fp8-representable values (layerwise casting)",
        "config": {"workload": f"HunyuanVideo LoRA rank={args.rank} SFT step, fp8 weight storage / bf16 compute, 61x544x960 clip ({S} video + {T} text tokens), batch 1 per GPU, "
                               f"{nd} dual-stream + {ns} single-stream blocks (BASELINE configs[4])" + ("" if full else " -- REDUCED depth"),
                   "model": "HunyuanVideo DiT: 20 dual + 40 single blocks, width 3072, 24 x 128 heads, 12.8 B frozen parameters", "seq_len": N,
                   "activation_checkpointing": ckpt, "weight_storage": "float8_e4m3fn bytes in HBM, per-block up-cast into a shared bf16 arena",
                   "orchestration": ("one C call per block and direction (ftmi_hy_single_* / ftmi_hy_dual_*)" if os.environ.get("FTMI_NATIVE_BLOCKS", "1") != "0" else "python, per kernel over the C ABI")},
        "layers": nd + ns,
        "dual_single": (nd, ns, dual, single),
    }


def cpu_baseline_hunyuan(args, ctx) -> Dict[str, Any]:
    from oracle import hunyuan as hy
    from oracle import ltx

    nd, ns, dual, single = ctx["dual_single"]
    cfg = hy.HunyuanVideoConfig(num_layers=0, num_single_layers=1, num_refiner_layers=1)
    torch.manual_seed(0)
    oblk = hy.SingleStreamBlock(cfg).to(bf16)
    for p in oblk.parameters():
        p.requires_grad_(False)
    for t in ("to_q", "to_k", "to_v"):
        setattr(oblk.attn, t, ltx.LoraLinear(getattr(oblk.attn, t), args.rank, float(args.rank)))
    D, S, T = cfg.inner_dim, 32640, 256
    g = torch.Generator().manual_seed(0)
    video, text, temb = torch.randn(1, S, D, generator=g).to(bf16), torch.randn(1, T, D, generator=g).to(bf16), torch.randn(1, D, generator=g).to(bf16)
    ang = torch.rand(S, 64, generator=g) * 6.283
    rope = (ang.cos().repeat_interleave(2, dim=1).float().contiguous(), ang.sin().repeat_interleave(2, dim=1).float().contiguous())

    def fb():
        for p_ in oblk.parameters():
            p_.grad = None
        vr, tr_ = video.clone().requires_grad_(True), text.clone().requires_grad_(True)
        hv, ht = oblk(vr, tr_, temb, None, rope)
        torch.autograd.backward([hv, ht], [torch.ones_like(hv), torch.ones_like(ht)])

    per_single, ts = _time_block(fb, warm=0, timed=1)  # ~70 TFLOP per pass: one un-warmed pass keeps the sample inside the baseline's time budget
    per_step = ns * per_single + nd * per_single * (dual / single)
    return {"value": 1.0 / per_step, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle single-stream block forward + backward at the full {S} + {T} tokens, 1 un-warmed pass = {per_single:.1f} s; step = {ns} single blocks + {nd} dual-stream "
                      f"blocks priced at {dual / single:.2f} x a single block (their algorithmic FLOP ratio) = {per_step:.0f} s per sample-step"}


WORKLOADS = {"cogvideox": (build_cogvideox, lambda a, c: cpu_baseline_cogvideox(a, c["layers"])),
             "wan": (build_wan, lambda a, c: cpu_baseline_wan(a, c["layers"])),
             "hunyuan": (build_hunyuan, cpu_baseline_hunyuan)}
