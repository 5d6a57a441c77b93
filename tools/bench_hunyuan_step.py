"""BASELINE config 5's shape on ONE GPU: HunyuanVideo LoRA r = 64 SFT optimisation step, 61 x 544 x 960 clip (latents [1, 16, 16, 68, 120]: 32 640 video tokens of
width 3072 + 256 text tokens), random-init weights of the architecture with fp8-representable values (layerwise casting), synthetic latents / text embeddings.
The block composition is still Python over the C ABI and keeps every activation: the full 20 + 40 blocks need ~225 GB of activations at this shape, so the
default here is HALF the depth (10 dual + 20 single blocks; 120 GiB) -- a first indication per block; `2 20 40` runs the full model (236 GiB peak on the 288 GB part).
    python tools/bench_hunyuan_step.py [steps] [dual_blocks] [single_blocks]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29513 tools/bench_hunyuan_step.py 2 20 40   # config 5's DP = 8"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd.hunyuan_video import HunyuanVideoTransformerConfig, MI355XHunyuanVideoSFTStep, MI355XHunyuanVideoTransformer3DModel  # noqa: E402

# one process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment): every rank trains on its own clip, the
# LoRA gradients are averaged over the ranks, the time is the maximum over the ranks and the rate counts all ranks' samples
world = int(os.environ.get("WORLD_SIZE", "1"))
par = None
if world > 1:
    from finetrainers_amd.parallel import DataParallelBackend  # noqa: E402

    par = DataParallelBackend()
dev = par.device if par is not None else torch.device("cuda", 0)
rank = par.rank if par is not None else 0
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(argv[0]) if len(argv) > 0 else 2
nd = int(argv[1]) if len(argv) > 1 else 10
ns = int(argv[2]) if len(argv) > 2 else 20
bf16 = torch.bfloat16
cfg = HunyuanVideoTransformerConfig(num_layers=nd, num_single_layers=ns)
model = MI355XHunyuanVideoTransformer3DModel(cfg, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
with torch.no_grad():
    from finetrainers_amd import ops  # noqa: E402

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
model.add_adapter(r=64, lora_alpha=64.0)
with torch.no_grad():
    for p in model.lora_parameters()[1::2]:
        p.normal_(0, 0.01, generator=g)  # B != 0 so every gradient path carries data
step = MI355XHunyuanVideoSFTStep(model, lr=2e-5, guidance=1.0, generator=torch.Generator(device=dev).manual_seed(1 + rank), parallel=par)
g.manual_seed(100 + rank)  # every rank its own clip
B, C, F_, H, W, T = 1, 16, 16, 68, 120, 256
lat = torch.randn((B, C, F_, H, W), generator=g, device=dev).to(bf16)
mask = torch.ones(B, T, dtype=torch.long, device=dev)
mask[:, 200:] = 0
cond = {"encoder_hidden_states": torch.randn((B, T, cfg.text_embed_dim), generator=g, device=dev).to(bf16), "encoder_attention_mask": mask,
        "pooled_projections": torch.randn((B, cfg.pooled_projection_dim), generator=g, device=dev).to(bf16)}
sig = torch.tensor([0.6], device=dev)
out = step.step(lat, cond, sig)
if par is not None:
    par.wait_for_everyone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = step.step(lat, cond, sig)
if par is not None:
    par.wait_for_everyone()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
if par is not None:
    tmax = torch.tensor([ms], device=dev)
    torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
    ms = tmax.item()
    if rank != 0:
        par.destroy()
        sys.exit(0)
S, D, N = 16 * 34 * 60, cfg.inner_dim, 16 * 34 * 60 + T
mlp = int(D * cfg.mlp_ratio)
dual = 2.0 * S * (4 * D * D + 2 * D * mlp) * 2 + 2.0 * T * (4 * D * D + 2 * D * mlp) * 2 + 4.0 * N * N * D * 3.5
single = 2.0 * N * (3 * D * D + D * mlp + (D + mlp) * D) * 2 + 4.0 * N * N * D * 3.5
flop = nd * dual + ns * single  # linears forward + input gradient, attention forward + 2.5 x backward (LoRA / front / head terms omitted)
print(f"HunyuanVideo LoRA r=64 SFT step at config 5's shape ({S} video + {T} text tokens), {nd} dual + {ns} single blocks of 20 + 40, batch 1 per GPU, {world} GPU(s): {ms:.1f} ms/step = {world * 1e3 / ms:.3f} samples/s; "
      f"{flop / ms / 1e9:.0f} TF/s algorithmic = {flop / ms / 1e9 / 2500:.3f} of the dense bf16 peak; x {60 / (nd + ns):.1f} for the full depth ~ {ms * 60 / (nd + ns) / 1e3:.1f} s/step; "
      f"loss {out['loss'].item():.4f} grad_norm {out['grad_norm'].item():.4e}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
full = (nd, ns) == (20, 40)
print(json.dumps({"what": "HunyuanVideo LoRA r=64 SFT step at BASELINE configs[4]'s shape, one GPU, batch 1" + ("" if full else ", REDUCED depth (not a config-5 measurement)"),
                  "samples_per_s": world * 1e3 / ms if full else None, "n_gpus": world, "dual_blocks": nd, "single_blocks": ns,
                  "ms_per_step": ms, "tflops_algorithmic": flop / ms / 1e9, "mfma_utilisation_step": flop / ms / 1e9 / 2500, "peak_memory_gib": torch.cuda.max_memory_allocated() / 2**30,
                  "final_loss": out["loss"].item()}))
if par is not None:
    par.destroy()
