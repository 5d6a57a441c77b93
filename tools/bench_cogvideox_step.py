"""BASELINE config 3 on one GPU: CogVideoX-2b LoRA r = 64 SFT optimisation step, 49 x 480 x 720 clip (latents [1, 13, 16, 60, 90]: 226 text + 17 550
video tokens), 30 blocks, random-init weights of the 2b architecture, synthetic latents / text embeddings, bf16 base + fp32-equivalent LoRA, nothing
recomputed.  Not the bench.py line (that is BASELINE's metric on configs[1]); this is config 3's measurement.
    python tools/bench_cogvideox_step.py [steps] [layers] [--cpu-baseline] [--python-blocks]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29512 tools/bench_cogvideox_step.py   # config 3's DP = 8
--cpu-baseline: also time the oracle (CPU restatement of the reference step, kind "port") on the box's host threads on a bounded sample of the same
workload -- ONE block forward + backward at the full 17 776 tokens, 1 warm-up + 1 timed, scaled by the block count (the embed / head / optimiser share of
the GPU step is < 2 %).  tools/ may import oracle/ for exactly this (it is the checker and the baseline, never the product)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd.cogvideox import (CogVideoXTransformerConfig, MI355XCogVideoXSFTStep, MI355XCogVideoXTransformer3DModel)  # noqa: E402

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
steps = int(argv[0]) if len(argv) > 0 else 5
layers = int(argv[1]) if len(argv) > 1 else 30
bf16 = torch.bfloat16
cfg = CogVideoXTransformerConfig(num_layers=layers)
model = MI355XCogVideoXTransformer3DModel(cfg, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
sd = {}


def rnd(shape, fan_in):
    return (torch.randn(shape, generator=g, device=dev) / fan_in ** 0.5).to(bf16)


D = cfg.inner_dim
for k, name in model._KEYS.items():
    shp = getattr(model, name).shape
    sd[k] = rnd(shp, shp[1]) if len(shp) == 2 else (torch.ones(shp, device=dev, dtype=bf16) if "norm" in k and k.endswith("weight") else 0.02 * rnd(shp, 1))
sd["patch_embed.proj.weight"] = rnd((D, cfg.in_channels, 2, 2), 64)
for i, blk in enumerate(model.transformer_blocks):
    for k, name in blk._KEYS.items():
        shp = getattr(blk, name).shape
        sd[f"transformer_blocks.{i}.{k}"] = rnd(shp, shp[1]) if len(shp) == 2 else (torch.ones(shp, device=dev, dtype=bf16) if "norm" in k and k.endswith("weight") else 0.02 * rnd(shp, 1))
model.load_diffusers_state_dict(sd)
del sd
model.add_adapter(r=64, lora_alpha=64.0)
with torch.no_grad():
    n = model.lora_flat.numel() // 2
    model.lora_flat[n:].normal_(0, 0.01, generator=g)  # B != 0 so every gradient path carries data
model.native_blocks = "--python-blocks" not in sys.argv  # default: all blocks in one C call per direction (csrc/cog_dit.hip)
step = MI355XCogVideoXSFTStep(model, lr=5e-5, betas=(0.9, 0.99), generator=torch.Generator(device=dev).manual_seed(1 + rank), parallel=par)
g.manual_seed(100 + rank)  # every rank its own clip
lat = torch.randn((1, 13, 16, 60, 90), generator=g, device=dev).to(bf16)
text = torch.randn((1, 226, 4096), generator=g, device=dev).to(bf16)
for _ in range(2):
    out = step.step(lat, text)
if par is not None:
    par.wait_for_everyone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = step.step(lat, text)
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
N, L = 226 + 17550, layers
flop = L * (2.0 * N * D * D * 12 * 2 + 4.0 * N * N * D * 3.5)  # linears forward + dgrad, attention forward + 2.5 x backward (LoRA / embed / head terms omitted)
print(f"CogVideoX-2b LoRA r=64 SFT step, 49x480x720 (226 + 17550 tokens), {L} blocks, batch 1 per GPU, {world} GPU(s): {ms:.1f} ms/step = {world * 1e3 / ms:.3f} samples/s; "
      f"{flop / ms / 1e9:.0f} TF/s algorithmic = {flop / ms / 1e9 / 2500:.3f} of the dense bf16 peak; loss {out['loss'].item():.4f} grad_norm {out['grad_norm'].item():.4e}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")

import json  # noqa: E402

line = {"metric": "train samples/sec (+ step ms) CogVideoX-2b LoRA 49x480x720 (BASELINE configs[2])", "value": world * 1e3 / ms, "unit": "samples/s", "n_gpus": world,
        "steps": steps, "warmup": 2, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic latents [1,13,16,60,90] + random text embeds [1,226,4096], random-init weights of the CogVideoX-2b DiT",
        "config": {"workload": f"CogVideoX-2b LoRA rank=64 bf16 SFT step, 49x480x720 clip (226 text + 17550 video tokens), batch 1 per GPU, {layers} blocks",
                   "global_batch": world, "seq_len": N, "parallelism": f"dp{world}", "activation_checkpointing": False, "orchestration": "C block stack (ftmi_cog_blocks_forward / _backward)" if model.native_blocks else "python, per block over the C ABI"},
        "step_tflop_algorithmic": flop / 1e12, "mfma_utilisation_step": flop / ms / 1e9 / 2500, "final_loss": out["loss"].item(),
        "peak_memory_gib": torch.cuda.max_memory_allocated() / 2**30}
if "--cpu-baseline" in sys.argv:
    from oracle import cogvideox as cvx  # noqa: E402

    ocfg = cvx.CogVideoXConfig(num_layers=1)
    omodel = cvx.build_model(ocfg, seed=0, rank=64, alpha=64.0, lora_b_std=0.02)
    oblk = omodel.transformer_blocks[0]
    gcpu = torch.Generator().manual_seed(0)
    vid = torch.randn(1, 17550, D, generator=gcpu).to(bf16)
    txt = torch.randn(1, 226, D, generator=gcpu).to(bf16)
    temb = torch.randn(1, 512, generator=gcpu).to(bf16)
    times = []
    for it in range(2):
        for p_ in oblk.parameters():
            p_.grad = None
        vr, tr_ = vid.clone().requires_grad_(True), txt.clone().requires_grad_(True)
        t0 = time.perf_counter()
        hv, ht = oblk(vr, tr_, temb)
        torch.autograd.backward([hv, ht], [torch.ones_like(hv), torch.ones_like(ht)])
        times.append(time.perf_counter() - t0)
    per_block = times[-1]
    print(f"cpu_baseline (kind port, {torch.get_num_threads()} threads): oracle block forward + backward at 17 776 tokens {per_block:.1f} s (warm-up {times[0]:.1f} s) "
          f"-> x{layers} blocks = {per_block * layers:.0f} s per sample-step = {1.0 / (per_block * layers):.5f} samples/s; GPU / CPU = {per_block * layers * 1e3 / ms:.0f}x")
    line["cpu_baseline"] = {"value": 1.0 / (per_block * layers), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": f"oracle CogVideoXBlock forward + backward at the full {N} tokens, 1 warm-up + 1 timed = {per_block:.1f} s, scaled x{layers} blocks"}
print(json.dumps(line))
if par is not None:
    par.destroy()
