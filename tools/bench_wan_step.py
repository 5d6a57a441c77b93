"""BASELINE config 4 on ONE GPU: Wan2.1-T2V-1.3B FULL fine-tune optimisation step, 81 x 512 x 512 clip (latents [1, 16, 21, 64, 64]: 21 504 video
tokens of width 1536, 512 text tokens), 30 blocks, random-init weights of the architecture, synthetic posterior moments / text embeddings, bf16
parameters and moments, fp32 gradients, nothing recomputed.  The config's FSDP-2 sharding needs 8 GPUs (the driver's scaling run); on one GPU the same
step object runs with whole shards and no collectives.  Not the bench.py line (that is BASELINE's metric on configs[1]).
    python tools/bench_wan_step.py [steps] [layers] [--cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 tools/bench_wan_step.py   # sharded over 8 GPUs"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd.wan import MI355XWanFullFinetuneStep, MI355XWanTransformer3DModel, WanTransformerConfig  # noqa: E402

# one process per GPU under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment): the parameters are sharded over the
# ranks, every rank trains on its own clip, the time is the maximum over the ranks and the rate counts all ranks' samples
world = int(os.environ.get("WORLD_SIZE", "1"))
par = None
if world > 1:
    from finetrainers_amd.parallel import DataParallelBackend  # noqa: E402

    par = DataParallelBackend()
dev = par.device if par is not None else torch.device("cuda", 0)
rank = par.rank if par is not None else 0
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
steps = int(argv[0]) if len(argv) > 0 else 3
layers = int(argv[1]) if len(argv) > 1 else 30
bf16 = torch.bfloat16
cfg = WanTransformerConfig(num_layers=layers)
model = MI355XWanTransformer3DModel(cfg, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
D = cfg.inner_dim
with torch.no_grad():
    def init(views):
        for name, v in views.items():
            if name.endswith("weight") and v.dim() >= 2:
                v.copy_((torch.randn(v.shape, generator=g, device=dev) / v.shape[-1] ** 0.5).to(bf16))
            elif "norm" in name and name.endswith("weight"):
                v.fill_(1.0)
            elif "scale_shift_table" in name:
                v.copy_((torch.randn(v.shape, generator=g, device=dev) / D ** 0.5).to(bf16))
            else:
                v.copy_((0.02 * torch.randn(v.shape, generator=g, device=dev)).to(bf16))
    init(model.state_dict_views())
step = MI355XWanFullFinetuneStep(model, lr=1e-5, betas=(0.9, 0.95), weight_decay=1e-4, generator=torch.Generator(device=dev).manual_seed(1 + rank), parallel=par)
g.manual_seed(100 + rank)  # every rank its own clip
B, C, F_, H, W, T = 1, 16, 21, 64, 64, 512
moments = torch.randn((B, 2 * C, F_, H, W), generator=g, device=dev).to(bf16)
moments[:, C:] = (moments[:, C:].float() * 0.3 - 2.0).to(bf16)
text = torch.randn((B, T, cfg.text_dim), generator=g, device=dev).to(bf16)
mean, std = torch.zeros(C, device=dev), torch.ones(C, device=dev)
sig = torch.tensor([0.6], device=dev)
for _ in range(2):
    out = step.step(moments, text, mean, std, sig)
if par is not None:
    par.wait_for_everyone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = step.step(moments, text, mean, std, sig)
if par is not None:
    par.wait_for_everyone()
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
if par is not None:
    t = torch.tensor([ms], device=dev)
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    ms = t.item()
    if rank != 0:
        par.destroy()
        sys.exit(0)
S, L, Fd = 21 * 32 * 32, layers, cfg.ffn_dim
lin = 2.0 * S * (6 * D * D + 2 * D * Fd) + 2.0 * T * 2 * D * D              # per block, forward: q|k|v, out, cross q, cross out, feed-forward; text k|v
att = 4.0 * S * S * D + 4.0 * S * T * D                                       # self + cross attention, forward
flop = L * (3.0 * lin + 3.5 * att)                                            # linears: forward + input gradient + weight gradient; attention backward = 2.5 x forward
print(f"Wan2.1-T2V-1.3B full fine-tune step, 81x512x512 ({S} video + {T} text tokens), {L} blocks, batch 1 per GPU, {world} GPU(s): {ms:.1f} ms/step = {world * 1e3 / ms:.3f} samples/s; "
      f"{flop / ms / 1e9:.0f} TF/s algorithmic per GPU = {flop / ms / 1e9 / 2500:.3f} of the dense bf16 peak; loss {out['loss'].item():.4f} grad_norm {out['grad_norm'].item():.4e}; "
      f"peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
line = ({"metric": "train samples/sec (+ step ms) Wan-T2V-1.3B full fine-tune 81x512x512 (BASELINE configs[3], one GPU, unsharded)", "value": world * 1e3 / ms,
                  "unit": "samples/s", "n_gpus": world, "steps": steps, "warmup": 2, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                  "dtype": "bf16", "data": "synthetic posterior moments [1,32,21,64,64] + random text embeds [1,512,4096], random-init weights of the Wan2.1-T2V-1.3B DiT",
                  "config": {"workload": f"Wan-T2V-1.3B full fine-tune bf16 step, 81x512x512 clip ({S} video + {T} text tokens), batch 1, {L} blocks",
                             "global_batch": world, "seq_len": S, "parallelism": f"fsdp{world} (parameters sharded per unit)" if world > 1 else "dp1 (whole shards)", "activation_checkpointing": False},
                  "step_tflop_algorithmic": flop / 1e12, "mfma_utilisation_step": flop / ms / 1e9 / 2500, "final_loss": out["loss"].item(),
                  "peak_memory_gib": torch.cuda.max_memory_allocated() / 2**30})
if "--cpu-baseline" in sys.argv:
    # the oracle (CPU restatement of the reference step, kind "port") on the box's host threads on a bounded sample of the same workload: ONE block forward +
    # backward (all parameter gradients) at the full 21 504 + 512 tokens, 1 warm-up + 1 timed, scaled by the block count.  tools/ may import oracle/ for
    # exactly this (it is the checker and the baseline, never the product).
    from oracle import wan  # noqa: E402

    oblk = wan.WanTransformerBlock(wan.WanConfig(num_layers=1)).to(bf16)
    gcpu = torch.Generator().manual_seed(0)
    xv = torch.randn(1, S, D, generator=gcpu).to(bf16)
    ev = torch.randn(1, T, D, generator=gcpu).to(bf16)
    tv = torch.randn(1, 6, D, generator=gcpu).to(bf16)
    ang = torch.rand(S, 64, generator=gcpu, dtype=torch.float64) * 6.283
    freqs = torch.polar(torch.ones_like(ang), ang).view(1, 1, S, 64)
    times = []
    for it in range(2):
        for p_ in oblk.parameters():
            p_.grad = None
        xr = xv.clone().requires_grad_(True)
        t0 = time.perf_counter()
        oblk(xr, ev, tv, freqs).backward(torch.ones_like(xv))
        times.append(time.perf_counter() - t0)
    per_block = times[-1]
    print(f"cpu_baseline (kind port, {torch.get_num_threads()} threads): oracle block forward + backward at {S} + {T} tokens {per_block:.1f} s (warm-up {times[0]:.1f} s) "
          f"-> x{L} blocks = {per_block * L:.0f} s per sample-step = {1.0 / (per_block * L):.5f} samples/s; GPU / CPU = {per_block * L * 1e3 / ms:.0f}x")
    line["cpu_baseline"] = {"value": 1.0 / (per_block * L), "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
                            "sample": f"oracle WanTransformerBlock forward + backward at the full {S} + {T} tokens, 1 warm-up + 1 timed = {per_block:.1f} s, scaled x{L} blocks"}
print(json.dumps(line))
if par is not None:
    par.destroy()
