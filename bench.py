#!/usr/bin/env python
"""bench.py -- LTX-Video LoRA SFT step on MI355X (BASELINE.json metric: train samples/sec + step ms,
49x512x768 clip, batch 2 per GPU, rank-64 LoRA, bf16, at 1/2/4/8 GPUs).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3          (spawns its 8 ranks itself: one process per GPU, RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
    python bench.py --workload {cogvideox,wan,hunyuan}      (BASELINE configs[2] / [3] / [4] through the same harness and JSON schema; default: ltx = configs[1])

One "step" = one full optimisation step of the reference's SFTTrainer._train body on synthetic latents already
resident in HBM: noise/flow-match mix/pack -> 28-block DiT forward -> weighted MSE -> backward (LoRA grads) ->
[DP: bucketed all-reduce (AVG) of the LoRA gradients over RCCL, overlapped with the backward] -> global-norm clip -> AdamW -> refresh of
the bf16 (hi, lo) LoRA working copies.
Nothing is skipped or cached between steps; weights are random-init of the production architecture (no checkpoints
are reachable offline).  Prints ONE JSON line on rank 0.
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic FLOPs per sample (multiply-add = 2), SURVEY section 8d / BASELINE.md: no activation recompute
STEP_TFLOP_PER_SAMPLE = 24.93
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--workload", choices=["ltx", "cogvideox", "wan", "hunyuan"], default="ltx",
                    help="ltx = BASELINE configs[1] (the metric's workload); cogvideox / wan / hunyuan = configs[2] / [3] / [4] (tools/bench_workloads.py)")
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: 30 for ltx -- ~2 s of GPU time, long enough that one box-level stall does not move the mean -- 10 otherwise)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=2, help="per-GPU batch (BASELINE configs[1]: 2; the other workloads run batch 1)")
    ap.add_argument("--layers", type=int, default=0, help="0 = the architecture's own depth (ltx 28, cogvideox 30, wan 30, hunyuan 20 + 40)")
    ap.add_argument("--rank", type=int, default=64)
    ap.add_argument("--frames", type=int, default=7, help="latent frames  (49 px frames / 8 + 1)")
    ap.add_argument("--height", type=int, default=16, help="latent height (512 / 32)")
    ap.add_argument("--width", type=int, default=24, help="latent width  (768 / 32)")
    ap.add_argument("--gemm-variant", type=int, default=int(os.environ.get("FTMI_GEMM_VARIANT", "8")))
    ap.add_argument("--gradient-checkpointing", action="store_true",
                    help="ltx / hunyuan: every block keeps its input and recomputes its forward inside the backward (the reference's --gradient_checkpointing)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=["full", "quick"], default="full",
                    help="full: SURVEY 8d protocol (1 warm-up + 3 timed steps; cfg 2 on --cpu-baseline-layers blocks in bf16 and fp32, cfg 1 at full depth); "
                         "quick: one un-warmed cfg-2 step on 2 blocks")
    ap.add_argument("--cpu-baseline-layers", type=int, default=4)
    ap.add_argument("--no-prof", action="store_true", help="do not bracket kernels with HIP events inside the timed region")
    ap.add_argument("--prof-stride", type=int, default=29, help="bracket every N-th launch of a kernel class with HIP events (1 = all)")
    a = ap.parse_args()
    if a.steps <= 0:
        a.steps = 30 if a.workload == "ltx" else 10
    if a.workload == "ltx" and a.layers <= 0:
        a.layers = 28
    if a.gradient_checkpointing and a.workload not in ("ltx", "hunyuan"):
        ap.error("--gradient-checkpointing: the ltx and hunyuan workloads recompute (cogvideox and wan keep their activations: 45 / 74 GiB of 288)")
    return a


def _profile_json(name: str):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except Exception:
        return None


NT_CLASS = "gemm_nt (all NT GEMM kernels)"


PMC_TRAFFIC_FILES = ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_b_pmc_traffic.json", "r02_a_pmc_traffic.json", "r01_pmc_traffic.json")
PMC_MFMA_FILES = ("r06_pmc_mfma.json", "r05_pmc_mfma.json", "r04_pmc_mfma.json", "r03_pmc_mfma.json", "r02_b_pmc_mfma.json", "r02_a_pmc_mfma.json")
PEAK_HBM_GBPS = 8000.0  # HBM3E spec, /opt/skills/guides/MI355X_MICROARCH.md (6.3 TB/s is what a streaming copy reaches)


def _pmc_traffic(kernel: str):
    """PMC counters cannot be read from inside the process being measured: the per-launch HBM traffic of the dominant kernel is
    taken from the committed rocprofv3 --pmc summary of this same command (tools/gpu_pmc.sh); None if absent."""
    for name in PMC_TRAFFIC_FILES:
        d = _profile_json(name)
        for key in (NT_CLASS, kernel):  # round 4 on: one entry for every NT GEMM launch (16x16x32 and 32x32x16 kernels together)
            if d and key in d:
                return d[key].get("hbm_bytes_per_launch"), name
    return None, None


def _pmc_step_traffic():
    """HBM / fabric bytes of ONE optimisation step from the committed counter passes of this command (FETCH_SIZE x 2 + WRITE_SIZE per kernel class x its
    launches, divided by the steps in the trace: 3).  Round-5 files carry the sum as `step.hbm_bytes_per_step`; older ones are summed here."""
    for name in PMC_TRAFFIC_FILES:
        d = _profile_json(name)
        if not d:
            continue
        if isinstance(d.get("step"), dict) and d["step"].get("hbm_bytes_per_step"):
            return float(d["step"]["hbm_bytes_per_step"]), name
        tot = sum(v["hbm_bytes_per_launch"] * v["launches_in_trace"] for k, v in d.items()
                  if isinstance(v, dict) and k != NT_CLASS and "hbm_bytes_per_launch" in v and "launches_in_trace" in v)
        if tot:
            return tot / 3.0, name
    return None, None


def _cpu_steps(ltx, cfg, rank, dtype, B, F_, H_, W_, warm, timed):
    model = ltx.build_model(cfg, seed=0, rank=rank, alpha=float(rank), dtype=dtype)
    opt = ltx.make_optimizer(model)
    inp = ltx.synth_inputs(cfg, B, F_, H_, W_, seed=1, mask_lens=[32, 96][:B], sigmas=[0.7, 0.25][:B], dtype=dtype)
    for _ in range(warm):
        ltx.sft_step(model, opt, inp)
    ts = []
    for _ in range(timed):
        t0 = time.time()
        ltx.sft_step(model, opt, inp)
        ts.append(time.time() - t0)
    return sum(ts) / len(ts), ts


def cpu_baseline(args):
    """The reference path (its CPU restatement, oracle/ltx.py -- the real SFTTrainer cannot be imported here: no diffusers / peft) timed on
    this box's host cores, SURVEY 8d protocol: all cores, no activation checkpointing, 1 warm-up + 3 timed optimisation steps
    (forward + backward + clip + AdamW).  A BOUNDED sample: cfg 2 (clip 49x512x768, 2688 tokens) runs batch 1 on a few of the 28
    identical blocks and is scaled linearly in blocks (embeddings / tail are < 1 % of the work) -- in bf16 (the reference's dtype)
    and in fp32; cfg 1 (9x128x128, 32 tokens, batch 1) is measured at full depth.  `value` is cfg 2 / bf16 in the metric's unit."""
    from oracle import ltx

    torch.manual_seed(0)
    cores = torch.get_num_threads()
    if args.cpu_baseline == "quick":
        nl = 2
        dt, _ = _cpu_steps(ltx, ltx.LTXConfig.production(num_layers=nl), args.rank, torch.bfloat16, 1, args.frames, args.height, args.width, 0, 1)
        return {"value": 1.0 / (dt * 28.0 / nl), "unit": "samples/s", "cores": cores, "kind": "port",
                "sample": f"QUICK: oracle, 1 un-warmed step, batch 1, cfg-2 clip, {nl} of 28 blocks = {dt:.1f} s, scaled x{28 / nl:g}", "step_s_measured": dt}
    nl = args.cpu_baseline_layers
    cfg = ltx.LTXConfig.production(num_layers=nl)
    out = {"unit": "samples/s", "cores": cores, "kind": "port"}
    dt_bf, ts_bf = _cpu_steps(ltx, cfg, args.rank, torch.bfloat16, 1, args.frames, args.height, args.width, 1, 3)
    dt_32, ts_32 = _cpu_steps(ltx, cfg, args.rank, torch.float32, 1, args.frames, args.height, args.width, 1, 3)
    dt_c1, ts_c1 = _cpu_steps(ltx, ltx.LTXConfig.production(num_layers=28), args.rank, torch.bfloat16, 1, 2, 4, 4, 1, 3)
    scale = 28.0 / nl
    out["value"] = 1.0 / (dt_bf * scale)  # this run's bounded sample (the once-measured full-size figure rides along as full_size_committed)
    out["sample"] = (f"oracle (CPU restatement of the reference step), {cores} threads, 1 warm-up + 3 timed optimisation steps each: cfg 2 clip 49x512x768 "
                     f"(2688 tokens) batch 1 on {nl} of 28 blocks, scaled x{scale:g}: bf16 {dt_bf:.2f} s/step -> {dt_bf * scale:.1f} s per sample-step, "
                     f"fp32 {dt_32:.2f} s/step -> {dt_32 * scale:.1f} s; cfg 1 (9x128x128, 32 tokens, batch 1) at full depth, bf16: {dt_c1 * 1e3:.0f} ms/step")
    out["cfg2_bf16"] = {"step_s_measured": ts_bf, "blocks": nl, "samples_per_s_scaled": 1.0 / (dt_bf * scale), "step_ms_scaled": dt_bf * scale * 1e3}
    out["cfg2_fp32"] = {"step_s_measured": ts_32, "blocks": nl, "samples_per_s_scaled": 1.0 / (dt_32 * scale), "step_ms_scaled": dt_32 * scale * 1e3}
    out["cfg1_bf16_full_depth"] = {"step_s_measured": ts_c1, "blocks": 28, "samples_per_s": 1.0 / dt_c1, "step_ms": dt_c1 * 1e3}
    full = _profile_json("r04_cpu_baseline_full.json")  # tools/cpu_baseline_full.py: cfg 2 exactly (batch 2, all 28 blocks), measured once, not scaled
    out["measured_in_this_run"] = True  # `value` and every *_measured list above were timed by THIS process on this box's host cores
    if full:
        # The configuration measured ONCE at FULL size (cfg 2 exactly: batch 2, 28 blocks -- 7 minutes of host time, tools/cpu_baseline_full.py) rides along as a
        # COMMITTED figure, labelled as such: `value` stays what this run timed within its bound (its x7 extrapolation is 6-10 % optimistic against it).
        out["full_size_committed"] = {
            "measured_in_this_run": False, "value": float(full["samples_per_s"]), "unit": "samples/s", "cores": full.get("cores"),
            "source": "profiles/r04_cpu_baseline_full.json (tools/cpu_baseline_full.py: oracle step at full size, 1 warm-up + 3 timed steps, round-4 box)",
            **{k: full[k] for k in ("step_s_measured", "step_s", "warmup_steps", "timed_steps") if k in full}}
    return out


def _self_spawn(args) -> int:
    """`python bench.py --gpus N` with N > 1 outside a launcher: re-execute under torch.distributed.run, one rank per GPU (the contract's
    own launch line), on a free local port.  RCCL needs HSA_ENABLE_IPC_MODE_LEGACY=0 on this pool; NCCL_P2P_DISABLE is never inherited."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.pop("NCCL_P2P_DISABLE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _build_ltx(args, par, dev):
    """BASELINE configs[1]: LTX-Video LoRA rank-64 SFT step, 49x512x768 clip, batch 2 per GPU."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification
    from finetrainers_amd.trainer import MI355XSFTStep

    tcfg = LTXTransformerConfig(num_layers=args.layers)
    spec = MI355XLTXVideoModelSpecification(transformer_config=tcfg, gemm_variant=args.gemm_variant)
    model = spec.load_diffusion_models(device=dev, random_init_seed=0)["transformer"]  # identical weights on every rank
    model.add_adapter(r=args.rank, lora_alpha=float(args.rank))
    if args.gradient_checkpointing:
        model.enable_gradient_checkpointing()
    with torch.no_grad():  # same LoRA init on every rank; B != 0 so every gradient path carries real data
        g = torch.Generator(device=dev).manual_seed(1)
        model.lora_flat.copy_(torch.randn(model.lora_flat.shape, generator=g, device=dev) * 0.01)
    step = MI355XSFTStep(model, spec, lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0, parallel=par,
                         generator=torch.Generator(device=dev).manual_seed(1234 + par.rank))

    # synthetic batch of the named clip shape, different data per rank (seed + rank), resident in HBM
    B, C = args.batch, tcfg.in_channels
    gd = torch.Generator(device=dev).manual_seed(100 + par.rank)
    latents = torch.randn((B, C, args.frames, args.height, args.width), generator=gd, device=dev).to(torch.bfloat16)
    text = torch.randn((B, 128, tcfg.caption_channels), generator=gd, device=dev).to(torch.bfloat16)
    mask = torch.zeros((B, 128), dtype=torch.bfloat16, device=dev)
    for i in range(B):
        mask[i, : (32 if i % 2 == 0 else 96)] = 1
    cond = {"encoder_hidden_states": text, "encoder_attention_mask": mask}
    lat = {"latents": latents, "latents_mean": torch.zeros(C, device=dev), "latents_std": torch.ones(C, device=dev),
           "num_frames": args.frames, "height": args.height, "width": args.width}

    S = args.frames * args.height * args.width
    full_shape = (args.layers == 28 and S == 2688 and args.rank == 64)
    if step.reducer is not None:
        step.reducer.measure_exposed = True
    return {
        "one_step": lambda: step.step(cond, lat),
        "reducer": step.reducer,
        "samples_per_step": B,
        "step_tflop": STEP_TFLOP_PER_SAMPLE * B * (args.layers / 28.0),
        "metric": "train samples/sec (+ step ms) LTX-Video LoRA 49x512x768 @1/2/4/8 MI355X",
        "data": "synthetic latents [B,128,7,16,24] + random text embeds, random-init weights of the production LTX-Video DiT",
        "config": {
            "workload": "LTX-Video LoRA rank=64 bf16 SFT step, 49x512x768 clip (latents 7x16x24 = 2688 tokens), batch 2 per GPU "
                        "(BASELINE configs[1])" if full_shape else f"REDUCED: layers={args.layers} tokens={S} rank={args.rank}",
            "model": "LTX-Video DiT 28 blocks, width 2048, 32x64 heads, 1.923B frozen bf16 params + 58.7M fp32 LoRA params",
            "seq_len": S,
            "activation_checkpointing": bool(args.gradient_checkpointing),
            "optimizer": "AdamW(lr 5e-5, betas (0.9,0.99), wd 1e-4) + clip 1.0, fused",
            "gemm_variant": args.gemm_variant,
        },
        "cpu_baseline": lambda: cpu_baseline(args),
    }


def _build_cpu_rehearsal(args, par, dev):
    """FTMI_BENCH_REHEARSAL_CPU=1: NO kernels, NO measurement -- the step is replaced by the exchange alone (the real GradBucketReducer driving real
    gloo collectives over a small flat buffer, bucketed exactly like the 28-block backward: 4 buckets of 7 blocks), so that everything AROUND the step
    that only exists at N > 1 -- self-spawn under torch.distributed.run, process group, broadcast, bucket schedule, barrier + max-over-ranks timing,
    the JSON schema with `exchange` / `exposed_comm_ms` / `buckets_per_step` -- runs end to end on a machine without a GPU (tests/test_host.py)."""
    from finetrainers_amd.parallel import GradBucketReducer

    if args.workload == "wan":
        return _build_cpu_rehearsal_wan(args, par, dev)
    L, per = 28, 64
    flat = torch.zeros(2 * L * per)
    par.broadcast_(flat, src=0)
    red = GradBucketReducer(par)
    red.measure_exposed = True
    ga, gb = flat[: L * per].view(L, per), flat[L * per:].view(L, per)
    state = {"n": 0}

    def one_step():
        ga.fill_(float(par.rank + 1))
        gb.fill_(-float(par.rank + 1))
        hi = L
        while hi > 0:
            lo = max(0, hi - 7)
            red.bucket_ready(lo, hi, ga[lo:hi], gb[lo:hi])
            hi = lo
        red.finish()
        want = (par.world_size + 1) / 2.0  # mean over ranks of (rank + 1)
        if abs(float(ga[0, 0]) - want) > 1e-6 or abs(float(gb[-1, -1]) + want) > 1e-6:
            raise RuntimeError(f"rehearsal exchange wrong on rank {par.rank}: {float(ga[0, 0])} / {float(gb[-1, -1])} vs {want}")
        state["n"] += 1
        return {"loss": torch.tensor(0.0)}

    return {"one_step": one_step, "reducer": red, "samples_per_step": args.batch, "step_tflop": 0.0,
            "metric": "train samples/sec (+ step ms) LTX-Video LoRA 49x512x768 @1/2/4/8 MI355X",
            "data": "NONE (CPU rehearsal of the multi-rank harness: no kernels run)",
            "config": {"workload": "REHEARSAL on CPU over gloo: the exchange of 4 x 7-block buckets only, no step"},
            "cpu_baseline": lambda: None}


def _build_cpu_rehearsal_wan(args, par, dev):
    """The same rehearsal for `--workload wan` (config 4: parameters sharded over the ranks): the real ParameterSharder walks the schedule of
    MI355XWanFullFinetuneStep's hooks (wan/trainer.py:62-76) over gloo -- root gathered once, every block gathered before its forward with the next one
    prefetched, gathered again before its backward (the last two are still resident), its fp32 gradient reduce-scattered (mean) after it -- on 30 small
    flat units, and checks every rank's shard of every averaged gradient.  No kernels, no measurement; the line carries the sharder's own counters."""
    from finetrainers_amd.wan.fsdp import ParameterSharder

    nb, n_root, n_blk = 30, 1000, 4104  # Wan2.1-T2V-1.3B has 30 blocks; sizes that do not divide by 8 ranks x 64 (padding path)
    g = torch.Generator().manual_seed(0)
    units = [torch.randn(n_root, generator=g).to(torch.bfloat16)] + [torch.randn(n_blk, generator=g).to(torch.bfloat16) for _ in range(nb)]
    full = [u.clone() for u in units]
    sh = ParameterSharder(units, ["root"] + [f"blocks.{i}" for i in range(nb)], par.world_size, par.rank, par.backend, force_collectives=par.world_size == 1)
    W = par.world_size

    def one_step():
        p0 = sh.acquire(0)
        if not torch.equal(p0, full[0]):
            raise RuntimeError(f"rehearsal: rank {par.rank} gathered a wrong root unit")
        sh.grad_buffer(0).fill_(float(par.rank + 1))
        for i in range(1, nb + 1):  # forward (_pre_forward)
            p = sh.acquire(i)
            sh.prefetch(i + 1)
            if not torch.equal(p, full[i]):
                raise RuntimeError(f"rehearsal: rank {par.rank} gathered a wrong unit {i} (forward)")
        for i in range(nb, 0, -1):  # backward (_pre_backward / _post_backward)
            p = sh.acquire(i)
            if i > 1:
                sh.prefetch(i - 1)
            if not torch.equal(p, full[i]):
                raise RuntimeError(f"rehearsal: rank {par.rank} gathered a wrong unit {i} (backward)")
            sh.grad_buffer(i).fill_(float((par.rank + 1) * i))
            sh.scatter_grad(i)
        sh.scatter_grad(0)
        sh.finish_gradients()
        mean = (W + 1) / 2.0
        for i, u in enumerate(sh.units):
            lo, hi = par.rank * u.k, min((par.rank + 1) * u.k, u.numel)
            want = mean * (i if i else 1)
            if hi > lo and not torch.allclose(u.shard_grad[: hi - lo], torch.full((hi - lo,), want)):
                raise RuntimeError(f"rehearsal: rank {par.rank} unit {i}: averaged gradient shard {float(u.shard_grad[0])} vs {want}")
        sh.zero_shard_grads()
        sh.release_all()
        return {"loss": torch.tensor(0.0)}

    return {"one_step": one_step, "reducer": None, "sharder": sh, "samples_per_step": args.batch, "step_tflop": 0.0,
            "metric": "train samples/sec (+ step ms) Wan-T2V full fine-tune 81x512x512, parameters sharded over the GPUs",
            "data": "NONE (CPU rehearsal of the multi-rank harness: no kernels run)",
            "config": {"workload": "REHEARSAL on CPU over gloo: the parameter sharder's gather / reduce-scatter schedule over 1 + 30 flat units, no step"},
            "cpu_baseline": lambda: None}


def main():
    args = parse()
    cpu_reh = os.environ.get("FTMI_BENCH_REHEARSAL_CPU") == "1"
    if not torch.cuda.is_available() and not cpu_reh:
        raise SystemExit("bench.py needs an MI355X (the MI355X backend has no CPU path)")

    from finetrainers_amd import _lib
    from finetrainers_amd.parallel import DataParallelBackend

    def dev_sync():
        if torch.cuda.is_available() and not cpu_reh:
            torch.cuda.synchronize()

    # FTMI_BENCH_SHARE_GPU=1: rehearsal of the multi-rank launch on a one-GPU box -- the N ranks time-share GPU 0 and exchange through gloo
    # (RCCL refuses two ranks on one device).  It executes the spawn, broadcast, bucketed exchange, barrier and max-over-ranks code; the
    # line it prints is marked "rehearsal" and its numbers mean nothing.
    share = os.environ.get("FTMI_BENCH_SHARE_GPU") == "1"
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if torch.cuda.device_count() < args.gpus and not share and not cpu_reh:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} MI355X visible")
        raise SystemExit(_self_spawn(args))
    if cpu_reh:
        par = DataParallelBackend(backend="gloo", device=torch.device("cpu"), exercise_collectives=True)
        args.no_prof, args.no_cpu_baseline = True, True
    else:
        par = DataParallelBackend(backend="gloo", device=torch.device("cuda", 0)) if share and args.gpus > 1 else DataParallelBackend()
    if par.world_size > 1 or cpu_reh:
        par.gather_rank_devices()  # collective, on every rank: one device record per rank for `exchange.rank_devices`
    if par.world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={par.world_size}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = par.device
    lib = _lib.load()

    if cpu_reh:
        ctx = _build_cpu_rehearsal(args, par, dev)
    elif args.workload == "ltx":
        ctx = _build_ltx(args, par, dev)
    else:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_workloads

        build, cpu_fn = bench_workloads.WORKLOADS[args.workload]
        ctx = build(args, par, dev)
        ctx["cpu_baseline"] = lambda: cpu_fn(args, ctx)
    one_step, B = ctx["one_step"], ctx["samples_per_step"]

    prof = not args.no_prof
    for w in range(args.warmup):
        if prof and w == args.warmup - 1:
            # the profiler is switched on for the LAST warm-up step: the first step that records events pays a one-off 25-90 ms
            # (first use of the event machinery; always step 0, never later), which must not land in the timed region
            lib.ftmi_prof_enable(args.prof_stride)
        one_step()
    dev_sync()
    par.wait_for_everyone()
    dev_sync()

    if prof:
        lib.ftmi_prof_enable(args.prof_stride)  # (also covers --warmup 0)
        for k in range(5):
            lib.ftmi_prof_summary(k, None, None, None, None, None, 1)  # drop what the warm-up step recorded
    # a cyclic-GC pause (tens of ms with torch's object graph) is invisible while the GPU queue is full but lands in full on the
    # first timed step, whose launches start from an empty queue: about one run in ten showed a +25 ms step 0 before this
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    out = None
    class _HostMark:  # CPU rehearsal: host clock instead of device events
        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    marks = [(_HostMark() if cpu_reh else torch.cuda.Event(enable_timing=True)) for _ in range(args.steps + 1)]  # per-step device time (diagnostics only)
    marks[0].record()
    for i in range(args.steps):
        out = one_step()
        marks[i + 1].record()
    dev_sync()
    par.wait_for_everyone()
    dev_sync()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if prof:
        lib.ftmi_prof_enable(0)

    reducer = ctx.get("reducer")
    exposed = reducer.exposed_ms() if reducer is not None else None
    t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if par.world_size > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    elapsed = t.item()
    loss = out["loss"].item()

    if par.rank == 0:
        ms = elapsed / args.steps * 1e3
        samples_per_s = par.world_size * B * args.steps / elapsed
        step_tflop = ctx["step_tflop"]
        res = {
            "metric": ctx["metric"],
            "value": samples_per_s,
            "unit": "samples/s",
            "n_gpus": par.world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms,
            "higher_is_better": True,
            "scaling": "weak",
            **({"rehearsal": "FTMI_BENCH_SHARE_GPU=1: all ranks on one GPU over gloo -- code-path check only, not a measurement"} if share and args.gpus > 1 else {}),
            **({"rehearsal": "FTMI_BENCH_REHEARSAL_CPU=1: no GPU, no kernels -- the multi-rank harness and the exchange over gloo only; not a measurement"} if cpu_reh else {}),
            "vs_baseline": None,
            "dtype": "bf16",
            "data": ctx["data"],
            "config": {**ctx["config"], "global_batch": par.world_size * B, "parallelism": ("fsdp" if args.workload == "wan" else "dp") + str(par.world_size)},
            "step_tflop_algorithmic": step_tflop,
            "peak_memory_gib": (torch.cuda.max_memory_allocated() / 2**30) if torch.cuda.is_available() else 0.0,
            "mfma_utilisation_step": step_tflop / (ms * 1e-3) / PEAK_BF16_TFLOPS,
            "final_loss": loss,
            # data parallelism: what the exchange ran on and how long the compute stream WAITED for it per step (events around the reducer's
            # finish(): the part of the bucketed all-reduce the backward did not cover; null on one GPU)
            # `exchange.group_size` / `rank_devices` / `distinct_devices` come from the communicator itself: N ranks on N distinct GPUs is checkable from the line
            "exchange": par.describe() if (par.world_size > 1 or cpu_reh) else None,
            "exposed_comm_ms": exposed,
            # True: the buckets went through the library's own RCCL communicator (ftmi_allreduce_bucket / _wait, FTMI_NATIVE_ALLREDUCE=1); False: torch.distributed's
            "exchange_in_library": (bool(getattr(reducer, "native", False)) if reducer is not None else None),
            "buckets_per_step": (reducer.buckets_issued / max(1, args.steps + args.warmup)) if reducer is not None else None,
            # parameter sharding (Wan): the sharder's own counters -- all-gathers (bf16 units) and reduce-scatters (fp32 gradients) issued per step
            **({"sharder": {"units": len(ctx["sharder"].units), "gathers_per_step": ctx["sharder"].gathers_issued / max(1, args.steps + args.warmup),
                            "scatters_per_step": ctx["sharder"].scatters_issued / max(1, args.steps + args.warmup), "world": ctx["sharder"].world}}
               if ctx.get("sharder") is not None else {}),
            "step_ms_min_median_max": [round(v, 3) for v in (lambda t: (t[0], t[len(t) // 2], t[-1]))(sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)))],
            # steps that took more than 1.2x the median (index, ms): isolated ~25 ms stalls appear about once in 10 s on the gpurun
            # boxes with and without the in-stream profiler (a box-level pause, not part of the step)
            "step_ms_outliers": [(i, round(t, 2)) for i, t in enumerate(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
                                 if t > 1.2 * sorted(marks[j].elapsed_time(marks[j + 1]) for j in range(args.steps))[args.steps // 2]][:16],
        }
        if prof:
            classes = {0: "gemm_nt", 1: "gemm_tn", 2: "attn_fwd", 3: "attn_bwd", 4: "gemm_nt_skinny"}
            kern = {}
            for k, name in classes.items():
                tms, n, fl, an, afl = ctypes.c_double(0), ctypes.c_long(0), ctypes.c_double(0), ctypes.c_long(0), ctypes.c_double(0)
                lib.ftmi_prof_summary(k, ctypes.byref(tms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(an), ctypes.byref(afl), 1)
                if n.value:
                    tflops = fl.value / (tms.value * 1e-3) / 1e12  # sampled launches: algorithmic FLOPs / event time
                    kern[name] = {"launches_per_step": an.value / args.steps, "sampled_launches": n.value,
                                  "avg_us": tms.value / n.value * 1e3, "tflops": tflops,
                                  "ms_per_step": afl.value / (tflops * 1e12) * 1e3 / args.steps}
            res["kernels"] = kern
            if args.workload != "ltx" and kern:
                # the dominant kernel class of this workload (largest share of the step) against the dense bf16 MFMA peak
                dom = max(kern, key=lambda k_: kern[k_]["ms_per_step"])
                d_ = kern[dom]
                names = {"gemm_nt": "ftmi::gemm_nt_kernel (bf16 MFMA GEMM + fused LoRA / epilogues)", "gemm_tn": "ftmi::gemm_tn2_kernel (weight gradients)",
                         "attn_fwd": "ftmi::attn_fwd_kernel", "attn_bwd": "ftmi::attn_bwd_dq / attn_bwd_dkdv kernels (flash attention backward)",
                         "gemm_nt_skinny": "ftmi::gemm_nt_skinny2_kernel"}
                res["roofline"] = {"kernel": names.get(dom, dom), "bound": "mfma", "achieved": d_["tflops"], "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                   "frac": d_["tflops"] / PEAK_BF16_TFLOPS, "traffic": None, "avg_launch_us": d_["avg_us"], "launches_per_step": d_["launches_per_step"],
                                   "share_of_step": d_["ms_per_step"] / ms,
                                   "note": f"achieved = sum of algorithmic FLOPs / sum of HIP-event durations over every {args.prof_stride}-th launch of the class, events "
                                           "recorded on the launch stream inside the timed region; attention backward counts 2.5 x the forward's 4 S^2 d per head (5 matmuls; "
                                           "the two-kernel backward executes 7 at head_dim 64, 8 at head_dim 128)"}
            if args.workload == "ltx" and "gemm_nt" in kern:
                g_ = kern["gemm_nt"]
                res["roofline"] = {
                    "kernel": "ftmi::gemm_nt16_kernel + ftmi::gemm_nt_kernel (bf16 MFMA NT GEMM + fused LoRA/epilogues; every Linear forward and dgrad -- the "
                              "16x16x32 hand-placed pipeline where N % 256 == 0 and M >= 1024, the 32x32x16 kernels elsewhere)",
                    "bound": "mfma",
                    "achieved": g_["tflops"],
                    "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s",
                    "frac": g_["tflops"] / PEAK_BF16_TFLOPS,
                    "traffic": _pmc_traffic("gemm_nt_kernel")[0],
                    "traffic_measured_in_this_run": False,  # a committed rocprofv3 --pmc pass of this command (counters cannot be collected inside the timed run)
                    "traffic_unit": f"HBM/fabric bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes; profiles/{_pmc_traffic('gemm_nt_kernel')[1]})",
                    "avg_launch_us": g_["avg_us"],
                    "launches_per_step": g_["launches_per_step"],
                    "share_of_step": g_["ms_per_step"] / ms,
                    "note": f"achieved = sum of algorithmic FLOPs (2*M*N*(K + r) per launch: the LoRA K-extension executes 3r deep -- hi / lo planes -- and counts r) / "
                            f"sum of HIP-event durations over every {args.prof_stride}-th launch of the kernel, events recorded on the launch stream inside the timed "
                            "region (the stride, coprime to the per-block launch counts, cycles through every shape; compare with the gemm_nt16_kernel<...> / gemm_nt_kernel<...> "
                            "rows of profiles/r04_kernel_stats.csv and profiles/r04_step_kernels.csv)",
                }
                mf_name = next((n for n in PMC_MFMA_FILES if _profile_json(n)), None)
                mf = _profile_json(mf_name) if mf_name else None
                if mf:  # counter-derived figures of the same command (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES ..., tools/gpu_profile_r02.sh)
                    keep = ("mfma_util", "valu_issue_share_of_simd_time", "wave_wait_share", "wave_issue_stall_share", "launches_in_trace")
                    res["roofline"]["counters"] = {k: mf.get(NT_CLASS, mf.get("gemm_nt_kernel", {})).get(k) for k in keep}
                    res["roofline"]["counters"]["measured_in_this_run"] = False
                    res["roofline"]["counters"]["source"] = (f"profiles/{mf_name} (rocprofv3 --pmc passes of this command, tools/gpu_profile_r04.sh): "
                                                             "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8)")
                    res["mfma_utilisation_counters"] = {"measured_in_this_run": False, "source": f"profiles/{mf_name}", "step": mf.get("step", {}).get("mfma_util_over_kernel_time"),
                                                        **{k: v.get("mfma_util") for k, v in mf.items() if isinstance(v, dict) and v.get("mfma_util")}}
                if "attn_fwd" in kern and "attn_bwd" in kern:
                    a_ms = kern["attn_fwd"]["ms_per_step"] + kern["attn_bwd"]["ms_per_step"]
                    a_fl = kern["attn_fwd"]["tflops"] * kern["attn_fwd"]["ms_per_step"] + kern["attn_bwd"]["tflops"] * kern["attn_bwd"]["ms_per_step"]
                    res["attention_roofline"] = {"achieved": a_fl / a_ms, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": a_fl / a_ms / PEAK_BF16_TFLOPS,
                                                 "share_of_step": a_ms / ms}
        if args.workload == "ltx":
            # north_star: "rocprof HBM-GB/s ... reported against peak".  Bytes from the committed counter passes of this command (FETCH_SIZE x 2 -- the gfx950
            # correction of MI355X_MICROARCH.md -- + WRITE_SIZE, separate passes), time from THIS run.
            step_bytes, src = _pmc_step_traffic()
            dom_bytes, dsrc = _pmc_traffic("gemm_nt_kernel")
            hb = {"peak": PEAK_HBM_GBPS, "unit": "GB/s", "source": f"profiles/{src}" if src else None,
                  "bytes_measured_in_this_run": False, "time_measured_in_this_run": True}  # bytes: committed counter pass; time: this run
            if step_bytes:
                hb["step"] = step_bytes / (ms * 1e-3) / 1e9
                hb["step_frac_of_peak"] = hb["step"] / PEAK_HBM_GBPS
                hb["step_gbytes"] = step_bytes / 1e9
            g_ = res.get("kernels", {}).get("gemm_nt") if prof else None
            if dom_bytes and g_:
                hb["dominant_kernel"] = dom_bytes / (g_["avg_us"] * 1e-6) / 1e9
                hb["dominant_kernel_frac_of_peak"] = hb["dominant_kernel"] / PEAK_HBM_GBPS
            res["hbm_gbps"] = hb
            res["hbm_frac_of_peak_committed_bytes"] = hb.get("step_frac_of_peak")  # (bytes from the committed counter pass / this run's time)
        # the same command with --no-prof on the round's evidence box: the instrument's cost as a stated quantity
        for rr in ("r06", "r05", "r04"):
            pr, pd = _profile_json(f"{rr}_bench_noprof.json"), _profile_json(f"{rr}_bench_default.json")
            if prof and pr and pd and args.workload == "ltx":
                res["ms_per_step_without_event_profiler"] = {
                    "measured_in_this_run": False, "ms_per_step": pr.get("ms_per_step"), "with_profiler_same_box": pd.get("ms_per_step"),
                    "source": f"profiles/{rr}_bench_noprof.json vs profiles/{rr}_bench_default.json: python bench.py with and without --no-prof on the round's evidence box "
                              f"({pd.get('ms_per_step', 0):.2f} vs {pr.get('ms_per_step', 0):.2f} ms)"}
                break
        if par.world_size == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = ctx["cpu_baseline"]()
            except Exception as e:  # the baseline must never take the bench line down
                res["cpu_baseline"] = {"value": None, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(res))
    par.destroy()


if __name__ == "__main__":
    main()
