"""A/B timing of attention kernel variants (experimental build: FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build --force) on the
cfg-2 shapes, interleaved rounds inside one process (guide rule 24).
  python tools/bench_attn.py fwd  "FTMI_ATTN_GEN=1" "FTMI_ATTN_GEN=2" ...
  python tools/bench_attn.py bwd  "FTMI_ATTN_DQ_GEN=1,FTMI_ATTN_DKV_GEN=1" "FTMI_ATTN_DQ_GEN=2,FTMI_ATTN_DKV_GEN=1" ...
  python tools/bench_attn.py xfwd / xbwd ...   (cross-attention shape: 2688 queries x 128 keys with a text mask)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
variants = sys.argv[2:] or [""]
cross = which.startswith("x")
bwd = which.endswith("bwd")
D = int(os.environ.get("HEAD_DIM", "64"))  # 128: forward only (Wan / HunyuanVideo head size)
B, H, S = (int(v_) for v_ in os.environ.get("BHS", "2,32,2688").split(","))
Sk = 128 if cross else S
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, S, H, D), generator=g, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
kv = torch.randn((B, Sk, 2, H, D), generator=g, device=dev).to(torch.bfloat16)
k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
dout = torch.randn((B, S, H, D), generator=g, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
bias = None
if cross:
    bias = torch.zeros((B, Sk), device=dev)
    bias[0, 32:] = -10000.0
    bias[1, 96:] = -10000.0
flops = 4.0 * B * H * S * Sk * D * (2.5 if bwd else 1.0)
KEYS = ("FTMI_ATTN_GEN", "FTMI_ATTN_FWD_GEN", "FTMI_ATTN_DQ_GEN", "FTMI_ATTN_DKV_GEN", "FTMI_ATTN_FWD", "FTMI_ATTN_FWD8", "FTMI_ATTN_BWD8")


def setenv(spec):
    for kk in KEYS:
        os.environ.pop(kk, None)
    for kvp in filter(None, spec.split(",")):
        a, b = kvp.split("=")
        os.environ[a] = b


setenv("FTMI_ATTN_FWD8=0,FTMI_ATTN_BWD8=0")
out0, lse0 = ops.attn_fwd(q, k, v, bias)
ref = ops.attn_bwd(q, k, v, out0, lse0, dout, bias) if bwd else (out0,)
torch.cuda.synchronize()


def run(spec):
    setenv(spec)
    if bwd:
        return ops.attn_bwd(q, k, v, out0, lse0, dout, bias)
    return ops.attn_fwd(q, k, v, bias)[:1]


times = {s_: [] for s_ in variants}
for rnd in range(7):
    for spec in variants:
        for _ in range(2):
            run(spec)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            res = run(spec)
        e1.record()
        torch.cuda.synchronize()
        times[spec].append(e0.elapsed_time(e1) / 10 * 1e3)
for spec in variants:
    res = run(spec)
    torch.cuda.synchronize()
    errs = [((a.float() - b.float()).norm() / b.float().norm()).item() for a, b in zip(res, ref)]
    t = sorted(times[spec])
    med = t[len(t) // 2]
    print(f"{which:5s} [{spec:45s}] median {med:8.1f} us  min {t[0]:8.1f} us -> {flops / med / 1e6:7.1f} TF/s   rel-diff vs gen 1: {max(errs):.2e}")
