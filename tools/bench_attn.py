"""A/B timing of attention kernel variants (experimental build: FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build) on the cfg-2
shapes, interleaved rounds inside one process (guide rule 24).  Usage: python tools/bench_attn.py [fwd|bwd] v0 v1 ..."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
variants = [int(v) for v in sys.argv[2:]] or [0]
B, H, S, D = 2, 32, 2688, 2048
g = torch.Generator(device=dev).manual_seed(0)
qkv = torch.randn((B, S, 3, H, 64), generator=g, device=dev).to(torch.bfloat16)
q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
dout = torch.randn((B, S, H, 64), generator=g, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
env = "FTMI_ATTN_FWD" if which == "fwd" else "FTMI_ATTN_BWD"
flops = 4.0 * B * H * S * S * 64 * (1.0 if which == "fwd" else 2.5)


os.environ["FTMI_ATTN_FWD"] = "0"
out0, lse0 = ops.attn_fwd(q, k, v)


def run(var):
    os.environ[env] = str(var)
    if which == "fwd":
        return ops.attn_fwd(q, k, v)
    return ops.attn_bwd(q, k, v, out0, lse0, dout)


ref = run(0)
torch.cuda.synchronize()
times = {v_: [] for v_ in variants}
for rnd in range(7):
    for var in variants:
        for _ in range(2):
            run(var)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            res = run(var)
        e1.record()
        torch.cuda.synchronize()
        times[var].append(e0.elapsed_time(e1) / 10 * 1e3)
for var in variants:
    res = run(var)
    torch.cuda.synchronize()
    errs = [((a.float() - b.float()).norm() / b.float().norm()).item() for a, b in zip(res if which == "bwd" else res[:1], ref if which == "bwd" else ref[:1])]
    t = sorted(times[var])
    med = t[len(t) // 2]
    print(f"{which} variant {var:3d}: median {med:8.1f} us  min {t[0]:8.1f} us  -> {flops / med / 1e6:7.1f} TF/s   rel-diff vs v0 {max(errs):.2e}")
