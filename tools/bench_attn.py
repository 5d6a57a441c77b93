"""Micro-benchmark of the attention kernels at the LTX shapes."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for (B, H, Sq, Sk, bias) in [(2, 32, 2688, 2688, False), (2, 32, 2688, 128, True)]:
    mk = lambda s: torch.randn((B, s, H, 64), device=dev, generator=g).to(torch.bfloat16).permute(0, 2, 1, 3)
    q, k, v, do = mk(Sq), mk(Sk), mk(Sk), mk(Sq)
    kb = None
    if bias:
        kb = torch.zeros(B, Sk, device=dev); kb[:, 96:] = -9984.0
    out, lse = ops.attn_fwd(q, k, v, kb)
    for name, fn, fl in (("fwd", lambda: ops.attn_fwd(q, k, v, kb), 4.0), ("bwd", lambda: ops.attn_bwd(q, k, v, out, lse, do, kb), 10.0)):
        for _ in range(2): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / n
        print(f"attn {name} B{B} H{H} {Sq}x{Sk}: {ms*1e3:8.1f} us  {fl*B*H*Sq*Sk*64/ms/1e9:7.1f} TF/s (algorithmic)", flush=True)
