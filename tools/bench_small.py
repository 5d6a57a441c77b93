"""Micro-benchmark of the LoRA-sized GEMMs (skinny NT and TN) at M = 5376."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
M = 5376
def timeit(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x = torch.randn((M, 2048), device=dev, generator=g).to(torch.bfloat16)
for N in (64, 192):
    w = (torch.randn((N, 2048), device=dev, generator=g) / 45).to(torch.bfloat16)
    for v in (1, 0):
        us = timeit(lambda: ops.gemm_nt(x, w, None, variant=v))
        print(f"nt M{M} N{N} K2048 variant {v} ({'skinny' if v else '128x64 tiles'}): {us:7.1f} us", flush=True)
xa = torch.randn((M, 64), device=dev, generator=g).to(torch.bfloat16)
xa3 = torch.randn((M, 192), device=dev, generator=g).to(torch.bfloat16)
out = torch.zeros((2048, 64), device=dev); out2 = torch.zeros((64, 2048), device=dev); out3 = torch.zeros((192, 2048), device=dev)
print(f"tn dB  P2048 Q64  : {timeit(lambda: ops.gemm_tn(x, xa, out=out)):7.1f} us")
print(f"tn dA  P64 Q2048  : {timeit(lambda: ops.gemm_tn(xa, x, out=out2)):7.1f} us")
print(f"tn dA3 P192 Q2048 : {timeit(lambda: ops.gemm_tn(xa3, x, out=out3)):7.1f} us")
print(f"copy 22MB (torch) : {timeit(lambda: x.clone()):7.1f} us")
