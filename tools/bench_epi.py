"""Micro-benchmark of the fused GEMM epilogues on the feed-forward shapes (M = 5376)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops, _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
M = 5376
def timeit(fn, n=40):
    best = 1e9
    for _ in range(3):
        for _ in range(20): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best
for (N, K) in [(8192, 2048), (2048, 8192), (2048, 2048)]:
    x = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn((N,), device=dev, generator=g).to(torch.bfloat16)
    resid = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16)
    gate = torch.randn((2, N), device=dev, generator=g).to(torch.bfloat16)
    z = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16)
    fl = 2.0 * M * N * K
    for name, fn in [
        ("store", lambda: ops.gemm_nt(x, w, b, variant=8)),
        ("gelu+stash", lambda: ops.gemm_nt(x, w, b, epilogue=_lib.EPI_GELU, want_out2=True, variant=8)),
        ("resid+gate", lambda: ops.gemm_nt(x, w, b, epilogue=_lib.EPI_RESID, resid=resid, gate=gate, rows_per_batch=M // 2, variant=8)),
        ("dgelu", lambda: ops.gemm_nt(x, w, None, epilogue=_lib.EPI_DGELU, aux=z, variant=8)),
    ]:
        us = timeit(fn)
        print(f"M{M} N{N} K{K} {name:11s}: {us:7.1f} us  {fl/us/1e6:7.1f} TF/s", flush=True)
