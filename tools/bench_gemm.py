"""Micro-benchmark of the NT GEMM variants on the shapes of the LTX step (M = 5376)."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops, _lib
dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else [(5376, 2048, 2048), (5376, 6144, 2048), (5376, 8192, 2048), (5376, 2048, 8192), (5376, 2048, 6144)]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3".split(","))]
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    # COLD=1: rotate through enough copies of W (> 600 MB) that every launch streams its weights from HBM like the real step does
    ncopy = max(1, int(6e8 // (N * K * 2))) if os.environ.get("COLD") == "1" else 1
    ws = [w] + [w.clone() for _ in range(ncopy - 1)]
    it = [0]
    def nextw():
        it[0] = (it[0] + 1) % ncopy
        return ws[it[0]]
    b = torch.randn((N,), device=dev, generator=g).to(torch.bfloat16)
    ref = None
    for v in variants:
        out = ops.gemm_nt(x, w, b, variant=v)
        if ref is None:
            ref = out
        ok = torch.equal(out, ref)
        ms = 1e9
        for rep in range(3):  # clocks ramp under load: long warm-up, best of 3 passes
            for _ in range(30):
                ops.gemm_nt(x, nextw(), b, variant=v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            n = 50
            for _ in range(n):
                ops.gemm_nt(x, nextw(), b, variant=v)
            e.record(); torch.cuda.synchronize()
            ms = min(ms, s.elapsed_time(e) / n)
        print(f"M{M} N{N} K{K} variant {v}: {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s  same_as_v{variants[0]}={ok}", flush=True)
    # calibration only: the vendor library GEMM (hipBLASLt through torch) on the same shape -- not used by the product path
    for _ in range(50):
        torch.nn.functional.linear(x, nextw(), b)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        torch.nn.functional.linear(x, nextw(), b)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"M{M} N{N} K{K} hipBLASLt  : {ms*1e3:8.1f} us  {2*M*N*K/ms/1e9:7.1f} TF/s", flush=True)
