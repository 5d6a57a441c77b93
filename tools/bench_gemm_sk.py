"""A/B of the persistent stream-K GEMM (variant 60) against the one-tile-per-workgroup kernels (61) on the step's shapes, weights streamed
from HBM (rotating copies) like the step does; interleaved rounds in one process, median and best."""
import math, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops, _lib
dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else [
    (5376, 2048, 2048), (5376, 6144, 2048), (5376, 8192, 2048), (5376, 2048, 8192), (5376, 2048, 6144)]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "61,60".split(","))]
lora = os.environ.get("LORA", "0") == "1"
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    ncopy = max(1, int(6e8 // (N * K * 2)))
    ws = [w] + [w.clone() for _ in range(ncopy - 1)]
    b = torch.randn((N,), device=dev, generator=g).to(torch.bfloat16)
    A = torch.randn(64, K, device=dev, generator=g) / math.sqrt(K)
    Bm = torch.randn(N, 64, device=dev, generator=g) * 0.05
    res = {v: [] for v in variants}
    it = [0]
    def run(v):
        it[0] = (it[0] + 1) % ncopy
        if lora:
            ops.linear_lora_fwd(x, ws[it[0]], b, A, Bm, 0.5, variant=v)
        else:
            ops.gemm_nt(x, ws[it[0]], b, variant=v)
    for v in variants:
        for _ in range(20): run(v)
    for rnd in range(5):
        for v in variants:
            for _ in range(10): run(v)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            n = 40
            for _ in range(n): run(v)
            e.record(); torch.cuda.synchronize()
            res[v].append(s.elapsed_time(e) / n)
    for v in variants:
        med, best = statistics.median(res[v]), min(res[v])
        extra = " (incl. the skinny down-projection launch)" if lora else ""
        print(f"M{M} N{N} K{K} {'lora ' if lora else ''}variant {v}: median {med*1e3:7.1f} us = {2*M*N*K/med/1e9:7.1f} TF/s   best {best*1e3:7.1f} us = {2*M*N*K/best/1e9:7.1f} TF/s{extra}", flush=True)
print("sk status", _lib.load().ftmi_gemm_sk_status() if hasattr(_lib.load(), "ftmi_gemm_sk_status") else "n/a (product build)")
