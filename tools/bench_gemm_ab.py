"""Interleaved A/B of NT GEMM variants on the step's shapes (M = 5376), weights streamed from HBM (rotating copies) like the step does:
5 rounds x 40 launches per variant in one process, median and best, bit-equality against the first variant, hipBLASLt (through torch) as
calibration.  LORA=1 times ftmi_linear_lora_fwd (skinny down-projection + GEMM with the fused K-extension), EPI=gelu|resid the fused
epilogues.  usage: bench_gemm_ab.py 61,70,72   [SHAPES=MxNxK,...]"""
import math, os, sys, statistics, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops, _lib
dev = torch.device("cuda", 0)
shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")] if os.environ.get("SHAPES") else [
    (5376, 2048, 2048), (5376, 6144, 2048), (5376, 8192, 2048), (5376, 2048, 8192)]
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "61,70".split(","))]
lora = os.environ.get("LORA", "0") == "1"
epi = os.environ.get("EPI", "store")
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    ncopy = max(1, int(6e8 // (N * K * 2)))
    ws = [w] + [w.clone() for _ in range(ncopy - 1)]
    b = torch.randn((N,), device=dev, generator=g).to(torch.bfloat16)
    A = torch.randn(64, K, device=dev, generator=g) / math.sqrt(K)
    Bm = torch.randn(N, 64, device=dev, generator=g) * 0.05
    resid = torch.randn((M, N), device=dev, generator=g).to(torch.bfloat16) if epi == "resid" else None
    res = {v: [] for v in variants}
    it = [0]
    def run(v, fixed=False):
        if not fixed:
            it[0] = (it[0] + 1) % ncopy
        wv = ws[0] if fixed else ws[it[0]]
        if lora:
            return ops.linear_lora_fwd(x, wv, b, A, Bm, 0.5, variant=v)[0]
        if epi == "gelu":
            return ops.gemm_nt(x, wv, b, epilogue=_lib.EPI_GELU, want_out2=True, variant=v)[0]
        if epi == "resid":
            return ops.gemm_nt(x, wv, b, epilogue=_lib.EPI_RESID, resid=resid, variant=v)
        return ops.gemm_nt(x, wv, b, variant=v)
    ref = run(variants[0], fixed=True).clone()
    same = {v: bool(torch.equal(run(v, fixed=True), ref)) for v in variants}
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
    tag = "lora " if lora else (epi + " " if epi != "store" else "")
    for v in variants:
        med, best = statistics.median(res[v]), min(res[v])
        print(f"M{M} N{N} K{K} {tag}variant {v:3d}: median {med*1e3:7.1f} us = {2*M*N*K/med/1e9:7.1f} TF/s   best {best*1e3:7.1f} us = {2*M*N*K/best/1e9:7.1f} TF/s   same_bits_as_v{variants[0]}={same[v]}", flush=True)
    if not lora and epi == "store" and os.environ.get("VENDOR", "1") == "1":
        for _ in range(30): torch.nn.functional.linear(x, ws[0], b)
        ts = []
        for rnd in range(3):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for i in range(20): torch.nn.functional.linear(x, ws[i % ncopy], b)
            e.record(); torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / 20)
        med = statistics.median(ts)
        print(f"M{M} N{N} K{K} hipBLASLt (calibration): median {med*1e3:7.1f} us = {2*M*N*K/med/1e9:7.1f} TF/s", flush=True)
