"""Timeline of one stream-K GEMM launch from the kernel's own clock stamps (FTMI_SK_TRACE=1): per workgroup the time of each K phase,
hand-off wait, partial add and epilogue, in microseconds at the shader clock the box reports (s_memtime ticks at 100 MHz on gfx950)."""
import ctypes, math, os, sys
os.environ["FTMI_SK_TRACE"] = "1"
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops, _lib
dev = torch.device("cuda", 0)
lib = _lib.load()
shapes = [(5376, 2048, 2048), (5376, 8192, 2048), (5376, 2048, 8192), (5376, 6144, 2048)]
lora = os.environ.get("LORA", "0") == "1"
g = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in shapes:
    x = torch.randn((M, K), device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev, generator=g) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn((N,), device=dev, generator=g).to(torch.bfloat16)
    A = torch.randn(64, K, device=dev, generator=g) / math.sqrt(K)
    Bm = torch.randn(N, 64, device=dev, generator=g) * 0.05
    def run():
        if lora: ops.linear_lora_fwd(x, w, b, A, Bm, 0.5, variant=60)
        else: ops.gemm_nt(x, w, b, variant=60)
    for _ in range(20): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); run(); e.record(); torch.cuda.synchronize()
    G = 256
    buf = (ctypes.c_ulonglong * (G * 16))()
    n = lib.ftmi_gemm_sk_trace(buf, G * 16)
    T = [[buf[v * 16 + i] for i in range(16)] for v in range(n)]
    t0 = min(t[0] for t in T if t[0])
    tend = max(max(t) for t in T)
    nk, nk2 = K // 64, (3 if lora else 0)
    wk = (ctypes.c_int * (G * 8))()
    lib.ftmi_gemm_sk_plan(((M + 255) // 256) * (N // 256), G, nk, nk2 + int(os.environ.get("FTMI_SK_EPI_COST", "4")), 4, int(os.environ.get("FTMI_SK_PCOST", "1")), int(os.environ.get("FTMI_SK_ACOST", "2")), wk)
    tick = 0.01  # us per s_memtime tick (100 MHz constant clock)
    print(f"== M{M} N{N} K{K} {'lora' if lora else ''}: event time {s.elapsed_time(e)*1e3:.1f} us; kernel span by stamps {(tend - t0) * tick:.1f} us; start skew {(max(t[0] for t in T) - t0) * tick:.1f} us")
    ends = sorted((max(t) - t0) * tick for t in T)
    print(f"   workgroup end times us: min {ends[0]:.1f} p25 {ends[len(ends)//4]:.1f} median {ends[len(ends)//2]:.1f} p75 {ends[3*len(ends)//4]:.1f} max {ends[-1]:.1f}")
    for v in (0, 1, 2, 3, 100, 101, 200, 255):
        t = [x_ for x_ in T[v] if x_]
        rel = [(a - T[v][0]) * tick for a in t]
        d = [f"{rel[i] - rel[i-1]:.1f}" for i in range(1, len(rel))]
        print(f"   wg {v:3d} work {list(wk[v*8:v*8+7])}: start +{(T[v][0]-t0)*tick:.1f}, deltas {d}, end {rel[-1]:.1f}")
