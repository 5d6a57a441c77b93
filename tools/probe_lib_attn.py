"""Calibration only: the vendor-library attention (torch SDPA on ROCm) on the self-attention shape of the step."""
import torch, time
dev = torch.device("cuda", 0)
B, H, S, d = 2, 32, 2688, 64
q, k, v = [torch.randn((B, H, S, d), device=dev, dtype=torch.bfloat16, requires_grad=True) for _ in range(3)]
do = torch.randn((B, H, S, d), device=dev, dtype=torch.bfloat16)
def run(bwd):
    o = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    if bwd:
        o.backward(do)
for bwd in (False, True):
    for _ in range(5): run(bwd)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): run(bwd)
    e.record(); torch.cuda.synchronize()
    print(("fwd+bwd" if bwd else "fwd    "), f"{s.elapsed_time(e)/20*1e3:8.1f} us")
