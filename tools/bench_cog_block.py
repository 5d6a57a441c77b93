"""One CogVideoX-2b block, forward + backward, at BASELINE config 3's token count (49 x 480 x 720 clip: 226 text + 13 x 30 x 45 = 17 550 video
tokens, width 1920 = 30 heads x 64, LoRA r = 64 on to_q / to_k / to_v / to_out.0), random-init weights, batch 1 per GPU.
    python tools/bench_cog_block.py [iters]
Prints ms per block (forward, forward + backward), the algorithmic TFLOP/s, and the per-kernel-class table of the in-stream HIP-event profiler."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import _lib  # noqa: E402
from finetrainers_amd.cogvideox import MI355XCogVideoXBlock  # noqa: E402

dev = torch.device("cuda", 0)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
B, T, S, D, H = 1, 226, 13 * 30 * 45, 1920, 30
N = T + S
bf16 = torch.bfloat16
g = torch.Generator(device=dev).manual_seed(0)
blk = MI355XCogVideoXBlock(dim=D, heads=H, device=dev)
sd = {}
for k, name in blk._KEYS.items():
    shp = getattr(blk, name).shape
    if len(shp) == 2:
        sd[k] = (torch.randn(shp, generator=g, device=dev) / shp[1] ** 0.5).to(bf16)
    elif "norm" in k and k.endswith("weight"):
        sd[k] = torch.ones(shp, device=dev, dtype=bf16)
    else:
        sd[k] = (0.02 * torch.randn(shp, generator=g, device=dev)).to(bf16)
blk.load_diffusers_state_dict(sd)
blk.add_adapter(r=64, lora_alpha=64.0)
with torch.no_grad():
    blk.lora_B.normal_(0, 0.02)
tokens = torch.randn((B, N, D), generator=g, device=dev).to(bf16).requires_grad_(True)
temb = torch.randn((B, 512), generator=g, device=dev).to(bf16)
dout = torch.randn((B, N, D), generator=g, device=dev).to(bf16)

lin_f = 2.0 * B * N * D * D * (4 + 8)  # q, k, v, out + feed-forward (2 x 4 D^2)
att_f = 4.0 * B * N * N * D
fwd_flop, bwd_flop = lin_f + att_f, lin_f + 2.5 * att_f  # frozen base weights: dgrads only; attention backward = 2.5 x forward (algorithmic)


def timed(fn, n):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def fwd():
    with torch.no_grad():
        return blk(tokens, temb, T)


def fwd_bwd():
    blk.lora_A.grad = blk.lora_B.grad = tokens.grad = None
    blk(tokens, temb, T).backward(dout)


t_f = timed(fwd, iters)
lib = _lib.load()
lib.ftmi_prof_enable(1)
t_fb = timed(fwd_bwd, iters)
lib.ftmi_prof_enable(0)
print(f"CogVideoX-2b block, B={B}, {T}+{S} tokens: forward {t_f:.2f} ms ({fwd_flop / t_f / 1e9:.0f} TF/s), forward+backward {t_fb:.2f} ms "
      f"({(fwd_flop + bwd_flop) / t_fb / 1e9:.0f} TF/s algorithmic = {(fwd_flop + bwd_flop) / t_fb / 1e9 / 2500:.3f} of the dense bf16 peak); "
      f"x30 blocks = {30 * t_fb:.0f} ms of block time per sample-step")
for k, name in {0: "gemm_nt", 1: "gemm_tn", 2: "attn_fwd", 3: "attn_bwd", 4: "gemm_nt_skinny"}.items():
    tms, n, fl, an, afl = ctypes.c_double(0), ctypes.c_long(0), ctypes.c_double(0), ctypes.c_long(0), ctypes.c_double(0)
    lib.ftmi_prof_summary(k, ctypes.byref(tms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(an), ctypes.byref(afl), 1)
    if n.value:
        per = tms.value / (iters + 2)
        print(f"  {name:15s} {n.value / (iters + 2):6.1f} launches/iter  {per:8.3f} ms/iter  avg {tms.value / n.value * 1e3:8.1f} us  {fl.value / (tms.value * 1e-3) / 1e12:7.1f} TF/s")
