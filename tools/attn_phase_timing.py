"""Where a wave of the attention forward spends its cycles (experimental build: FTMI_EXPERIMENTAL=1 python -m finetrainers_amd.csrc.build).
Variant 19 of FTMI_ATTN_FWD sums s_memtime deltas (ticks: ~0.58 of a core cycle under this load) between four program points of the tile loop per wave and writes them over
lse2[row .. row+3] of the wave's first rows.  Prints per-tile averages over all waves of the cfg-2 self-attention launch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops  # noqa: E402

dev = torch.device("cuda", 0)
B, H, S = 2, 32, 2688
g = torch.Generator(device=dev).manual_seed(0)
q = torch.randn((B, S, H, 64), generator=g, device=dev).to(torch.bfloat16).permute(0, 2, 1, 3)
kv = torch.randn((B, S, 2, H, 64), generator=g, device=dev).to(torch.bfloat16)
k, v = kv[:, :, 0].permute(0, 2, 1, 3), kv[:, :, 1].permute(0, 2, 1, 3)
os.environ["FTMI_ATTN_FWD"] = "19"
for _ in range(3):
    out, lse = ops.attn_fwd(q, k, v, None)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    out, lse = ops.attn_fwd(q, k, v, None)
e1.record()
torch.cuda.synchronize()
raw = lse.reshape(B * H, S // 32, 32)[:, :, :8].double()
t = raw[:, :, :4]  # [bh, wave-tile, phase]
nt = S // 64
names = ["scores: K fragments + 8 MFMA issued", "softmax VALU (max, exp2, lazy check)", "pack + P.V / row-sum MFMA issued", "stage commit + DMA wait + barrier"]
tot = t.sum(-1)
print(f"instrumented launch {e0.elapsed_time(e1) / 10 * 1e3:.1f} us; per wave and tile (s_memtime ticks), {t.shape[0] * t.shape[1]} waves x {nt} tiles:")
for i, n in enumerate(names):
    x = t[:, :, i] / nt
    print(f"  {n:42s} mean {x.mean().item():7.1f}  min {x.min().item():7.1f}  max {x.max().item():7.1f}   {100 * (t[:, :, i].sum() / tot.sum()).item():5.1f} %")
print(f"  {'whole tile':42s} mean {(tot / nt).mean().item():7.1f}  min {(tot / nt).min().item():7.1f}  max {(tot / nt).max().item():7.1f}")
slots = raw[:, :, 4].flatten().long()
print("  waves per SIMD slot (HW_ID[3:0]):", torch.bincount(slots, minlength=4).tolist())
for sl in sorted(set(slots.tolist())):
    print(f"    slot {sl}: whole tile mean {(tot / nt).flatten()[slots == sl].mean().item():7.1f} ticks")
