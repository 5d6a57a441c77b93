"""CogVideoX (BASELINE config 3) -- what exists of it on the GPU, against oracle/cogvideox.py:
spec-level DDIM noising / velocity / weighted loss kernels (bit-exact: every op of the reference is a bf16 torch op), and the joint
text + video attention of a CogVideoX-2b block (226 + 17 550 tokens) through the mi355x provider.  pytest -m gpu."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


def _dev():
    return torch.device("cuda", 0)


def test_ddim_noise_velocity_loss_match_oracle():
    from finetrainers_amd.cogvideox import CogVideoXDDIMTables, MI355XCogVideoXSpecOps
    from oracle import cogvideox as cvx

    dev = _dev()
    g = torch.Generator().manual_seed(2)
    B, F_, C, H, W = 2, 5, 16, 12, 18
    lat = torch.randn(B, F_, C, H, W, generator=g).to(bf16)
    noise = torch.randn(B, F_, C, H, W, generator=g).to(bf16)
    vel = torch.randn(B, F_, C, H, W, generator=g).to(bf16)
    sig = torch.tensor([0.031, 0.874])
    osch = cvx.CogVideoXDDIMScheduler()
    sch = CogVideoXDDIMTables()
    assert torch.equal(sch.alphas_cumprod, osch.alphas_cumprod)
    spec = MI355XCogVideoXSpecOps()
    # oracle: the reference's own op sequence (base_specification.py:283-293, 326-329)
    x0_ref = lat * 1.15258426
    ts = (sig.flatten() * 1000.0).long()
    noisy_ref = osch.add_noise(x0_ref, noise, ts)
    pred_ref = osch.get_velocity(vel, noisy_ref, ts)
    noisy, x0, ts_g = spec.noise_and_target(lat.to(dev), sig.to(dev), noise=noise.to(dev))
    assert torch.equal(ts_g.cpu(), ts)
    assert torch.equal(x0.cpu(), x0_ref) and torch.equal(noisy.cpu(), noisy_ref.to(bf16))
    pred, target, _ = spec.forward(lambda **kw: (vel.to(dev),), lat.to(dev), None, sig.to(dev), noise=noise.to(dev))
    assert torch.equal(pred.cpu(), pred_ref.to(bf16)) and torch.equal(target.cpu(), x0_ref)
    loss_ref = cvx.sft_loss(pred_ref, x0_ref, sig, osch)
    loss = spec.loss(pred, target, sig.to(dev))
    assert abs(loss.item() - loss_ref.item()) <= 2e-6 * abs(loss_ref.item()), (loss.item(), loss_ref.item())


@pytest.mark.parametrize("heads_checked", [2])
def test_joint_attention_at_cogvideox_scale(heads_checked):
    """One CogVideoX-2b block's attention: 30 heads x 64, 226 text + 13 x 30 x 45 video tokens = 17 776 (not a multiple of the 64-key
    tile).  Forward + backward of all heads on the GPU; a subset of heads against torch's SDPA on the CPU (what the oracle calls)."""
    from finetrainers_amd import ops
    from oracle import ltx

    dev = _dev()
    B, H, S = 1, 30, 226 + 13 * 30 * 45
    assert S == 17776
    g = torch.Generator(device=dev).manual_seed(0)
    qkv = torch.randn((B, S, 3, H, 64), generator=g, device=dev).to(bf16)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    dout = torch.randn((B, S, H, 64), generator=g, device=dev).to(bf16).permute(0, 2, 1, 3)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    out, lse = ops.attn_fwd(q, k, v)  # warm-up
    dq, dk, dv = ops.attn_bwd(q, k, v, out, lse, dout)
    e0.record()
    out, lse = ops.attn_fwd(q, k, v)
    e1.record()
    dq, dk, dv = ops.attn_bwd(q, k, v, out, lse, dout)
    e2.record()
    torch.cuda.synchronize()
    fl = 4.0 * B * H * S * S * 64
    t_f, t_b = e0.elapsed_time(e1) * 1e-3, e1.elapsed_time(e2) * 1e-3
    print(f"[cogvideox-attn] S={S} H={H}: fwd {t_f * 1e3:.2f} ms = {fl / t_f / 1e12:.0f} TF/s, bwd {t_b * 1e3:.2f} ms = {2.5 * fl / t_b / 1e12:.0f} TF/s (algorithmic)")
    assert torch.isfinite(out.float()).all() and torch.isfinite(dq.float()).all()
    hs = [0, H - 1][:heads_checked]
    qc, kc, vc = (t[:, hs].float().cpu().to(bf16).requires_grad_() for t in (q, k, v))
    o_ref = ltx.native_sdpa(qc, kc, vc, None)
    o_ref.backward(dout[:, hs].cpu())
    rel = lambda a, b: ((a.float().cpu() - b.float()).norm() / b.float().norm()).item()
    errs = {"o": rel(out[:, hs], o_ref.detach()), "dq": rel(dq[:, hs], qc.grad), "dk": rel(dk[:, hs], kc.grad), "dv": rel(dv[:, hs], vc.grad)}
    print("[cogvideox-attn] vs torch SDPA (CPU flash kernel):", {k_: f"{v_:.2e}" for k_, v_ in errs.items()})
    assert errs["o"] < 6e-3 and max(errs["dq"], errs["dk"], errs["dv"]) < 1.2e-2
    lse_ref = torch.logsumexp((qc.detach().float() @ kc.detach().float().transpose(-1, -2)) / 8.0, dim=-1)
    assert rel(lse[:, hs] * math.log(2.0), lse_ref) < 1e-4
