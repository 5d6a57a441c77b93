"""Wan-T2V full fine-tune path (BASELINE config 4) on the GPU against the CPU oracle (oracle/wan.py): the row-wise kernels one by one
against the oracle's modules, then a whole block -- forward, input gradients and EVERY parameter gradient."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


def _dev():
    return torch.device("cuda", 0)


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _rope_tables(S, hd, seed=0):
    """Random unit complex numbers per (position, pair): (cos, sin) fp32 [S, hd / 2] for the kernel, complex128 [1, 1, S, hd / 2] for the oracle."""
    g = torch.Generator().manual_seed(seed)
    ang = torch.rand(S, hd // 2, generator=g, dtype=torch.float64) * 6.283
    return (torch.cos(ang).float(), torch.sin(ang).float()), torch.polar(torch.ones_like(ang), ang).view(1, 1, S, hd // 2)


def test_wan_rowwise_kernels_match_the_oracle_modules():
    from finetrainers_amd import ops
    from oracle import ltx, wan

    dev = _dev()
    g = torch.Generator().manual_seed(0)
    B, S, D, hd = 2, 75, 1536, 128  # 75 rows per sample: the 32-row strips of the reducing kernels end ragged
    M = B * S
    x = torch.randn(B, S, D, generator=g).to(bf16)
    dy = torch.randn(B, S, D, generator=g).to(bf16)
    res = torch.randn(B, S, D, generator=g).to(bf16)
    mod = (0.3 * torch.randn(B, 6, D, generator=g)).float()
    w = (1 + 0.1 * torch.randn(D, generator=g)).to(bf16)
    b = (0.1 * torch.randn(D, generator=g)).to(bf16)
    xg, dyg, resg, modg, wg, bg = (t.to(dev) for t in (x, dy, res, mod, w, b))

    # FP32LayerNorm + modulation (norm1 / norm3 of the block), forward and backward with the shift / scale gradients
    xr, mr = x.clone().requires_grad_(True), mod.clone().requires_grad_(True)
    ln = wan.FP32LayerNorm(D, 1e-6, elementwise_affine=False)
    y_ref = (ln(xr.float()) * (1 + mr[:, 1:2]) + mr[:, 0:1]).type_as(xr)
    y_ref.backward(dy)
    y = ops.wan_ln(xg.view(M, D), S, shift=modg[:, 0], scale=modg[:, 1])
    assert _rel(y.view(B, S, D), y_ref.detach()) < 2e-3 and (y.view(B, S, D).cpu().float() - y_ref.detach().float()).abs().max() < 0.05
    red = torch.zeros(2, B, D, device=dev)
    dx = ops.wan_ln_bwd(xg.view(M, D), dyg.view(M, D), S, scale=modg[:, 1], dres=resg.view(M, D), red1=red[0], red2=red[1], red_per_batch=True)
    dx_ref = res + xr.grad  # bf16 accumulation of the two branches
    assert _rel(dx.view(B, S, D), dx_ref) < 3e-3
    assert _rel(red[0], mr.grad[:, 0]) < 1e-5 and _rel(red[1], mr.grad[:, 1]) < 1e-4

    # FP32LayerNorm with affine parameters (norm2): weight / bias gradients in one row of sums
    xr = x.clone().requires_grad_(True)
    ln2 = wan.FP32LayerNorm(D, 1e-6, elementwise_affine=True).to(bf16)
    with torch.no_grad():
        ln2.weight.copy_(w)
        ln2.bias.copy_(b)
    y_ref = ln2(xr.float()).type_as(xr)
    y_ref.backward(dy)
    y = ops.wan_ln(xg.view(M, D), S, w=wg, b=bg)
    assert _rel(y.view(B, S, D), y_ref.detach()) < 2e-3
    gw, gb = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    dx = ops.wan_ln_bwd(xg.view(M, D), dyg.view(M, D), S, w=wg, red1=gb, red2=gw)
    assert _rel(dx.view(B, S, D), xr.grad) < 3e-3
    assert _rel(gw, ln2.weight.grad) < 3e-3 and _rel(gb, ln2.bias.grad) < 3e-3  # the reference's gradients are bf16 roundings of the same sums

    # RMSNorm across heads + rotary embedding, on a strided view (the q third of a fused q|k|v buffer), with the weight gradient
    qkv = torch.randn(B, S, 3 * D, generator=g).to(bf16)
    (cos, sin), freqs = _rope_tables(S, hd)
    qr = qkv[..., D:2 * D].clone().requires_grad_(True)
    rms = ltx.RMSNorm(D, 1e-6, elementwise_affine=True).to(bf16)
    with torch.no_grad():
        rms.weight.copy_(w)
    split = lambda t: t.unflatten(2, (D // hd, -1)).transpose(1, 2)
    y_ref = wan.apply_rotary_emb(split(rms(qr)), freqs)  # [B, H, S, hd]
    dyh = split(dy)
    y_ref.backward(dyh)
    qg = qkv.to(dev).view(M, 3 * D)
    rope = (cos.to(dev), sin.to(dev))
    y = ops.wan_rms_rope(qg[:, D:2 * D], wg, S, rope=rope, head_dim=hd)
    assert _rel(y.view(B, S, D), y_ref.detach().transpose(1, 2).flatten(2)) < 2e-3
    gw = torch.zeros(D, device=dev)
    out = torch.zeros(M, 3 * D, dtype=bf16, device=dev)
    ops.wan_rms_rope_bwd(qg[:, D:2 * D], wg, dyg.view(M, D), S, rope=rope, head_dim=hd, dweight=gw, out=out[:, D:2 * D])
    assert _rel(out[:, D:2 * D].view(B, S, D), qr.grad) < 4e-3 and float(out[:, :D].abs().max()) == 0.0
    assert _rel(gw, rms.weight.grad) < 3e-3
    y0 = ops.wan_rms_rope(qg[:, D:2 * D], wg, S)  # no rotary embedding (cross-attention)
    assert _rel(y0.view(B, S, D), rms(qr).detach()) < 2e-3

    # gated residual with an fp32 gate, and the plain bf16 residual
    xr, yr, gr = x.clone().requires_grad_(True), dy.clone().requires_grad_(True), mod[:, 2:3].clone().requires_grad_(True)
    o_ref = (xr.float() + yr * gr).type_as(xr)
    o_ref.backward(res)
    o = ops.wan_gate_res(xg.view(M, D), dyg.view(M, D), S, gate=modg[:, 2])
    assert torch.equal(o.view(B, S, D).cpu(), o_ref.detach())
    dgate = torch.zeros(B, D, device=dev)
    dyy = ops.wan_gate_res_bwd(resg.view(M, D), dyg.view(M, D), modg[:, 2], S, dgate=dgate)
    assert torch.equal(dyy.view(B, S, D).cpu(), yr.grad) and _rel(dgate, gr.grad[:, 0]) < 1e-5
    assert torch.equal(ops.wan_gate_res(xg.view(M, D), dyg.view(M, D), S).view(B, S, D).cpu(), x + dy)

    # column sums (bias gradients), also wider than one 4096-column slab
    wide = torch.randn(M, 8960, generator=g).to(bf16)
    cs = torch.zeros(8960, device=dev)
    ops.wan_colsum(wide.to(dev), cs)
    ops.wan_colsum(wide.to(dev), cs)  # += semantics
    assert _rel(cs, 2 * wide.float().sum(0)) < 1e-5
    with pytest.raises(ValueError):
        ops.wan_ln(xg.view(M, D)[:, :100], S)


def test_adamw_bf16_matches_torch_adamw_on_bf16_parameters():
    """The optimiser of the bf16 full fine-tune: torch.optim.AdamW on bf16 parameters (bf16 moments) is a chain of bf16-rounded ops; the kernel follows
    it op by op.  Reference run on the CPU with the gradient the reference would see (bf16), 3 steps, with and without the global-norm clip."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(0)
    n = 20000
    p0 = torch.randn(n, generator=g).to(bf16)
    grads = [(torch.randn(n, generator=g) * s).float() for s in (0.5, 1e-3, 2.0)]
    for clip in (False, True):
        p_ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([p_ref], lr=1e-2, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-2)
        p, m, v = p0.clone().to(dev), torch.zeros(n, dtype=bf16, device=dev), torch.zeros(n, dtype=bf16, device=dev)
        scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, device=dev)
        for step, gr in enumerate(grads, 1):
            p_ref.grad = gr.to(bf16)
            if clip:
                torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
            opt.step()
            gd = gr.to(dev)
            ss = ops.grad_sumsq(gd.to(bf16).float(), scratch) if clip else None
            gn = torch.zeros(1, device=dev)
            ops.adamw_bf16_step(p, gd, m, v, step, 1e-2, (0.9, 0.95), 1e-8, 1e-2, sumsq=ss, max_norm=1.0, grad_norm_out=gn if clip else None)
            if clip:
                assert abs(gn.item() - gr.to(bf16).float().norm().item()) < 1e-3 * gn.item()
        diff = (p.cpu().float() - p_ref.detach().float()).abs()
        ulp = p_ref.detach().float().abs().clamp_min(1e-3) * 2.0 ** -7
        frac_exact = (p.cpu() == p_ref.detach()).float().mean().item()
        print(f"[adamw-bf16 clip={clip}] bit-identical parameters {frac_exact:.4f}, max diff / ulp {(diff / ulp).max().item():.2f}")
        # with the clip the reference's norm and coefficient are bf16 tensors, ours fp32: the scaled gradients differ in the last bit now and then
        assert frac_exact > (0.5 if clip else 0.98) and (diff <= 2 * ulp + 1e-2 * 2.0 ** -6).all()  # a last-bit difference of an update of size lr
        st = opt.state[p_ref]
        if not clip:
            assert (m.cpu() == st["exp_avg"]).float().mean() > 0.98 and (v.cpu() == st["exp_avg_sq"]).float().mean() > 0.98


def _wan_block_pair(D=256, heads=2, ffn=512, seed=0):
    from finetrainers_amd.wan import MI355XWanBlock
    from oracle import wan

    cfg = wan.WanConfig(num_attention_heads=heads, attention_head_dim=D // heads, ffn_dim=ffn, num_layers=1, text_dim=64)
    torch.manual_seed(seed)
    oblk = wan.WanTransformerBlock(cfg)
    with torch.no_grad():
        g = torch.Generator().manual_seed(7)
        for n, p in oblk.named_parameters():
            if "norm" in n and n.endswith("weight"):
                p.copy_(1 + 0.1 * torch.randn(p.shape, generator=g))
            elif n.endswith("bias"):
                p.copy_(0.05 * torch.randn(p.shape, generator=g))
    oblk = oblk.to(bf16)
    sd = {k.replace("ffn.proj_in.", "ffn.net.0.proj.").replace("ffn.proj_out.", "ffn.net.2."): v for k, v in oblk.state_dict().items()}
    gblk = MI355XWanBlock(dim=D, heads=heads, ffn_dim=ffn, eps=cfg.eps, device=_dev())
    gblk.load_diffusers_state_dict(sd)
    return cfg, oblk, gblk


@pytest.mark.parametrize("B,S,T", [(2, 48, 16), (1, 200, 64)])
def test_wan_block_full_finetune_parity(B, S, T):
    """One Wan block (heads of 128 like Wan2.1), forward + backward: output, gradients of the video tokens, the text tokens and the time projection,
    and the gradient of EVERY parameter (26 tensors + the modulation table) against the bf16 CPU oracle.  The yardstick printed next to each error is
    the bf16 oracle's own distance from the same block evaluated in fp32."""
    from oracle import ltx, wan

    cfg, oblk, gblk = _wan_block_pair()
    dev, D, hd = _dev(), 256, 128
    g = torch.Generator().manual_seed(B * 1000 + S)
    x = torch.randn(B, S, D, generator=g).to(bf16)
    enc = torch.randn(B, T, D, generator=g).to(bf16)
    temb = (0.5 * torch.randn(B, 6, D, generator=g)).to(bf16)
    dout = torch.randn(B, S, D, generator=g).to(bf16)
    (cos, sin), freqs = _rope_tables(S, hd, seed=3)

    def run(blk, cast):
        for p in blk.parameters():
            p.grad = None
        xr, er, tr = (t.to(cast).clone().requires_grad_(True) for t in (x, enc, temb))
        out = blk(xr, er, tr, freqs)
        out.backward(dout.to(cast))
        grads = {n.replace("ffn.proj_in.", "ffn.net.0.proj.").replace("ffn.proj_out.", "ffn.net.2."): p.grad.detach().clone() for n, p in blk.named_parameters()}
        return out.detach(), xr.grad, er.grad, tr.grad, grads

    o_ref, dx_ref, de_ref, dt_ref, g_ref = run(oblk, bf16)
    import copy

    o32, dx32, de32, dt32, g32 = run(copy.deepcopy(oblk).float(), torch.float32)
    floor, floor_worst = ltx.grads_rel_l2(g_ref, g32)

    xg, eg, tg = (t.to(dev).requires_grad_(True) for t in (x, enc, temb))
    gblk.zero_grad_flat()
    out = gblk(xg, eg, tg, (cos.to(dev), sin.to(dev)))
    out.backward(dout.to(dev))
    torch.cuda.synchronize()
    got = {k: v.cpu() for k, v in gblk.named_grads().items()}
    assert set(got) == set(g_ref)
    glob, worst = ltx.grads_rel_l2({k: v.reshape(g_ref[k].shape) for k, v in got.items()}, g_ref)
    glob32, worst32 = ltx.grads_rel_l2({k: v.reshape(g32[k].shape) for k, v in got.items()}, g32)
    e_o, e_dx, e_de, e_dt = _rel(out, o_ref), _rel(xg.grad, dx_ref), _rel(eg.grad, de_ref), _rel(tg.grad, dt_ref)
    print(f"[wan-block B={B} S={S} T={T}] out {e_o:.2e} (oracle bf16 vs fp32 {_rel(o_ref, o32):.2e}) | dx {e_dx:.2e} ({_rel(dx_ref, dx32):.2e}) "
          f"d text {e_de:.2e} ({_rel(de_ref, de32):.2e}) d temb {e_dt:.2e} ({_rel(dt_ref, dt32):.2e}) | parameter grads vs bf16 oracle {glob:.2e} (worst {worst:.2e}), "
          f"vs fp32 oracle {glob32:.2e} (worst {worst32:.2e}); bf16 oracle vs fp32 oracle {floor:.2e} (worst {floor_worst:.2e})")
    assert e_o < 5e-3 and e_dx < 1e-2 and e_de < 1e-2 and e_dt < 1e-2
    assert glob < 2.0 * floor + 2e-3 and worst < 2.0 * floor_worst + 5e-3  # both sides carry bf16 noise of the size of the floor
    # against the fp32 evaluation of the same block the kernels must not be further away than the reference's own bf16 path (x 1.5)
    assert glob32 < 1.5 * floor + 1e-3 and worst32 < 1.5 * floor_worst + 2e-3

    # a second backward without clearing: the fp32 gradient buffer accumulates
    xg.grad = eg.grad = tg.grad = None
    gblk(xg, eg, tg, (cos.to(dev), sin.to(dev))).backward(dout.to(dev))
    torch.cuda.synchronize()
    twice = gblk.grad_flat.cpu()
    once = torch.cat([got[n].flatten() for n, _ in gblk.layout.entries])
    assert _rel(twice, 2 * once) < 1e-4
