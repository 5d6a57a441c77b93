"""GPU parity tests of the individual gfx950 kernels, through the C ABI (ctypes), against plain PyTorch
fp32 references evaluated on the same bf16-rounded inputs.  Run on the MI355X box: pytest -m gpu."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16



@pytest.fixture
def sw():
    """Set an FTMI_* kernel switch for the rest of the test.  The library reads its switches from the environment ONCE (no getenv on the launch path);
    ftmi_reload_switches() makes it look again, here after every change and after the environment has been put back."""
    import os

    from finetrainers_amd import _lib

    saved = {}

    def set_(name, value):
        if name not in saved:
            saved[name] = os.environ.get(name)
        os.environ[name] = value
        _lib.load().ftmi_reload_switches()

    yield set_
    for k, v in saved.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v
    _lib.load().ftmi_reload_switches()

def _dev():
    return torch.device("cuda", 0)


def rel_l2(got, ref):
    got, ref = got.float().cpu(), ref.float().cpu()
    return ((got - ref).norm() / ref.norm().clamp_min(1e-30)).item()


def report(name, got, ref, rel_tol, max_ulp_frac=None):
    got, ref = got.float().cpu(), ref.float().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    assert torch.isfinite(got).all(), f"{name}: non-finite output"
    err = rel_l2(got, ref)
    mx = (got - ref).abs().max().item()
    mism = (got != ref).float().mean().item()
    print(f"[parity] {name:42s} rel_l2={err:.3e} max_abs={mx:.3e} mismatch_frac={mism:.4f}")
    assert err <= rel_tol, f"{name}: rel_l2 {err:.3e} > {rel_tol:.1e} (max_abs {mx:.3e})"
    return err


# the tiled-GEMM kernels the product library ships: 8 = automatic choice (production), 42 / 44 / 47 pin 192x128 / 128x128 / 256x256 tiles,
# 70 = the hand-placed 4-wave pipeline (128 x 128 per wave, accumulators in AGPRs), 80 / 86 = that pipeline on 16 x 16 x 32 MFMAs (256 x 256 / 192 x 256 tiles), 72 = the 32 x 32
# stream on 8 waves (the research variants of tools/experimental/gemm_experimental.hip.h are not in the product build)
SHIPPED_VARIANTS = [8, 42, 44, 47, 70, 72, 80, 86, 87, 2286, 1386, 1387, 1380]


def rnd(shape, gen, scale=1.0, dtype=bf16):
    return (torch.randn(shape, generator=gen) * scale).to(dtype)


# ----------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", SHIPPED_VARIANTS)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 128), (32, 2048, 2048), (1000, 192, 2048), (257, 64, 2048), (515, 6144, 2048), (2048, 6144, 256)])
def test_gemm_nt_store(M, N, K, variant):
    from finetrainers_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N + K)
    x, w, b = rnd((M, K), g), rnd((N, K), g, 1 / math.sqrt(K)), rnd((N,), g)
    ref = (x.float() @ w.float().t() * 0.5 + b.float()).to(bf16)
    out = ops.gemm_nt(x.to(_dev()), w.to(_dev()), b.to(_dev()), alpha=0.5, variant=variant)
    torch.cuda.synchronize()
    report(f"gemm_nt store {M}x{N}x{K} v{variant}", out, ref, 2e-3)
    # asymmetric check against transposition / tile misplacement: exact comparison of a large fraction
    assert (out.cpu() != ref).float().mean() < 0.02


@pytest.mark.parametrize("variant", SHIPPED_VARIANTS)
def test_gemm_nt_epilogues(variant):
    from finetrainers_amd import _lib, ops

    dev = _dev()
    g = torch.Generator().manual_seed(5)
    M, N, K, S = 300, 256, 512, 150
    x, w, b = rnd((M, K), g), rnd((N, K), g, 1 / math.sqrt(K)), rnd((N,), g)
    resid, gate, z = rnd((M, N), g), rnd((2, N), g), rnd((M, N), g)
    y = (x.float() @ w.float().t() + b.float())
    # gelu: out = gelu(bf(y)), out2 = bf(y)
    out, out2 = ops.gemm_nt(x.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_GELU, want_out2=True, variant=variant)
    zref = y.to(bf16)
    report("epi gelu pre-activation", out2, zref, 2e-3)
    report("epi gelu", out, torch.nn.functional.gelu(zref.float(), approximate="tanh").to(bf16), 3e-3)
    # resid + gate
    out = ops.gemm_nt(x.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_RESID, resid=resid.to(dev), gate=gate.to(dev), rows_per_batch=S,
                      variant=variant)
    gexp = gate.float().repeat_interleave(S, dim=0)
    ref = (resid.float() + (y.to(bf16).float() * gexp).to(bf16).float()).to(bf16)
    report("epi resid+gate", out, ref, 3e-3)
    out = ops.gemm_nt(x.to(dev), w.to(dev), b.to(dev), epilogue=_lib.EPI_RESID, resid=resid.to(dev), variant=variant)
    report("epi resid", out, (resid.float() + y.to(bf16).float()).to(bf16), 3e-3)
    # gelu'
    out = ops.gemm_nt(x.to(dev), w.to(dev), None, epilogue=_lib.EPI_DGELU, aux=z.to(dev), variant=variant)
    zz = z.float().clone().requires_grad_(True)
    torch.nn.functional.gelu(zz, approximate="tanh").backward((x.float() @ w.float().t()).to(bf16).float())
    report("epi dgelu", out, zz.grad.to(bf16), 3e-3)


def _split_ref(t):
    """What the kernels carry for an fp32 LoRA value: hi = bf16(t), lo = bf16(t - hi)."""
    hi = t.to(bf16)
    lo = (t - hi.float()).to(bf16)
    return hi, lo


@pytest.mark.parametrize("base,others", [(86, (2286, 1386)), (87, (1387,)), (80, (1380,))])
@pytest.mark.parametrize("M,N,K", [(5376, 2048, 2048), (1100, 512, 8192), (1024, 256, 64), (1024, 256, 128), (1024, 256, 192), (1024, 256, 256), (1024, 256, 320)])
def test_nt16_k_loops_are_bit_identical(M, N, K, base, others):
    """The K loops of gemm_nt16_kernel (round 6: register-staged prefetch 22xx, W on a three-slot direct-to-LDS ring 13xx) issue the same MFMAs on the
    same fragments in the same order as the two-slot loop: every epilogue and the LoRA K-extension must agree bit for bit -- including K of one, two and three
    stages, where the prologue's loads are all there is."""
    from finetrainers_amd import _lib, ops

    dev = _dev()
    g = torch.Generator().manual_seed(M + N + K)
    x, w, b = rnd((M, K), g).to(dev), rnd((N, K), g, 1 / math.sqrt(K)).to(dev), rnd((N,), g).to(dev)
    resid, z = rnd((M, N), g).to(dev), rnd((M, N), g).to(dev)
    A = (torch.randn(64, K, generator=g) / math.sqrt(K)).to(dev)
    Bm = (torch.randn(N, 64, generator=g) * 0.05).to(dev)

    def run(v):
        outs = [ops.gemm_nt(x, w, b, variant=v), ops.gemm_nt(x, w, b, epilogue=_lib.EPI_RESID, resid=resid, variant=v),
                ops.gemm_nt(x, w, b, epilogue=_lib.EPI_DGELU, aux=z, variant=v)]
        y, pre = ops.gemm_nt(x, w, b, epilogue=_lib.EPI_GELU, want_out2=True, variant=v)
        outs += [y, pre]
        if K >= 256:  # (the down-projection kernel's minimum)
            outs.append(ops.linear_lora_fwd(x, w, b, A, Bm, 0.5, variant=v)[0])
        return outs

    ref = run(base)
    for v in others:
        for i, (a, r) in enumerate(zip(run(v), ref)):
            assert torch.equal(a, r), f"variant {v} differs from {base} in output {i}"


@pytest.mark.parametrize("variant", SHIPPED_VARIANTS)
@pytest.mark.parametrize("M", [32, 300, 5376])
def test_linear_lora_fwd(M, variant):
    """peft lora.Linear semantics with fp32 adapters (trainer.py:132-136): bf16(base) + scale * (x A^T) B^T evaluated in fp32,
    re-rounded.  The LoRA branch must be fp32-equivalent: checked against an fp64 evaluation, far below bf16 resolution."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(17)
    K, N, r, s = 2048, 2048, 64, 0.5
    x, w, b = rnd((M, K), g), rnd((N, K), g, 1 / math.sqrt(K)), rnd((N,), g)
    A = torch.randn(r, K, generator=g) / math.sqrt(K)          # fp32 adapter weights, NOT bf16-representable
    Bm = torch.randn(N, r, generator=g) * 0.05
    base = (x.float() @ w.float().t() + b.float()).to(bf16)
    xa64 = x.double() @ A.double().t() * s
    ref = (base.double() + xa64 @ Bm.double().t()).float().to(bf16)
    y, xa = ops.linear_lora_fwd(x.to(dev), w.to(dev), b.to(dev), A.to(dev), Bm.to(dev), s, variant=variant)
    report(f"linear+lora M={M} v{variant}", y, ref, 3e-3)
    xa = xa.float().cpu()
    hi, lo, hi2 = xa[:, :r], xa[:, r:2 * r], xa[:, 2 * r:]
    assert torch.equal(hi, hi2)
    err = ((hi.double() + lo.double()) - xa64).norm() / xa64.norm()
    print(f"[parity] lora xa (hi+lo) vs fp64   M={M}: rel_l2={err:.3e}")
    assert err < 2e-5, "x A^T must carry fp32-equivalent precision (bf16 operands would give ~3e-3)"
    # the down-projection may cut K across workgroups (gemm_skinny.hip: the last slice to arrive adds the partial tiles in slice order): the bits
    # must not depend on the arrival order
    for _ in range(3):
        y2, xa2 = ops.linear_lora_fwd(x.to(dev), w.to(dev), b.to(dev), A.to(dev), Bm.to(dev), s, variant=variant)
        assert torch.equal(xa2.cpu(), xa.to(bf16)) and torch.equal(y2, y)


@pytest.mark.parametrize("M", [96, 700, 5376])
def test_linear_lora_bwd(M):
    """Autograd of peft lora.Linear over a frozen Linear with fp32 adapters: dx (bf16), dA / dB (fp32-equivalent)."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(23)
    K, N, r, s = 2048, 2048, 64, 0.5
    x, w, b = rnd((M, K), g), rnd((N, K), g, 1 / math.sqrt(K)), rnd((N,), g)
    A = torch.randn(r, K, generator=g) / math.sqrt(K)
    Bm = torch.randn(N, r, generator=g) * 0.05
    dy = rnd((M, N), g)
    xr = x.double().requires_grad_(True)
    Ar, Br = A.double().requires_grad_(True), Bm.double().requires_grad_(True)
    y = xr @ w.double().t() + b.double() + (xr @ Ar.t()) @ Br.t() * s
    y.backward(dy.double())
    _, xa = ops.linear_lora_fwd(x.to(dev), w.to(dev), b.to(dev), A.to(dev), Bm.to(dev), s, variant=8)
    w_t = ops.transpose_bf16(w.to(dev))
    ga0 = torch.ones((r, K), dtype=torch.float32, device=dev)  # pre-existing .grad content must be accumulated into
    dx, ga, gb = ops.linear_lora_bwd(x.to(dev), dy.to(dev), xa, w_t, A.to(dev), Bm.to(dev), s, grad_a=ga0, variant=8)
    torch.cuda.synchronize()
    report(f"lora bwd dx M={M}", dx, xr.grad.float().to(bf16), 4e-3)
    report(f"lora bwd dA M={M}", ga - 1.0, Ar.grad.float(), 5e-5)   # fp32-equivalent (bf16 operands: ~4e-3)
    report(f"lora bwd dB M={M}", gb, Br.grad.float(), 5e-5)
    dx2, _, _ = ops.linear_lora_bwd(None, dy.to(dev), None, w_t, None, None, s, variant=8)  # r = 0: plain dgrad
    report("plain dgrad", dx2, (dy.float() @ w.float()).to(bf16), 3e-3)


@pytest.mark.parametrize("M,K,N,r", [(1024, 256, 256, 512), (2048, 512, 256, 512)])
def test_linear_lora_bwd_fold_on_the_long_operand(M, K, N, r):
    """Advisor finding of round 5: when the LoRA rank exceeds the Linear's width the (hi, lo) pair of the weight-gradient GEMM is the LONG operand (dB: P =
    out_features 256, Q = rank 512 with the planes of x A^T folded).  A 256-wide tile of a folded operand would need 216 KB of LDS (launch failure); such
    launches must stay on the 128-wide tile and give the fp32-equivalent gradients.  (Narrower Linears cannot carry a LoRA through this entry point at all:
    the fp32-equivalent down-projection needs >= 256 input AND output features.)"""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(29)
    s = 0.5
    x, w = rnd((M, K), g), rnd((N, K), g, 1 / math.sqrt(K))
    A = torch.randn(r, K, generator=g) / math.sqrt(K)
    Bm = torch.randn(N, r, generator=g) * 0.05
    dy = rnd((M, N), g)
    xr = x.double().requires_grad_(True)
    Ar, Br = A.double().requires_grad_(True), Bm.double().requires_grad_(True)
    y = xr @ w.double().t() + (xr @ Ar.t()) @ Br.t() * s
    y.backward(dy.double())
    _, xa = ops.linear_lora_fwd(x.to(dev), w.to(dev), None, A.to(dev), Bm.to(dev), s, variant=8)
    w_t = ops.transpose_bf16(w.to(dev))
    dx, ga, gb = ops.linear_lora_bwd(x.to(dev), dy.to(dev), xa, w_t, A.to(dev), Bm.to(dev), s, variant=8)
    torch.cuda.synchronize()
    report(f"lora bwd (rank {r}, {K}->{N}) dx", dx, xr.grad.float().to(bf16), 4e-3)
    report(f"lora bwd (rank {r}, {K}->{N}) dA", ga, Ar.grad.float(), 5e-5)
    report(f"lora bwd (rank {r}, {K}->{N}) dB", gb, Br.grad.float(), 5e-5)


@pytest.mark.parametrize("M,P,Q", [(64, 64, 64), (300, 128, 64), (1000, 64, 2048), (5376, 2048, 64), (333, 192, 2048), (256, 4096, 64), (1024, 64, 2048), (640, 192, 2048), (17776, 1920, 64), (1100, 64, 1920)])
def test_gemm_tn(M, P, Q):
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(M + P + Q)
    u, v = rnd((M, P), g), rnd((M, Q), g)
    ref = u.float().t() @ v.float() * 0.25
    out = ops.gemm_tn(u.to(dev), v.to(dev), scale=0.25)
    report(f"gemm_tn {M}x{P}x{Q}", out, ref, 1e-4)
    out2 = ops.gemm_tn(u.to(dev), v.to(dev), out=out, scale=0.25)  # accumulation
    report(f"gemm_tn accumulate {M}x{P}x{Q}", out2, 2 * ref, 1e-4)


# ----------------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, bias, dout=None):
    """fp32 math attention on bf16-rounded inputs (+ autograd)."""
    q, k, v = (t.float().clone().requires_grad_(dout is not None) for t in (q, k, v))
    s = q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1])
    if bias is not None:
        s = s + bias[:, None, None, :].float()
    o = torch.softmax(s, dim=-1) @ v
    if dout is None:
        return o
    o.backward(dout.float())
    return o.detach(), q.grad, k.grad, v.grad


ATTN_CASES = [
    # B, H, Sq, Sk, biased
    (2, 8, 256, 256, False),   # the reference's own test shape (tests/models/attention_dispatch.py:113-120)
    (1, 2, 32, 32, False),     # cfg 1 self-attention (one partial tile)
    (2, 3, 100, 77, True),     # ragged both ways
    (1, 4, 2688, 2688, False), # cfg 2 self-attention length
    (2, 4, 2688, 128, True),   # cfg 2 cross-attention with text mask (few keys: split-query dK/dV kernel)
    (1, 2, 600, 77, True),     # split-query dK/dV kernel, ragged queries (last 128-block holds 88) and keys
    (1, 1, 520, 33, False),    # split-query dK/dV kernel, second 64-row tile of the last block wholly past the end
    (2, 8, 1000, 128, False),  # few keys without a bias (two full key tiles), ragged last query block
    (1, 8, 640, 64, False),    # ... a single key tile
    (1, 3, 700, 128, True),    # ... with a mask, B * H = 3 (plain block order)
    (3, 5, 257, 65, False),    # one row / one key past a tile boundary; B * H = 15 is not a multiple of the 8 XCDs (plain block order)
    (1, 1, 1, 1, False),       # a single query and a single key
    (2, 2, 129, 191, True),    # ragged both ways with per-sample masks, keys one short of three tiles
    (1, 3, 64, 4097, False),   # one query tile against many key tiles plus one key
    # the hand-placed backward pipelines DIRECTLY against fp32 autograd (round-5 review): >= 256 key blocks x heads, no bias, >= 256 keys dispatch to
    # attn_bwd_dq_pl_kernel + attn_bwd_dkdv_pl_kernel (the cases above mostly stay below that threshold and reach them only through the bit-identity tests)
    (1, 16, 2688, 2688, False),  # cfg 2 self-attention length, 22 x 16 = 352 key blocks
    (2, 16, 1000, 1000, False),  # ragged queries and keys (the bounds-checked DMA zero-fills), 8 x 32 = 256 key blocks
]


@pytest.mark.parametrize("B,H,Sq,Sk,biased", ATTN_CASES)
def test_attention_fwd_bwd(B, H, Sq, Sk, biased):
    from finetrainers_amd import ops

    dev = _dev()
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(0)
    q, k, v = rnd((B, H, Sq, 64), g), rnd((B, H, Sk, 64), g), rnd((B, H, Sk, 64), g)
    dout = rnd((B, H, Sq, 64), g)
    bias = None
    if biased:
        mask = torch.zeros(B, Sk)
        for b in range(B):
            mask[b, : max(1, (Sk * (b + 1)) // (B + 1))] = 1
        bias = ((1 - mask.to(bf16)) * -10000.0).float()
    o_ref, dq_ref, dk_ref, dv_ref = _attn_ref(q, k, v, bias, dout)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    bd = None if bias is None else bias.to(dev)
    out, lse = ops.attn_fwd(qd, kd, vd, bd)
    tag = f"attn B{B} H{H} {Sq}x{Sk}{' bias' if biased else ''}"
    report(tag + " fwd", out, o_ref, 6e-3)
    assert (out.float().cpu() - o_ref).abs().max() < 5e-3 * max(1.0, o_ref.abs().max().item())  # reference's own atol (5e-3)
    lse_ref = torch.logsumexp((q.float() @ k.float().transpose(-1, -2)) / 8.0 + (0 if bias is None else bias[:, None, None, :]), dim=-1)
    report(tag + " lse", lse.cpu() * math.log(2.0), lse_ref, 1e-4)
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, out, lse, dout.to(dev), bd)
    report(tag + " dq", dq, dq_ref, 1e-2)
    report(tag + " dk", dk, dk_ref, 1e-2)
    report(tag + " dv", dv, dv_ref, 1e-2)


@pytest.mark.parametrize("B,H,Sq,Sk,biased", [(2, 32, 2688, 128, True), (2, 8, 1000, 128, False), (1, 2, 600, 77, True), (1, 8, 640, 64, False)])
def test_few_keys_resident_kernels_are_bit_identical_to_the_general_ones(B, H, Sq, Sk, biased, sw):
    """LTX cross-attention (128 text keys): the resident-K/V dQ kernel (K / V staged once, Q / dO / O through LDS with the row-contiguous DMA, a walk over
    several 128-row query blocks with counted waits that leave the output stores in flight) does the arithmetic of the general kernel statement for statement --
    dQ, and dK / dV through the delta it publishes, must be the same bits as with FTMI_ATTN_FEWKEYS=0 (the general 64-row dQ kernel).  In an FTMI_EXPERIMENTAL
    build FTMI_ATTN_FEWKEYS_FWD=1 adds the resident forward kernel (no faster than the general one: not shipped) to the comparison."""
    from finetrainers_amd import _lib, ops

    if hasattr(_lib.load(), "ftmi_gemm_sk_status"):  # (an entry point only the FTMI_EXPERIMENTAL build exports)
        sw("FTMI_ATTN_FEWKEYS_FWD", "1")
    dev = _dev()
    g = torch.Generator().manual_seed(7)
    q, k, v = rnd((B, H, Sq, 64), g).to(dev), rnd((B, H, Sk, 64), g).to(dev), rnd((B, H, Sk, 64), g).to(dev)
    dout = rnd((B, H, Sq, 64), g).to(dev)
    bias = None
    if biased:
        mask = torch.zeros(B, Sk)
        for b in range(B):
            mask[b, : max(1, (Sk * (b + 1)) // (B + 1))] = 1
        bias = ((1 - mask.to(bf16)) * -10000.0).float().to(dev)
    res = {}
    for few in ("1", "0"):
        sw("FTMI_ATTN_FEWKEYS", few)
        if few == "0":
            sw("FTMI_ATTN_FEWKEYS_FWD", "0")
        out, lse = ops.attn_fwd(q, k, v, bias)
        dq, dk, dv = ops.attn_bwd(q, k, v, out, lse, dout, bias)
        torch.cuda.synchronize()
        res[few] = (out, lse, dq, dk, dv)
    for name, x, y in zip(("out", "lse", "dq", "dk", "dv"), res["1"], res["0"]):
        assert torch.equal(x, y), f"{name}: resident few-keys kernels differ from the general kernels (max |diff| {(x.float() - y.float()).abs().max().item():.3e})"


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 32, 2688, 2688), (1, 3, 100, 128), (2, 2, 700, 1024), (1, 2, 129, 192), (1, 1, 2688, 64), (2, 16, 1000, 1000), (1, 4, 300, 1777)])
def test_pipelined_dq_kernels_match_the_kernel_they_replace(B, H, Sq, Sk, sw):
    """Round 5: the hand-placed dQ pipelines (csrc/attention_pl.hip.h: software pipeline over 32 x 32 units, every instruction of the loop an asm statement,
    three-slot K / V ring).  Their `x0` streams do the arithmetic of attn_bwd_dq2_kernel statement for statement -- same MFMA chains, fma / exp2 / sub / mul /
    pack per score -- so dQ (and dK / dV through the delta they publish) must be THE SAME BITS as with FTMI_ATTN_PL=0, for 32 rows x two waves per SIMD
    (0x001) and 64 rows x one wave (0x101), at whole and ragged query counts, one and many key tiles.  The shipped streams (0x011 / 0x111) put -delta into the
    accumulator input of the dP chain: one rounding differs per score, checked against the old kernel to 1e-3 and against fp32 autograd like every attention case.
    Sk = 64 (one tile) keeps the old kernel: the switch must fall through.  Ragged key counts (1000, 1777: CogVideoX's 17 776 = 277 x 64 + 48 in the lab, profiles/
    r05_attn_lab_4_ragged.txt): the bounds-checked DMA zero-fills the rows past the end, and zero K rows cancel in the dQ products -- same bits as the masking kernel."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(11)
    q, k, v = rnd((B, H, Sq, 64), g), rnd((B, H, Sk, 64), g), rnd((B, H, Sk, 64), g)
    dout = rnd((B, H, Sq, 64), g)
    qd, kd, vd, dd = q.to(dev), k.to(dev), v.to(dev), dout.to(dev)
    out, lse = ops.attn_fwd(qd, kd, vd, None)
    res = {}
    for pl in ("0", "0x001", "0x101", "0x011", "0x111"):
        sw("FTMI_ATTN_PL", pl)
        res[pl] = ops.attn_bwd(qd, kd, vd, out, lse, dd, None)
        torch.cuda.synchronize()
    for pl in ("0x001", "0x101"):
        for name, x, y in zip(("dq", "dk", "dv"), res[pl], res["0"]):
            assert torch.equal(x, y), f"FTMI_ATTN_PL={pl} {name}: not bit-identical to the compiler-scheduled kernel (max |diff| {(x.float() - y.float()).abs().max().item():.3e})"
    _, dq_ref, dk_ref, dv_ref = _attn_ref(q, k, v, None, dout)
    for pl in ("0x011", "0x111"):
        rel = ((res[pl][0].float() - res["0"][0].float()).norm() / res["0"][0].float().norm()).item()
        print(f"[parity] dq pipeline {pl} vs attn_bwd_dq2_kernel B{B} H{H} {Sq}x{Sk}: rel_l2 {rel:.2e}")
        assert rel < 1e-3 and torch.equal(res[pl][1], res["0"][1]) and torch.equal(res[pl][2], res["0"][2])  # delta (hence dK / dV) is the same number
        report(f"attn-pl {pl} B{B} H{H} {Sq}x{Sk} dq", res[pl][0], dq_ref, 1e-2)


@pytest.mark.parametrize("B,H,Sq,Sk", [(2, 32, 2688, 2688), (1, 16, 1024, 2048), (1, 16, 1024, 2000), (1, 64, 128, 512), (2, 16, 1000, 1000), (1, 32, 777, 1100)])
def test_pipelined_dkdv_kernel_is_bit_identical_to_the_kernel_it_replaces(B, H, Sq, Sk, sw):
    """Round 5: attn_bwd_dkdv_pl_kernel (64 keys per wave, one wave per SIMD, the same software pipeline as the dQ kernel; the lse / delta rows arrive by DMA and
    wave 0 turns them into the accumulator inputs -lse / sl and -delta before the tile's hand-over barrier) does the arithmetic of attn_bwd_dkdv_kernel<1, 2>
    statement for statement: dK and dV must be the same bits with FTMI_ATTN_PL bit 1 on and off -- whole and ragged key counts, few and many query tiles.
    Ragged QUERY counts (1000, 777): the DMA zero-fills the rows past the end -- Q = dO = 0 there, so they add nothing whatever p they get.
    (Shapes with fewer than 256 key blocks x heads keep the split-query kernel, key biases the old one: covered by the other cases.)"""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(13)
    q, k, v = rnd((B, H, Sq, 64), g).to(dev), rnd((B, H, Sk, 64), g).to(dev), rnd((B, H, Sk, 64), g).to(dev)
    dout = rnd((B, H, Sq, 64), g).to(dev)
    out, lse = ops.attn_fwd(q, k, v, None)
    res = {}
    for pl in ("0x111", "0x1113", "0x2113"):  # same dQ kernel (it publishes delta), dK / dV kernel old | pipelined, rolling order | pipelined, groups of four
        sw("FTMI_ATTN_PL", pl)
        res[pl] = ops.attn_bwd(q, k, v, out, lse, dout, None)
        torch.cuda.synchronize()
    for pl in ("0x1113", "0x2113"):
        for name, x, y in zip(("dq", "dk", "dv"), res[pl], res["0x111"]):
            assert torch.equal(x, y), f"FTMI_ATTN_PL={pl} {name}: not bit-identical (max |diff| {(x.float() - y.float()).abs().max().item():.3e})"


@pytest.mark.parametrize("B,H,Sq,Sk,biased", [(1, 4, 1024, 1024, False), (2, 3, 1000, 777, False), (1, 2, 2112, 2304, True), (1, 8, 4096, 4096, False), (1, 2, 520, 136, True)])
def test_fused_head_dim_128_dkdv_kernel_is_bit_identical_to_the_two_passes(B, H, Sq, Sk, biased, sw):
    """Round 5: attn_bwd_dkdv_pl128_kernel (head_dim 128: Wan, HunyuanVideo) computes dK and dV in ONE pass over the queries -- 32 keys per wave, one wave per
    SIMD, single-buffered scores, every instruction placed -- where attn_bwd_dkdv_kernel<2, 0> + <2, 1> make two (4 executed matmuls instead of 5).  Same
    arithmetic statement for statement, the key bias (HunyuanVideo's text mask, -inf entries included) as the addend of the fma that scales the scores: dK and dV
    must be the same bits with FTMI_ATTN_PL bit 3 on and off; ragged query and key counts included (zero-filled rows by the bounds-checked DMA)."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(17)
    q, k, v = rnd((B, H, Sq, 128), g).to(dev), rnd((B, H, Sk, 128), g).to(dev), rnd((B, H, Sk, 128), g).to(dev)
    dout = rnd((B, H, Sq, 128), g).to(dev)
    bias = None
    if biased:
        bias = torch.zeros(B, Sk)
        bias[:, Sk - Sk // 5:] = float("-inf")  # a padded tail, like a text mask
        bias[:, : Sk // 2] = torch.randn(B, Sk // 2, generator=g) * 0.3
        bias = bias.to(dev)
    out, lse = ops.attn_fwd(q, k, v, bias)
    res = {}
    for pl in ("0", "0x8"):
        sw("FTMI_ATTN_PL", pl)
        res[pl] = ops.attn_bwd(q, k, v, out, lse, dout, bias)
        torch.cuda.synchronize()
    for name, x, y in zip(("dq", "dk", "dv"), res["0x8"], res["0"]):
        assert not torch.isnan(x.float()).any(), f"{name}: NaN"
        assert torch.equal(x, y), f"FTMI_ATTN_PL=0x8 {name}: not bit-identical (max |diff| {(x.float() - y.float()).abs().max().item():.3e})"


@pytest.mark.parametrize("M", [5376, 5400, 17776, 2700])
def test_lora_down_projection_64_row_tiles_are_bit_identical(M, sw):
    """Round 5: gemm_nt_skinny4_kernel (64-row tiles, one round of workgroups, two-stage 16-KB ring per wave) keeps the K split, the MFMA order per accumulator
    and the cross-wave / cross-plane reduction order of gemm_nt_skinny2_kernel: the fp32-equivalent (hi | lo | hi) down-projection must come out bit for bit
    the same, also with a ragged last row tile (M = 5400, 17776) and where the tile count keeps the launch on the old kernel (M = 2700: both settings equal
    trivially).  Through the C ABI (ftmi_linear_lora_fwd's x A^T); the three-adapter launch of the fused q|k|v (N = 384) is compared bit for bit by
    tools/skinny_lab.hip (profiles/r05_skinny_lab_*.txt) and runs in every DiT parity case."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(31)
    K, r, s = 2048, 64, 0.5
    x = rnd((M, K), g).to(dev)
    A = (torch.randn(r, K, generator=g) / math.sqrt(K)).to(dev)
    w = rnd((K, K), g, 1 / math.sqrt(K)).to(dev)
    Bm = (torch.randn(K, r, generator=g) * 0.05).to(dev)
    got = {}
    for sk4 in ("0", "1"):
        sw("FTMI_SKINNY4", sk4)
        y, xa = ops.linear_lora_fwd(x, w, None, A, Bm, s, variant=8)
        torch.cuda.synchronize()
        got[sk4] = (y, xa)
    for name, a_, b_ in zip(("y", "xa"), got["1"], got["0"]):
        assert torch.equal(a_, b_), f"{name}: 64-row skinny kernel differs from the 32-row kernel (max |diff| {(a_.float() - b_.float()).abs().max().item():.3e})"
    xa64 = x.double().cpu() @ A.double().cpu().t() * s
    xa = got["1"][1].float().cpu()
    err = ((xa[:, :r].double() + xa[:, r:2 * r].double()) - xa64).norm() / xa64.norm()
    assert err < 2e-5


@pytest.mark.parametrize("B,H,Sq,Sk,biased", [(2, 3, 300, 257, True), (1, 2, 128, 128, False), (1, 2, 200, 330, False), (1, 4, 2688, 2688, False),
                                              (2, 2, 2688, 512, True)])
def test_attention_head_dim_128_fwd_bwd(B, H, Sq, Sk, biased):
    """Head size of Wan / HunyuanVideo: the forward kernel templated on head_dim / 64, the 32-row dQ kernel, and dK / dV from the fused
    hand-placed pass attn_bwd_dkdv_pl128_kernel -- every case here has >= 128 queries and keys, which is all that kernel's dispatch asks for (FTMI_ATTN_PL bit 3,
    on by default), so this IS its direct check against fp32 softmax attention + autograd, with and without a key bias, whole and ragged tiles."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(1)
    q, k, v = rnd((B, H, Sq, 128), g), rnd((B, H, Sk, 128), g), rnd((B, H, Sk, 128), g)
    dout = rnd((B, H, Sq, 128), g)
    bias = None
    if biased:
        mask = torch.zeros(B, Sk)
        for b in range(B):
            mask[b, : max(1, (Sk * (b + 1)) // (B + 1))] = 1
        bias = ((1 - mask.to(bf16)) * -10000.0).float()
    o_ref, dq_ref, dk_ref, dv_ref = _attn_ref(q, k, v, bias, dout)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    bd = None if bias is None else bias.to(dev)
    out, lse = ops.attn_fwd(qd, kd, vd, bd)
    tag = f"attn d128 B{B} H{H} {Sq}x{Sk}{' bias' if biased else ''}"
    report(tag + " fwd", out, o_ref, 6e-3)
    assert (out.float().cpu() - o_ref).abs().max() < 5e-3 * max(1.0, o_ref.abs().max().item())
    sc = (q.float() @ k.float().transpose(-1, -2)) / math.sqrt(128.0) + (0 if bias is None else bias[:, None, None, :])
    report(tag + " lse", lse.cpu() * math.log(2.0), torch.logsumexp(sc, dim=-1), 1e-4)
    dq, dk, dv = ops.attn_bwd(qd, kd, vd, out, lse, dout.to(dev), bd)
    report(tag + " dq", dq, dq_ref, 1e-2)
    report(tag + " dk", dk, dk_ref, 1e-2)
    report(tag + " dv", dv, dv_ref, 1e-2)


def test_attention_strided_layout():
    """[B,S,H*d] storage viewed as [B,H,S,d] (what the DiT uses) == contiguous [B,H,S,d]."""
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, H, S = 2, 4, 192
    qkv = rnd((B, S, 3, H, 64), g).to(dev)
    q, k, v = (qkv[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    out_s, _ = ops.attn_fwd(q, k, v)
    out_c, _ = ops.attn_fwd(q.contiguous(), k.contiguous(), v.contiguous())
    assert torch.equal(out_s.contiguous(), out_c.contiguous())


def test_attention_rejects_bad_arguments():
    from finetrainers_amd import ops

    dev = _dev()
    q = torch.zeros(1, 2, 64, 32, dtype=bf16, device=dev)
    with pytest.raises(ValueError):
        ops.attn_fwd(q, q, q)
    with pytest.raises(ValueError):
        ops.attn_fwd(torch.zeros(1, 2, 64, 64, dtype=torch.float16, device=dev), q, q)


# ----------------------------------------------------------------------------------------------------
def test_noise_pack_matches_oracle():
    from finetrainers_amd import ops
    from oracle import ltx

    dev = _dev()
    g = torch.Generator().manual_seed(9)
    B, C, F, H, W = 2, 128, 3, 4, 6
    lat, noise = rnd((B, C, F, H, W), g), rnd((B, C, F, H, W), g)
    mean, std = torch.randn(C, generator=g) * 0.1, 1 + 0.2 * torch.rand(C, generator=g)
    sig = torch.tensor([0.3, 0.9])
    for ffs in (None, torch.tensor([0.1, 0.6])):
        x0 = ltx.normalize_latents(lat, mean, std)
        s5 = sig.view(-1, 1, 1, 1, 1)
        if ffs is None:
            noisy = ltx.flow_match_xt(x0, noise, s5)
        else:
            f5 = torch.min(ffs.view(-1, 1, 1, 1, 1), s5.new_full(s5.shape, 0.25))
            noisy = torch.cat([ltx.flow_match_xt(x0[:, :, :1], noise[:, :, :1], f5), ltx.flow_match_xt(x0[:, :, 1:], noise[:, :, 1:], s5)], dim=2)
        ref_xt = ltx.pack_latents(noisy).to(bf16)
        ref_tg = ltx.flow_match_target(ltx.pack_latents(noise), ltx.pack_latents(x0))
        ffd = None if ffs is None else torch.min(ffs, torch.full_like(ffs, 0.25)).to(dev)
        xt, tg = ops.noise_pack(lat.to(dev), noise.to(dev), mean.to(dev), std.to(dev), sig.to(dev), ffd, H * W if ffs is not None else 0)
        assert torch.equal(xt.cpu(), ref_xt), (xt.cpu().float() - ref_xt.float()).abs().max()
        assert torch.equal(tg.cpu(), ref_tg)


def test_mse_loss_matches_oracle():
    from finetrainers_amd import ops
    from oracle import ltx

    dev = _dev()
    g = torch.Generator().manual_seed(4)
    pred, target = rnd((2, 96, 128), g), rnd((2, 96, 128), g)
    sig = torch.tensor([0.3, 0.9]).view(-1, 1, 1).expand(-1, 96, -1)
    for scheme in ("none", "sigma_sqrt", "cosmap"):
        p = pred.clone().requires_grad_(True)
        ref = ltx.sft_loss(p, target, sig, scheme)
        ref.backward()
        w = ltx.compute_loss_weighting_for_sd3(scheme, torch.tensor([0.3, 0.9])).float()
        loss, dpred = ops.mse_loss(pred.to(dev), target.to(dev), w.to(dev))
        assert abs(loss.item() - ref.item()) <= 2e-6 * abs(ref.item()), (loss.item(), ref.item())
        report(f"mse dpred {scheme}", dpred, p.grad, 2e-3)


def test_posterior_sample_matches_eager_bf16():
    """compute_posterior = False (precomputed moments): mean + exp(0.5 * clamp(logvar)) * eps against the eager bf16 graph on the CPU."""
    from finetrainers_amd import ops
    from oracle import ltx

    dev = _dev()
    g = torch.Generator().manual_seed(21)
    mom = torch.randn(2, 256, 3, 4, 6, generator=g)
    mom[:, 128:] = mom[:, 128:] * 6.0 - 4.0  # log-variances, a few beyond the clamp on either side
    mom[0, 128, 0, 0, :3] = torch.tensor([-45.0, 25.0, 20.0])
    mom = mom.to(bf16)
    eps = torch.randn(2, 128, 3, 4, 6, generator=g).to(bf16)
    ref = ltx.posterior_sample(mom, eps)
    out = ops.posterior_sample(mom.to(dev), eps.to(dev)).cpu()
    # expf on the two sides may differ in the last fp32 bit: at most one bf16 ulp on a handful of entries
    diff = (out.float() - ref.float()).abs()
    tol = ref.float().abs() * 2.0 ** -7 + 1e-30
    assert (diff <= tol).all() and (diff > 0).float().mean().item() < 5e-3, ((diff > 0).float().mean().item(), (diff / tol).max().item())
    with pytest.raises(ValueError):
        ops.posterior_sample(mom.to(dev)[:, :200], eps.to(dev))


def test_clip_adamw_matches_torch():
    from finetrainers_amd import ops
    from oracle import ltx

    dev = _dev()
    g = torch.Generator().manual_seed(8)
    n = 64 * 2048 * 3 + 5
    for gscale in (1e-4, 3.0):  # below / above the clip threshold
        p0 = torch.randn(n, generator=g) * 0.02
        p_ref = torch.nn.Parameter(p0.clone())
        opt = torch.optim.AdamW([p_ref], lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4, fused=False)
        p = p0.clone().to(dev)
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        for step in (1, 2, 3):
            grad = torch.randn(n, generator=g) * gscale
            p_ref.grad = grad.clone()
            gn_ref = ltx.clip_grad_norm_([p_ref], 1.0)
            opt.step()
            gn = ops.clip_adamw_step(p, grad.to(dev), m, v, step, lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4, max_norm=1.0)
            assert abs(gn.item() - gn_ref.item()) <= 1e-5 * gn_ref.item()
        torch.testing.assert_close(p.cpu(), p_ref.detach(), rtol=2e-6, atol=1e-9)


def test_clip_grad_norm_in_place():
    from finetrainers_amd import ops

    dev = _dev()
    g = torch.randn(3_000_003, generator=torch.Generator(device=dev).manual_seed(2), device=dev) * 1e-3
    ref_norm = g.double().norm().item()
    keep = g.clone()
    n = ops.clip_grad_norm_(g, max_norm=10.0)  # inside the bound: untouched
    assert torch.equal(g, keep) and abs(n.item() - ref_norm) <= 2e-6 * ref_norm
    n = ops.clip_grad_norm_(g, max_norm=0.5 * ref_norm)
    assert abs(n.item() - ref_norm) <= 2e-6 * ref_norm
    coef = 0.5 * ref_norm / (ref_norm + 1e-6)
    torch.testing.assert_close(g, keep * coef, rtol=2e-7, atol=0)
    with pytest.raises(ValueError):
        ops.clip_grad_norm_(g.cpu(), 1.0)


def test_grad_norm_is_bitwise_reproducible():
    """The clip's total norm is reduced in a fixed order: the same gradients give the same bits on every call (and on every rank of a
    data-parallel job -- a last-bit difference in the clip coefficient would let replicas drift apart)."""
    from finetrainers_amd import ops

    dev = _dev()
    n = 58_720_256  # the full LoRA buffer of the production model (2048 blocks in the first pass)
    g = torch.randn(n, generator=torch.Generator(device=dev).manual_seed(11), device=dev) * 1e-3
    p, m, v = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    scratch = torch.zeros(ops.CLIP_SCRATCH_FLOATS, device=dev)
    norms = {ops.clip_adamw_step(p, g, m, v, 1, lr=0.0, max_norm=1.0, scratch=scratch).item() for _ in range(20)}
    assert len(norms) == 1, norms
    ref = g.double().norm().item()
    assert abs(norms.pop() - ref) <= 2e-6 * ref
    with pytest.raises(ValueError):
        ops.clip_adamw_step(p, g, m, v, 1, lr=0.0, scratch=torch.zeros(2, device=dev))


def test_lora_refresh_layouts():
    """The bf16 (hi, lo) working copies of the flat fp32 LoRA buffer (include/ftmi355.h: ftmi_ltx_weights)."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoTransformer3DModel

    model = MI355XLTXVideoTransformer3DModel(LTXTransformerConfig(num_layers=2), device=_dev())
    model.add_adapter(r=64, lora_alpha=64)
    with torch.no_grad():
        model.lora_B.normal_(0, 0.02)
    model.refresh_lora_copies(force=True)
    torch.cuda.synchronize()
    A, Bm = model.lora_A.detach(), model.lora_B.detach()   # [L,8,r,D], [L,8,D,r]
    r = 64

    def planes(t):
        hi = t.to(bf16)
        return hi, (t - hi.float()).to(bf16)

    def interleave(hi, lo):  # [..., rows, cols] -> [..., 2 rows, cols]: groups of 32 rows, hi then lo
        sh = hi.shape
        h = hi.reshape(*sh[:-2], sh[-2] // 32, 32, sh[-1])
        l = lo.reshape(*sh[:-2], sh[-2] // 32, 32, sh[-1])
        return torch.cat([h, l], dim=-2).reshape(*sh[:-2], 2 * sh[-2], sh[-1])

    a_hi, a_lo = planes(A)
    b_hi, b_lo = planes(Bm)
    assert (a_lo.float().abs().max() > 0) and ((a_hi.float() + a_lo.float() - A).abs().max() < 2.0 ** -16 * A.abs().max())
    assert torch.equal(model.lora_a_sp, interleave(a_hi, a_lo))
    assert torch.equal(model.lora_bt_sp, interleave(b_hi.transpose(-1, -2).contiguous(), b_lo.transpose(-1, -2).contiguous()))
    assert torch.equal(model.lora_b_ext, torch.cat([b_hi, b_hi, b_lo], dim=-1))
    at_hi, at_lo = a_hi.transpose(-1, -2), a_lo.transpose(-1, -2)
    at_ext = torch.cat([at_hi, at_hi, at_lo], dim=-1)
    assert torch.equal(model.lora_at_ext, at_ext)
    for l in range(2):
        assert torch.equal(model.lora_at_qkv_ext[l], torch.cat([at_ext[l, 0], at_ext[l, 1], at_ext[l, 2]], dim=-1))


# ----------------------------------------------------------------------------------------------------
# row-wise kernels, forward AND backward, against the eager bf16 torch graph of the oracle (same rounding points)
@pytest.mark.parametrize("layernorm", [False, True])
@pytest.mark.parametrize("rows,rpb", [(8, 4), (301, 301), (5376, 2688)])
def test_norm_modulate_fwd_bwd(rows, rpb, layernorm):
    from finetrainers_amd import ops
    from oracle import ltx

    dev = _dev()
    D = 2048
    g = torch.Generator().manual_seed(rows + layernorm)
    nb = (rows + rpb - 1) // rpb
    x, dy, dres = rnd((rows, D), g, 2.0), rnd((rows, D), g), rnd((rows, D), g)
    scale, shift = rnd((nb, D), g, 0.3), rnd((nb, D), g, 0.3)
    onep = (1 + scale.float()).to(bf16)   # the eager graph materialises (1 + scale) in bf16
    xr = x.clone().requires_grad_(True)
    bidx = torch.arange(rows) // rpb
    if layernorm:
        n = torch.nn.functional.layer_norm(xr, (D,), eps=1e-6)
    else:
        n = ltx.RMSNorm(D, eps=1e-6, elementwise_affine=False)(xr)
    y_ref = n * onep[bidx] + shift[bidx]
    y_ref.backward(dy)
    y = ops.norm_modulate(x.to(dev), shift.to(dev), onep.to(dev), rpb, 1e-6, layernorm)
    tag = f"norm_modulate {'LN' if layernorm else 'RMS'} {rows}"
    report(tag + " fwd", y, y_ref.detach(), 2e-3)
    dx = ops.norm_modulate_bwd(x.to(dev), dy.to(dev), onep.to(dev), rpb, 1e-6, layernorm)
    report(tag + " bwd", dx, xr.grad, 4e-3)
    dx2 = ops.norm_modulate_bwd(x.to(dev), dy.to(dev), onep.to(dev), rpb, 1e-6, layernorm, dres=dres.to(dev))
    report(tag + " bwd+res", dx2, (dres.float() + xr.grad.float()).to(bf16), 4e-3)


@pytest.mark.parametrize("rope", [True, False])
@pytest.mark.parametrize("B,S", [(1, 32), (2, 150), (2, 2688)])
def test_qknorm_rope_fwd_bwd(B, S, rope):
    """norm_q / norm_k (rms_norm.py:17-29) + apply_rotary_emb (patch.py:23-33) and their autograd backward."""
    from finetrainers_amd import ops
    from finetrainers_amd.ltx_video.transformer import ltx_rope_tables
    from oracle import ltx

    dev = _dev()
    D = 2048
    g = torch.Generator().manual_seed(B * 1000 + S + rope)
    x, dy = rnd((B, S, D), g, 1.5), rnd((B, S, D), g)
    w = (1.0 + 0.1 * torch.randn(D, generator=g)).to(bf16)
    norm = ltx.RMSNorm(D, eps=1e-5, elementwise_affine=True)
    norm.weight.data = w.clone()
    xr = x.clone().requires_grad_(True)
    y_ref = norm(xr)
    cos = sin = None
    if rope:
        F_, H_, W_ = {32: (2, 4, 4), 150: (3, 5, 10), 2688: (7, 16, 24)}[S]
        scale = [1 / (25 / 8), 32, 32]
        cos_full, sin_full = ltx.LTXVideoRotaryPosEmbed(D)(x, F_, H_, W_, scale)
        y_ref = ltx.apply_rotary_emb(y_ref, (cos_full, sin_full))
        cos, sin = (t.to(dev) for t in ltx_rope_tables(F_, H_, W_, scale, dim=D))
    y_ref.backward(dy)
    xd = x.reshape(B * S, D).to(dev)
    y = ops.qknorm_rope(xd, w.to(dev), cos, sin, rows_per_batch=S, eps=1e-5)
    tag = f"qknorm{'+rope' if rope else ''} B{B} S{S}"
    report(tag + " fwd", y, y_ref.detach().reshape(B * S, D), 2e-3)
    dx = ops.qknorm_rope_bwd(xd, w.to(dev), dy.reshape(B * S, D).to(dev), cos, sin, rows_per_batch=S, eps=1e-5)
    report(tag + " bwd", dx, xr.grad.reshape(B * S, D), 5e-3)
