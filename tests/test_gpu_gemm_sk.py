"""The persistent 256 x 256 stream-K GEMM (tools/experimental/gemm_sk.hip, variant 60) against the one-tile-per-workgroup kernels (variant 61) on the
SAME inputs, through the C ABI.  Inside a tile both accumulate K in the same order, so every tile a single workgroup computes must come
out bit-identical; a tile split along K adds its fp32 partials in another order, which may flip the last bf16 bit of a few outputs
(bounded below).  Also: the hand-off never gave up (ftmi_gemm_sk_status), concurrent launches on two streams, repeated launches
(epoch flags), ragged M.  Run on the MI355X box: pytest -m gpu."""

import math

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


@pytest.fixture(autouse=True)
def _needs_the_research_build():
    """The stream-K kernel is a measured negative result (profiles/r03_gemm_streamk.txt): it ships only in FTMI_EXPERIMENTAL builds of the library."""
    from finetrainers_amd import _lib

    if not hasattr(_lib.load(), "ftmi_gemm_sk_status"):
        pytest.skip("libftmi355.so was built without FTMI_EXPERIMENTAL: no stream-K kernel")


def _dev():
    return torch.device("cuda", 0)


def _rnd(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g, device=_dev()) * scale).to(bf16)


def _compare(name, got, ref, split_tiles):
    """Whole tiles: bit-identical.  Split tiles: the fp32 sum of an output moves by ~1e-6 of the magnitude of its partial sums, so a value may
    land on the other side of a bf16 rounding boundary -- at the output, or at the epilogue's intermediate rounding point (pre-activation,
    base result before the LoRA term), whose one-ulp flip then passes through GELU / the second rounding.  So: a small fraction of outputs
    differs, by about an ulp of an O(1) intermediate -- while any hand-off bug (lost or stale partial, wrong tile) corrupts most of a whole
    256 x 256 tile by O(1).  Checked per tile."""
    got, ref = got.float(), ref.float()
    neq = (got != ref)
    frac = neq.float().mean().item()
    M, N = ref.shape
    Mp = (M + 255) // 256 * 256
    pad = torch.zeros((Mp, N), dtype=torch.bool, device=ref.device)
    pad[:M] = neq
    per_tile = pad.view(Mp // 256, 256, N // 256, 256).float().mean(dim=(1, 3))
    worst_tile = per_tile.max().item()
    excess = ((got - ref).abs() - 0.05 * (ref.abs() + 1.0)).max().item()
    print(f"[sk] {name:46s} mismatching outputs {frac:.2e} (worst tile {worst_tile:.2e}); |diff| beyond 0.05 (|ref| + 1): {max(excess, 0.0):.2e}")
    if not split_tiles:
        assert frac == 0.0, f"{name}: whole-tile launches must be bit-identical to the one-tile-per-workgroup kernel"
    else:
        assert worst_tile < 0.01 and excess <= 0.0, f"{name}: a tile differs in {worst_tile:.2e} of its outputs, worst excess {excess:.2e}"


SHAPES = [
    # M, N, K                      (21 x 8 = 168 tiles: all stream-K)   (504)            (672)             (K = 8192)        (dqkv)
    (5376, 2048, 2048), (5376, 6144, 2048), (5376, 8192, 2048), (5376, 2048, 8192), (5376, 2048, 6144),
    (4096, 4096, 512),   # 256 tiles: one whole round, nothing split -> bit-identical
    (2688, 2048, 2048),  # batch 1: 11 ragged row tiles (2688 = 10.5 x 256)
    (1100, 512, 256),    # few tiles, short K: shares snap to tile edges
]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_sk_store_matches_tile_kernel(M, N, K):
    from finetrainers_amd import _lib, ops

    g = torch.Generator(device=_dev()).manual_seed(M + N + K)
    x, w, b = _rnd((M, K), g), _rnd((N, K), g, 1 / math.sqrt(K)), _rnd((N,), g)
    ref = ops.gemm_nt(x, w, b, alpha=0.5, variant=61)
    out = ops.gemm_nt(x, w, b, alpha=0.5, variant=60)
    torch.cuda.synchronize()
    ntiles = ((M + 255) // 256) * (N // 256)
    _compare(f"store {M}x{N}x{K}", out, ref, split_tiles=ntiles % 256 != 0)
    # and against fp32 arithmetic, so that both being wrong the same way cannot pass
    exact = (x.float() @ w.float().t() * 0.5 + b.float()).to(bf16)
    assert ((out.float() - exact.float()).norm() / exact.float().norm()).item() < 2e-3
    assert _lib.load().ftmi_gemm_sk_status() == 0


def test_sk_epilogues_match_tile_kernel():
    from finetrainers_amd import _lib, ops

    g = torch.Generator(device=_dev()).manual_seed(11)
    M, N, K, S = 5376, 2048, 2048, 2688
    x, w, b = _rnd((M, K), g), _rnd((N, K), g, 1 / math.sqrt(K)), _rnd((N,), g)
    resid, gate, z = _rnd((M, N), g), _rnd((2, N), g), _rnd((M, N), g)
    for v_out in (None,):
        a = ops.gemm_nt(x, w, b, epilogue=_lib.EPI_GELU, want_out2=True, variant=61)
        c = ops.gemm_nt(x, w, b, epilogue=_lib.EPI_GELU, want_out2=True, variant=60)
        _compare("gelu", c[0], a[0], True)
        _compare("gelu pre-activation", c[1], a[1], True)
        a = ops.gemm_nt(x, w, b, epilogue=_lib.EPI_RESID, resid=resid, gate=gate, rows_per_batch=S, variant=61)
        c = ops.gemm_nt(x, w, b, epilogue=_lib.EPI_RESID, resid=resid, gate=gate, rows_per_batch=S, variant=60)
        _compare("residual + gate", c, a, True)
        a = ops.gemm_nt(x, w, None, epilogue=_lib.EPI_DGELU, aux=z, variant=61)
        c = ops.gemm_nt(x, w, None, epilogue=_lib.EPI_DGELU, aux=z, variant=60)
        _compare("gelu'", c, a, True)
    # the FF1 shape: N = 8192 with the GELU epilogue and its second output
    w8, b8 = _rnd((8192, K), g, 1 / math.sqrt(K)), _rnd((8192,), g)
    a = ops.gemm_nt(x, w8, b8, epilogue=_lib.EPI_GELU, want_out2=True, variant=61)
    c = ops.gemm_nt(x, w8, b8, epilogue=_lib.EPI_GELU, want_out2=True, variant=60)
    _compare("gelu N=8192", c[0], a[0], True)
    _compare("gelu N=8192 pre-activation", c[1], a[1], True)
    torch.cuda.synchronize()
    assert _lib.load().ftmi_gemm_sk_status() == 0


@pytest.mark.parametrize("M,N,K", [(5376, 2048, 2048), (5376, 6144, 2048), (2688, 2048, 2048)])
def test_sk_lora_extension_matches_tile_kernel(M, N, K):
    """The fused LoRA K-extension (base rounded to bf16 first, then the (hi, lo, hi) planes): forward with the q|k|v-style plain layout and the
    backward's dgrad, stream-K against the tile kernels; the extension runs only in the workgroup that owns the tile, after the partials."""
    from finetrainers_amd import _lib, ops

    g = torch.Generator(device=_dev()).manual_seed(23)
    r, s = 64, 0.5
    x, w, b = _rnd((M, K), g), _rnd((N, K), g, 1 / math.sqrt(K)), _rnd((N,), g)
    A = torch.randn(r, K, generator=g, device=_dev()) / math.sqrt(K)
    Bm = torch.randn(N, r, generator=g, device=_dev()) * 0.05
    y1, xa1 = ops.linear_lora_fwd(x, w, b, A, Bm, s, variant=61)
    y2, xa2 = ops.linear_lora_fwd(x, w, b, A, Bm, s, variant=60)
    assert torch.equal(xa1, xa2)
    _compare(f"linear + lora fwd {M}x{N}x{K}", y2, y1, True)
    dy = _rnd((M, N), g)
    w_t = ops.transpose_bf16(w)
    dx1, ga1, gb1 = ops.linear_lora_bwd(x, dy, xa1, w_t, A, Bm, s, variant=61)
    dx2, ga2, gb2 = ops.linear_lora_bwd(x, dy, xa1, w_t, A, Bm, s, variant=60)
    _compare(f"linear + lora dgrad {M}x{K}x{N}", dx2, dx1, True)
    torch.cuda.synchronize()
    assert _lib.load().ftmi_gemm_sk_status() == 0


def test_sk_repeated_and_concurrent_launches():
    """Flags carry a per-launch epoch and are never cleared: 40 launches in a row on the current stream, then launches alternating on two more streams
    (each stream has its own partial slots; persistent launches of one device are chained through completion events, so they never run side by side), then
    the same with a third stream's unrelated kernels in between.  TWO input sets alternate, so that a consumer which read a stale partial (the previous
    launch's, of the other input set) would produce a wrong tile: every result is compared with its input set's first result.  The hand-off status word is
    printed per phase.  With the default share numbering (share = G - 1 - blockIdx: every wait is on an earlier-dispatched workgroup) it must be clean
    in all three; the first numbering (FTMI_SK_ORDER=0) raised it in 2 of 4 whole-suite runs of round 3."""
    from finetrainers_amd import _lib, ops

    lib = _lib.load()
    g = torch.Generator(device=_dev()).manual_seed(5)
    M, N, K = 5376, 2048, 2048
    sets = [(_rnd((M, K), g), _rnd((N, K), g, 1 / math.sqrt(K)), _rnd((N,), g)) for _ in range(2)]
    torch.cuda.synchronize()
    lib.ftmi_gemm_sk_status()  # (read-and-clear: start from a clean word)
    firsts = [ops.gemm_nt(x, w, b, variant=61) for x, w, b in sets]  # the one-tile kernels: an independent reference for both input sets
    ref_sk = [ops.gemm_nt(x, w, b, variant=60) for x, w, b in sets]
    for r, f in zip(ref_sk, firsts):
        assert float((r.float() - f.float()).abs().max()) < 0.1
    for i in range(40):
        out = ops.gemm_nt(*sets[i % 2], variant=60)
        assert torch.equal(out, ref_sk[i % 2]), f"launch {i} differs from the first of its input set"
    torch.cuda.synchronize()
    st1 = lib.ftmi_gemm_sk_status()
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()

    def two_streams(with_foreign_work: bool):
        outs = []
        junk = torch.randn(4096, 4096, device=_dev()) if with_foreign_work else None
        for i in range(12):
            if with_foreign_work:
                with torch.cuda.stream(s3):
                    junk = junk @ junk * 1e-4  # unrelated work that takes CUs away at varying times
            for j, st in enumerate((s1, s2)):
                with torch.cuda.stream(st):
                    outs.append(((i + j) % 2, ops.gemm_nt(*sets[(i + j) % 2], variant=60)))
        torch.cuda.synchronize()
        status = lib.ftmi_gemm_sk_status()
        for n, (k, o) in enumerate(outs):
            assert torch.equal(o, ref_sk[k]), f"launch {n} ({'with' if with_foreign_work else 'without'} foreign work) differs"
        return status

    st2 = two_streams(False)
    st3 = two_streams(True)
    print(f"[sk] hand-off status after 40 launches on one stream: {st1}; after 24 launches alternating on two streams: {st2}; with a third stream's matmuls in between: {st3}")
    import os

    if os.environ.get("FTMI_SK_ORDER", "1") == "0":
        assert st1 >= 0 and st2 >= 0 and st3 >= 0  # (< 0: the status query itself failed)
        if st1 or st2 or st3:
            pytest.xfail(f"XCD-contiguous share numbering: a bounded hand-off poll gave up (status {st1} / {st2} / {st3}); every result was right")
    else:
        assert st1 == 0 and st2 == 0 and st3 == 0


def test_sk_refuses_what_it_cannot_do():
    from finetrainers_amd import ops

    g = torch.Generator(device=_dev()).manual_seed(1)
    x, w = _rnd((512, 256), g), _rnd((256, 256), g)
    with pytest.raises((RuntimeError, ValueError)):
        ops.gemm_nt(x, w, None, variant=60)  # M < 1024
