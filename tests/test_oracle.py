"""The oracle restatement vs fixtures produced by the reference's own functions
(oracle/make_golden.py) + the structural pins of SURVEY section 7 step 0."""

import torch

from oracle import ltx


def _eq(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a, b), (a.float() - b.float()).abs().max()


def test_rope_apply_matches_reference(golden):
    out = ltx.apply_rotary_emb(golden["rope.x"], (golden["rope.cos"], golden["rope.sin"]))
    _eq(out, golden["rope.out"])


def test_rmsnorm_patch_matches_reference(golden):
    x, w = golden["rms.x"], golden["rms.w"]
    n = ltx.RMSNorm(64, eps=1e-5, elementwise_affine=True)
    n.weight.data = w
    _eq(n(x), golden["rms.out_affine"])
    _eq(ltx.RMSNorm(64, eps=1e-6, elementwise_affine=False)(x), golden["rms.out_plain"])


def test_pack_normalize_flowmatch_match_reference(golden):
    lat = golden["pack.in"]
    _eq(ltx.pack_latents(lat, 1, 1).contiguous(), golden["pack.out"])
    _eq(ltx.pack_latents(lat[:, :, :2], 2, 1).contiguous(), golden["pack.out_p2"])
    _eq(ltx.normalize_latents(lat, golden["norm.mean"], golden["norm.std"]), golden["norm.out"])
    _eq(ltx.flow_match_xt(golden["fm.x0"], golden["fm.n"], golden["fm.t"]), golden["fm.xt"])
    _eq(ltx.flow_match_target(golden["fm.n"], golden["fm.x0"]), golden["fm.target"])


def test_sigma_sampling_matches_reference(golden):
    table = ltx.scheduler_sigmas()
    assert table.shape == (1000,) and table[0] == 1.0 and abs(table[-1].item() - 0.001) < 1e-9
    for scheme in ("none", "logit_normal", "mode"):
        gen = torch.Generator().manual_seed(21)
        s = ltx.prepare_sigmas(table, 16, flow_weighting_scheme=scheme, generator=gen)
        _eq(s, golden[f"sigmas.{scheme}"])


def test_clip_grad_norm_matches_reference(golden):
    for tag in ("clip_big", "clip_small"):
        ps = []
        for i in range(4):
            g = golden[f"{tag}.g{i}"]
            p = torch.nn.Parameter(torch.zeros_like(g))
            p.grad = g.clone()
            ps.append(p)
        ps.append(torch.nn.Parameter(torch.zeros(3)))
        total = ltx.clip_grad_norm_(ps, 1.0)
        torch.testing.assert_close(total.reshape(1), golden[f"{tag}.total_norm"], rtol=1e-6, atol=0)
        for i in range(4):
            torch.testing.assert_close(ps[i].grad, golden[f"{tag}.out{i}"], rtol=1e-6, atol=0)


def _spec_case(golden, tag, cfg):
    frames, height, width, first_frame, seed, rank = [int(v) for v in golden[f"{tag}.meta"]]
    model = ltx.build_model(cfg, seed=0, rank=rank, lora_b_std=0.02 if rank else None)
    inp = ltx.synth_inputs(cfg, 1, frames, height, width, seed=seed, mask_lens=[cfg.text_seq_len // 4], sigmas=[0.7])
    inp.latents_mean = torch.randn(cfg.in_channels, generator=torch.Generator().manual_seed(5)) * 0.1
    inp.latents_std = 1.0 + 0.2 * torch.rand(cfg.in_channels, generator=torch.Generator().manual_seed(6))
    sig5 = inp.sigmas.view(-1, 1, 1, 1, 1)
    lat_n = ltx.normalize_latents(inp.latents, inp.latents_mean, inp.latents_std)
    noise = torch.zeros_like(lat_n).normal_(generator=torch.Generator().manual_seed(seed + 100))
    ffs = None
    if first_frame:
        torch.manual_seed(seed + 200)
        ffs = torch.rand_like(sig5) * sig5
    with torch.no_grad():
        pred, target, sig = ltx.spec_forward(
            model, inp.latents.clone(), inp.latents_mean, inp.latents_std, inp.encoder_hidden_states,
            inp.encoder_attention_mask, sig5, noise=noise, first_frame_sigma=ffs,
        )
    _eq(pred.contiguous(), golden[f"{tag}.pred"])
    _eq(target.contiguous(), golden[f"{tag}.target"])
    _eq(sig.contiguous(), golden[f"{tag}.sigmas"])


def test_spec_forward_dummy_matches_reference(golden):
    """Reference spec.forward + reference patched DiT forward (driving oracle sub-modules) on the
    reference's own tiny fixture config (tests/models/ltx_video/base_specification.py:48-58)."""
    _spec_case(golden, "spec_dummy", ltx.LTXConfig.dummy())
    _spec_case(golden, "spec_dummy_ff", ltx.LTXConfig.dummy())


def test_spec_forward_production_dims_matches_reference(golden):
    _spec_case(golden, "spec_prod1", ltx.LTXConfig.production(num_layers=1))
    _spec_case(golden, "spec_prod1_ff", ltx.LTXConfig.production(num_layers=1))


def test_structure_param_counts():
    """1 923 385 472 base params, 58 720 256 LoRA params @ r=64,
    224 adapters with peft-compatible names."""
    cfg = ltx.LTXConfig.production(num_layers=1)
    m = ltx.LTXVideoTransformer3DModel(cfg)
    one = sum(p.numel() for p in m.parameters())
    blk = sum(p.numel() for p in m.transformer_blocks[0].parameters())
    assert one + 27 * blk == 1_923_385_472
    names = ltx.add_lora(m, 64, 64)
    assert len(names) * 28 == 224
    lora = sum(p.numel() for _, p in ltx.lora_parameters(m))
    assert lora * 28 == 58_720_256
    keys = dict(m.named_parameters())
    assert "transformer_blocks.0.attn1.to_q.lora_A.default.weight" in keys
    assert "transformer_blocks.0.attn2.to_out.0.lora_B.default.weight" in keys
    assert keys["transformer_blocks.0.attn1.to_q.lora_A.default.weight"].dtype == torch.float32
    assert keys["transformer_blocks.0.attn1.to_q.base_layer.weight"].dtype == torch.float32  # not cast here


def test_lora_identity_at_init():
    """(iii) B = 0 at init => loss independent of A and dA == 0."""
    cfg = ltx.LTXConfig.dummy()
    m = ltx.build_model(cfg, seed=0, rank=4, alpha=4.0, dtype=torch.float32)
    inp = ltx.synth_inputs(cfg, 2, 2, 4, 4, dtype=torch.float32)
    loss, _, _ = ltx.forward_loss(m, inp)
    loss.backward()
    for n, p in ltx.lora_parameters(m):
        if "lora_A" in n:
            assert p.grad.abs().max() == 0
        else:
            assert p.grad.abs().max() > 0
    m0 = ltx.build_model(cfg, seed=0, rank=0, dtype=torch.float32)
    loss0, _, _ = ltx.forward_loss(m0, inp)
    torch.testing.assert_close(loss, loss0)


def test_sft_step_runs_and_updates():
    cfg = ltx.LTXConfig.dummy()
    m = ltx.build_model(cfg, seed=0, rank=4, alpha=4.0, dtype=torch.float32, lora_b_std=0.02)
    opt = ltx.make_optimizer(m)
    inp = ltx.synth_inputs(cfg, 2, 2, 4, 4, dtype=torch.float32)
    before = {n: p.detach().clone() for n, p in ltx.lora_parameters(m)}
    loss, gn, grads = ltx.sft_step(m, opt, inp)
    assert torch.isfinite(loss) and gn > 0
    assert any((before[n] != p).any() for n, p in ltx.lora_parameters(m))


def test_gradient_noise_floor_of_bf16_rounding_points():
    """Why the LoRA-gradient tolerance of the GPU parity tests is 1e-2 and not the north star's 1e-3.

    Moving ONE family of bf16 rounding points of the reference graph -- (i) LoRA operands (A, B, x A^T) rounded to bf16, as
    any MFMA path must, or (ii) the softmax probabilities rounded to bf16 before P.V, as every fused (flash) attention kernel
    the reference itself dispatches to on a GPU does -- already moves the LoRA gradients by 3.5e-3 ... 5e-3 (relative L2,
    whole gradient), while the loss moves by < 1e-4.  A fused bf16 implementation cannot agree with the eager bf16 graph to
    1e-3 on gradients; it can (and the GPU tests require it to) stay inside this noise floor."""
    import math

    from oracle import ltx

    cfg = ltx.LTXConfig.production(num_layers=1)
    model = ltx.build_model(cfg, seed=0, rank=64, alpha=64.0, lora_b_std=0.02)
    inp = ltx.synth_inputs(cfg, 1, 2, 4, 4, seed=3, mask_lens=[32], sigmas=[0.25])

    def grads():
        for p in model.parameters():
            p.grad = None
        loss = ltx.forward_loss(model, inp, contiguous_hidden_states=True)[0]
        loss.backward()
        return {n: p.grad.detach().clone() for n, p in ltx.lora_parameters(model)}, loss.item()

    def rel(a, b):
        num = sum((a[k] - b[k]).float().pow(2).sum().item() for k in a)
        den = sum(b[k].float().pow(2).sum().item() for k in a)
        return math.sqrt(num / den)

    class RoundBoth(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.to(torch.bfloat16).float()

        @staticmethod
        def backward(ctx, g):
            return g.to(torch.bfloat16).float()

    class RoundFwd(torch.autograd.Function):
        @staticmethod
        def forward(ctx, t):
            return t.to(torch.bfloat16).float()

        @staticmethod
        def backward(ctx, g):
            return g

    g_ref, l_ref = grads()

    orig_fwd = ltx.LoraLinear.forward

    def lora_bf16(self, x):
        result = self.base_layer(x)
        a, b = self.lora_A["default"], self.lora_B["default"]
        xa = RoundBoth.apply(torch.nn.functional.linear(x.float(), RoundBoth.apply(a.weight)) * self.scaling)
        return (result.float() + torch.nn.functional.linear(xa, RoundBoth.apply(b.weight))).to(result.dtype)

    ltx.LoraLinear.forward = lora_bf16
    try:
        g_i, l_i = grads()
    finally:
        ltx.LoraLinear.forward = orig_fwd

    orig_sdpa = ltx.sdpa_math

    def sdpa_bf16_p(q, k, v, attn_mask):
        s = torch.matmul(q.float(), k.float().transpose(-1, -2)) / math.sqrt(q.shape[-1])
        if attn_mask is not None:
            s = s + attn_mask.float()
        return torch.matmul(RoundFwd.apply(torch.softmax(s, dim=-1)), v.float()).to(q.dtype)

    ltx.sdpa_math = sdpa_bf16_p
    try:
        g_ii, l_ii = grads()
    finally:
        ltx.sdpa_math = orig_sdpa

    r_i, r_ii = rel(g_i, g_ref), rel(g_ii, g_ref)
    print(f"[noise floor] bf16 LoRA operands: grad {r_i:.3e}, loss {abs(l_i - l_ref) / l_ref:.1e};  bf16 P: grad {r_ii:.3e}, loss {abs(l_ii - l_ref) / l_ref:.1e}")
    assert abs(l_i - l_ref) / l_ref < 1e-3 and abs(l_ii - l_ref) / l_ref < 1e-3
    assert 1.5e-3 < r_i < 2e-2 and 1.5e-3 < r_ii < 2e-2
