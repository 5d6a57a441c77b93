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


def test_native_sdpa_dispatch():
    """The oracle's attention IS ``torch.nn.functional.scaled_dot_product_attention`` (what the reference's native provider
    calls, attention_dispatch.py:938-962).  Record which CPU kernels that dispatches to for the oracle's bf16 tensors: the
    fused CPU flash kernels, forward and backward -- bf16 probabilities, like a GPU flash kernel, not the fp32 math form."""
    from torch.profiler import ProfilerActivity, profile

    q, k, v = (torch.randn(1, 4, 48, 64).to(torch.bfloat16).requires_grad_() for _ in range(3))
    mask = torch.zeros(1, 4, 1, 48, dtype=torch.bfloat16)
    mask[..., 40:] = -10000.0
    with profile(activities=[ProfilerActivity.CPU]) as prof:
        ltx.native_sdpa(q, k, v, mask).float().sum().backward()
    names = {e.key for e in prof.key_averages()}
    print("[sdpa-dispatch]", sorted(n for n in names if "attention" in n))
    assert "aten::scaled_dot_product_attention" in names
    assert "aten::_scaled_dot_product_flash_attention_for_cpu" in names
    assert "aten::_scaled_dot_product_flash_attention_for_cpu_backward" in names
    # and it agrees with the fp32-probability math form to the reference's own provider tolerance (atol 5e-3)
    # (the reference's own shape and seed: tests/models/attention_dispatch.py:113-120)
    torch.manual_seed(0)
    q, k, v = (torch.randn(2, 8, 256, 64).to(torch.bfloat16) for _ in range(3))
    assert (ltx.native_sdpa(q, k, v, None).float() - ltx.sdpa_math(q, k, v, None).float()).abs().max() < 5e-3


def test_lora_gradient_tolerance_triangle():
    """Why the GPU parity tests bound the LoRA-gradient error by a measured figure and not by the north star's 1e-3.

    Triangle on one case (production width, 1 block, 32 tokens; identical inputs, identical weights):
      (a) bf16 oracle  vs  the same oracle with another fp32 SUMMATION ORDER in the frozen linears -- every rounding point
          identical (``ltx.accumulation_order_variant``): the gradients move by ~4e-3 (relative L2, all adapters), the loss
          by ~1e-5.  This is the floor for ANY two implementations of the reference's bf16 graph (CPU vs GPU included);
      (b) bf16 oracle  vs  the fp32 model on the same (bf16-valued) weights and inputs: ~8e-3 -- the reference's own
          distance from exact arithmetic;
      (c) [GPU tests] kernel vs bf16 oracle must stay below 1.5 x (a) measured on the same case.
    A 1e-3 bound on gradients is below (a): it cannot be met by any bf16 implementation, the reference on other
    hardware included; the loss bound of 1e-3 is met with two orders of magnitude to spare."""
    cfg = ltx.LTXConfig.production(num_layers=1)
    model = ltx.build_model(cfg, seed=0, rank=64, alpha=64.0, lora_b_std=0.02)
    inp = ltx.synth_inputs(cfg, 1, 2, 4, 4, seed=3, mask_lens=[32], sigmas=[0.25])
    g_ref, l_ref = ltx.lora_grads(model, inp)
    with ltx.accumulation_order_variant(512):
        g_ord, l_ord = ltx.lora_grads(model, inp)
    m32 = ltx.build_model(cfg, seed=0, rank=64, alpha=64.0, lora_b_std=0.02, dtype=torch.float32)
    m32.load_state_dict({k: v.float() for k, v in model.state_dict().items()})
    inp32 = ltx.synth_inputs(cfg, 1, 2, 4, 4, seed=3, mask_lens=[32], sigmas=[0.25], dtype=torch.float32)
    for f in ("latents", "noise", "encoder_hidden_states", "encoder_attention_mask"):
        setattr(inp32, f, getattr(inp, f).float())
    g_32, l_32 = ltx.lora_grads(m32, inp32)
    a_glob, a_worst = ltx.grads_rel_l2(g_ord, g_ref)
    b_glob, b_worst = ltx.grads_rel_l2(g_ref, g_32)
    print(f"[triangle] (a) summation order: grad {a_glob:.3e} / worst adapter {a_worst:.3e}, loss {abs(l_ord - l_ref) / l_ref:.1e};  "
          f"(b) bf16 vs fp32: grad {b_glob:.3e} / {b_worst:.3e}, loss {abs(l_32 - l_ref) / l_ref:.1e}")
    assert abs(l_ord - l_ref) / l_ref < 1e-4 and abs(l_32 - l_ref) / l_ref < 1e-3
    assert 1.5e-3 < a_glob < 1e-2, "summation order alone moves the gradients by more than the north star's 1e-3"
    assert a_glob < b_glob < 3e-2
