"""CogVideoX (BASELINE config 3): the oracle restatement vs fixtures produced by EXECUTING the reference's own functions
(oracle/make_golden.py: CogVideoXModelSpecification.forward / _pad_frames, prepare_rotary_positional_embeddings, prepare_loss_weights,
prepare_sigmas) over the oracle's [upstream] sub-modules, plus structural pins of the 2b architecture."""

import torch

from oracle import cogvideox as cvx


def _eq(a, b):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    assert torch.equal(a, b), (a.float() - b.float()).abs().max()


def test_rope_tables_and_frame_padding_match_reference(golden):
    for name, (h, w, f, pt) in {"v10": (48, 48, 3, None), "v15": (48, 64, 4, 2)}.items():
        c, s = cvx.prepare_rotary_positional_embeddings(h, w, f, 8, 2, pt, 16, 192, 192)
        _eq(c, golden[f"cvx.rope_{name}.cos"])
        _eq(s, golden[f"cvx.rope_{name}.sin"])
    lat = golden["cvx.pad.in"]
    _eq(cvx.pad_frames(lat, 2).contiguous(), golden["cvx.pad.out2"])
    _eq(cvx.pad_frames(lat, 3).contiguous(), golden["cvx.pad.out3"])
    assert golden["cvx.pad.out3"].shape[1] == 6  # the reference pads a FULL group when the frame count already divides (reproduced)


def _spec_case(golden, tag, cfg):
    frames, hh, ww, seed, rank = [int(v) for v in golden[f"{tag}.meta"]]
    model = cvx.build_model(cfg, seed=0, rank=rank, alpha=float(max(rank, 1)), lora_b_std=0.02 if rank else None)
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(2, frames, cfg.in_channels, hh, ww, generator=g).bfloat16()
    text = torch.randn(2, cfg.max_text_seq_length, cfg.text_embed_dim, generator=g).bfloat16()
    with torch.no_grad():
        pred, target, sig = cvx.spec_forward(model, cvx.CogVideoXDDIMScheduler(), lat, text, torch.tensor([0.25, 0.7]),
                                             generator=torch.Generator().manual_seed(seed + 100))
    _eq(pred.contiguous(), golden[f"{tag}.pred"])
    _eq(target.contiguous(), golden[f"{tag}.target"])
    _eq(sig.contiguous(), golden[f"{tag}.sigmas"])


def test_spec_forward_matches_reference(golden):
    _spec_case(golden, "cvx.spec_dummy", cvx.CogVideoXConfig.dummy())
    _spec_case(golden, "cvx.spec_2b1", cvx.CogVideoXConfig(num_layers=1, sample_height=8, sample_width=12, sample_frames=9))


def test_ddim_loss_weights_and_sigma_sampling_match_reference(golden):
    sch = cvx.CogVideoXDDIMScheduler()
    ts = torch.tensor([0, 250, 700, 999])
    _eq(1 / (1 - sch.alphas_cumprod[ts]), golden["cvx.loss_weights"])
    # loss = mean w(t) (pred - target)^2 with the same weights
    pred, target = torch.randn(4, 2, 3, 4, 4).bfloat16(), torch.randn(4, 2, 3, 4, 4).bfloat16()
    sig = ts.float() / 1000.0 + 1e-4
    want = (golden["cvx.loss_weights"].view(-1, 1, 1, 1, 1) * (pred.float() - target.float()) ** 2).mean(dim=(1, 2, 3, 4)).mean()
    torch.testing.assert_close(cvx.sft_loss(pred, target, sig, sch), want)
    # uniform sigma sampling of the DDIM branch (utils/diffusion.py:105-108): same generator, same table lookup
    from oracle import ltx

    u = torch.rand(size=(16,), generator=torch.Generator().manual_seed(33))
    _eq(ltx.scheduler_sigmas()[(u * 1000).long()], golden["cvx.sigmas"])
    assert sch.alphas_cumprod.shape == (1000,) and 0 < sch.alphas_cumprod[-1] < sch.alphas_cumprod[0] < 1


def test_structure_2b():
    """CogVideoX-2b: 30 blocks x (30 heads x 64) = width 1920; joint sequence 226 text + 13 x 30 x 45 video tokens = 17 776 at 49x480x720;
    default LoRA regex -> 4 adapters per block."""
    cfg = cvx.CogVideoXConfig(num_layers=1)
    m = cvx.CogVideoXTransformer3DModel(cfg)
    blk = sum(p.numel() for p in m.transformer_blocks[0].parameters())
    rest = sum(p.numel() for p in m.parameters()) - blk
    total = rest + 30 * blk
    assert cfg.inner_dim == 1920 and 1.6e9 < total < 1.8e9, total
    names = cvx.add_lora(m, 64, 64)
    assert len(names) * 30 == 120
    lora = sum(p.numel() for n, p in m.named_parameters() if "lora_" in n) * 30
    assert lora == 30 * 4 * 2 * 64 * 1920
    f, h, w = (49 - 1) // 4 + 1, 480 // 8 // 2, 720 // 8 // 2
    assert 226 + f * h * w == 17776
    assert m.patch_embed.pos_embedding.shape == (1, 226 + f * h * w, 1920)  # sincos table incl. the (zero) text part


def test_lora_identity_and_gradient_flow():
    cfg = cvx.CogVideoXConfig.dummy()
    m = cvx.build_model(cfg, seed=0, rank=4, alpha=4.0, dtype=torch.float32)  # B = 0
    g = torch.Generator().manual_seed(1)
    lat, text = torch.randn(2, 3, 4, 6, 6, generator=g), torch.randn(2, 16, 32, generator=g)
    sch = cvx.CogVideoXDDIMScheduler()
    noise = torch.randn(2, 3, 4, 6, 6, generator=g)
    pred, target, sig = cvx.spec_forward(m, sch, lat, text, torch.tensor([0.3, 0.8]), noise=noise)
    loss = cvx.sft_loss(pred, target, sig, sch)
    loss.backward()
    for n, p in m.named_parameters():
        if "lora_A" in n:
            assert p.grad.abs().max() == 0
        if "lora_B" in n:
            assert p.grad.abs().max() > 0
    m0 = cvx.build_model(cfg, seed=0, rank=0, dtype=torch.float32)
    pred0, _, _ = cvx.spec_forward(m0, sch, lat, text, torch.tensor([0.3, 0.8]), noise=noise)
    torch.testing.assert_close(pred, pred0)
