"""oracle/hunyuan.py (the row after Wan): the spec forward against fixtures produced by executing the reference's own
``HunyuanVideoModelSpecification.forward`` (oracle/make_golden.py), and structural checks of the restated (unpinned) DiT."""

import pytest
import torch

from oracle import hunyuan as hy


@pytest.fixture(scope="module")
def golden():
    import os

    from safetensors.torch import load_file

    return load_file(os.path.join(os.path.dirname(__file__), "golden", "reference_fixtures.safetensors"))


def _stub(seen):
    def f(**kw):
        seen.update(kw)
        return ((kw["hidden_states"].float() * 0.5 + kw["guidance"].view(-1, 1, 1, 1, 1).float() * 1e-4 + kw["timestep"].view(-1, 1, 1, 1, 1).float() * 1e-3).to(kw["hidden_states"].dtype),)

    return f


@pytest.mark.parametrize("tag,posterior", [("hunyuan.spec_moments", False), ("hunyuan.spec_latents", True)])
def test_spec_forward_matches_the_reference(golden, tag, posterior):
    """(stored moments -> posterior draw | latents) x scaling factor -> flow-match mix -> integer timesteps, guidance x 1000 -> DiT keywords -> target:
    bit for bit what the reference's forward produced and handed to the transformer for the same generator."""
    lat = golden[f"{tag}.latents_in"]
    sig = golden["hunyuan.spec.sigmas"].view(-1, 1, 1, 1, 1)
    g = torch.Generator().manual_seed(91)  # the reference draws the posterior eps first (when it samples), then the flow-match noise
    C = lat.shape[1] // (1 if posterior else 2)
    eps = None if posterior else torch.randn(lat.shape[0], C, *lat.shape[2:], generator=g, dtype=lat.dtype)
    noise = torch.zeros(lat.shape[0], C, *lat.shape[2:], dtype=lat.dtype).normal_(generator=g)
    seen = {}
    cond = {"encoder_hidden_states": torch.zeros(2, 5, 16), "encoder_attention_mask": torch.ones(2, 5, dtype=torch.long), "pooled_projections": torch.zeros(2, 8)}
    pred, target, _ = hy.spec_forward(_stub(seen), lat, cond, sig, noise, scaling_factor=0.476986, guidance=6.0, compute_posterior=posterior, posterior_noise=eps)
    assert torch.equal(target, golden[f"{tag}.target"]) and torch.equal(seen["hidden_states"], golden[f"{tag}.noisy"])
    assert torch.equal(seen["guidance"], golden[f"{tag}.guidance"]) and torch.equal(seen["timestep"], golden[f"{tag}.timestep"])
    assert torch.equal(pred, golden[f"{tag}.pred"])
    assert set(seen) == {"hidden_states", "guidance", "encoder_hidden_states", "encoder_attention_mask", "pooled_projections", "timestep", "return_dict"}


def test_dit_structure_masking_and_gradients():
    """The reference's dummy configuration runs; the parameter count of the production configuration [upstream: 12.8 B]; text tokens beyond a sample's
    mask length cannot influence anything (keys masked in the blocks, masked mean + masked self-attention in the token refiner); every parameter of
    both block kinds receives a gradient; the rotary table has the per-axis layout."""
    cfg = hy.HunyuanVideoConfig.dummy()
    m = hy.build_model(cfg, seed=0, dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 3, 4, 6, generator=g)
    txt = torch.randn(2, 5, 16, generator=g)
    mask = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]])
    pooled = torch.randn(2, 8, generator=g)
    t, guid = torch.tensor([300, 800]), torch.tensor([6000.0, 6000.0])
    out = m(x, t, txt, mask, pooled, guid, return_dict=False)[0]
    assert out.shape == x.shape and torch.isfinite(out).all()
    txt2 = txt.clone()
    txt2[0, 3:] = 100.0 * torch.randn(2, 16, generator=g)  # padded positions of sample 0
    out2 = m(x, t, txt2, mask, pooled, guid, return_dict=False)[0]
    assert torch.allclose(out2, out, atol=1e-5, rtol=1e-5)
    txt3 = txt.clone()
    txt3[0, 1] += 1.0  # a real token does matter
    assert not torch.allclose(m(x, t, txt3, mask, pooled, guid, return_dict=False)[0][0], out[0], atol=1e-5)
    out.square().mean().backward()
    missing = [n for n, p in m.named_parameters() if p.grad is None or not torch.isfinite(p.grad).all()]
    assert not missing, missing
    cos, sin = hy.rotary_tables(cfg, 3, 4, 6)
    assert cos.shape == (3 * 4 * 6, 10) and torch.equal(cos[:, 0], cos[:, 1]) and torch.allclose(cos[0], torch.ones(10)) and torch.allclose(sin[0], torch.zeros(10))
    # second frame, first row / column: only the temporal axis (first 2 channels) has moved
    assert not torch.allclose(cos[24, :2], torch.ones(2)) and torch.allclose(cos[24, 2:], torch.ones(8))

    full = hy.HunyuanVideoConfig()
    d, mlp = full.inner_dim, int(full.inner_dim * full.mlp_ratio)
    lin = lambda i, o: i * o + o
    dual = 2 * lin(d, 6 * d) + 8 * lin(d, d) + 4 * full.attention_head_dim + 2 * (lin(d, mlp) + lin(mlp, d))
    single = lin(d, 3 * d) + 3 * lin(d, d) + 2 * full.attention_head_dim + lin(d, mlp) + lin(d + mlp, d)
    refiner_blk = 2 * 2 * d + 4 * lin(d, d) + lin(d, mlp) + lin(mlp, d) + lin(d, 2 * d)
    refiner = lin(256, d) + lin(d, d) + lin(full.text_embed_dim, d) + lin(d, d) + lin(full.text_embed_dim, d) + full.num_refiner_layers * refiner_blk
    cond = 2 * (lin(256, d) + lin(d, d)) + lin(full.pooled_projection_dim, d) + lin(d, d)
    total = full.num_layers * dual + full.num_single_layers * single + refiner + cond + lin(full.in_channels * 4, d) + lin(d, 2 * d) + lin(d, 4 * full.out_channels)
    assert 12.7e9 < total < 12.9e9, total
    small = hy.HunyuanVideoTransformer3DModel(hy.HunyuanVideoConfig(num_layers=1, num_single_layers=1, num_refiner_layers=1, num_attention_heads=2, attention_head_dim=128,
                                                                      text_embed_dim=64, pooled_projection_dim=32))
    d2 = 256
    mlp2 = 4 * d2
    lin2 = lambda i, o: i * o + o
    want = (2 * lin2(d2, 6 * d2) + 8 * lin2(d2, d2) + 4 * 128 + 2 * (lin2(d2, mlp2) + lin2(mlp2, d2))) + (lin2(d2, 3 * d2) + 3 * lin2(d2, d2) + 2 * 128 + lin2(d2, mlp2) + lin2(d2 + mlp2, d2)) \
        + (lin2(256, d2) + lin2(d2, d2) + lin2(64, d2) + lin2(d2, d2) + lin2(64, d2) + (2 * 2 * d2 + 4 * lin2(d2, d2) + lin2(d2, mlp2) + lin2(mlp2, d2) + lin2(d2, 2 * d2))) \
        + (2 * (lin2(256, d2) + lin2(d2, d2)) + lin2(32, d2) + lin2(d2, d2)) + lin2(16 * 4, d2) + lin2(d2, 2 * d2) + lin2(d2, 64)
    assert sum(p.numel() for p in small.parameters()) == want
