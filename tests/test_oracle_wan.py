"""oracle/wan.py (the row after CogVideoX) against fixtures produced by executing the reference's own Wan code
(oracle/make_golden.py: WanModelSpecification.forward, _normalize_latents, DiagonalGaussianDistribution, the patched time/text embedding)."""

import pytest
import torch

from oracle import wan


@pytest.fixture(scope="module")
def golden():
    import os

    from safetensors.torch import load_file

    return load_file(os.path.join(os.path.dirname(__file__), "golden", "reference_fixtures.safetensors"))


def _model():
    return wan.build_model(wan.WanConfig.dummy(), seed=0, dtype=torch.bfloat16)


def test_spec_forward_matches_the_reference(golden):
    """Stored moments -> normalise mean AND log-variance (x latents_std) -> sample -> flow-match mix -> DiT -> (pred, noise - latents): bit for bit what
    the reference's forward returned for the same generator."""
    m = _model()
    mom, text, sig = golden["wan.spec.moments"], golden["wan.spec.text"], golden["wan.spec.sigmas"].view(-1, 1, 1, 1, 1)
    g = torch.Generator().manual_seed(77)  # the reference draws the posterior eps first, then the flow-match noise, from one generator
    eps = torch.randn(mom.shape[0], mom.shape[1] // 2, *mom.shape[2:], generator=g, dtype=mom.dtype)
    noise = torch.zeros_like(eps).normal_(generator=g)
    with torch.no_grad():
        pred, target, _ = wan.spec_forward(m, mom, golden["wan.spec.latents_mean"], golden["wan.spec.latents_std"], text, sig, eps, noise)
    assert torch.equal(target, golden["wan.spec.target"])
    assert torch.equal(pred, golden["wan.spec.pred"])


def test_patched_time_text_embedding(golden):
    m = _model()
    with torch.no_grad():
        temb, tproj, text, _ = m.condition_embedder(torch.tensor([310, 840]), golden["wan.spec.text"])
    assert torch.equal(temb, golden["wan.embed.temb"]) and torch.equal(tproj, golden["wan.embed.timestep_proj"]) and torch.equal(text, golden["wan.embed.text"])
    assert temb.dtype == torch.bfloat16  # the patch casts the sinusoidal projection to the text dtype


def test_structure_and_full_finetune_gradients():
    """Wan2.1-T2V-1.3B parameter count [upstream], and every parameter receives a gradient (config 4 is a FULL fine-tune)."""
    cfg = wan.WanConfig()
    d, f = cfg.inner_dim, cfg.ffn_dim
    per_block = 2 * (4 * (d * d + d) + 2 * d) + (d * f + f) + (f * d + d) + 2 * d + 6 * d
    total = (per_block * cfg.num_layers + (cfg.in_channels * 4 * d + d) + (cfg.freq_dim * d + d + d * d + d) + (d * 6 * d + 6 * d)
             + (cfg.text_dim * d + d + d * d + d) + (d * cfg.out_channels * 4 + cfg.out_channels * 4) + 2 * d)
    assert total == 1_418_996_800, total
    m = wan.build_model(wan.WanConfig.dummy(), dtype=torch.float32)
    x, t, txt = torch.randn(1, 16, 3, 4, 6), torch.tensor([500]), torch.randn(1, 5, 32)
    out = m(x, t, txt, return_dict=False)[0]
    assert out.shape == x.shape
    out.square().mean().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    r = m.rope(x)
    assert r.shape == (1, 1, 3 * 2 * 3, 6) and r.dtype == torch.complex128
