"""Full-size (BASELINE config 2: 28 blocks, batch 2, 2688 tokens, rank 64) checks through size-independent properties.

The oracle cannot run this size in seconds (minutes per sample on the host), so the production shape is
covered by properties the domain offers -- each would be broken by a mis-tiled GEMM / attention tile, a wrong tile ->
XCD map, an out-of-bounds workspace slice or a stale LoRA working copy, none of which the small parity cases can see:
  * determinism          : the forward has no atomics, so two runs give bit-identical predictions;
  * batch independence   : sample 0 of a batch of two is bit-identical to the same sample run alone (row tiles of 192 /
                           128 tokens straddle nothing observable; heads and samples never mix);
  * B = 0 LoRA           : the prediction equals the adapter-free model's bit for bit, dA is exactly zero, dB is not;
  * linearity of backward: gradients for 2 x dL/dpred are 2 x the gradients (a power of two is exact in bf16 / fp32;
                           only the fp32 atomic accumulation order of the weight-gradient GEMMs differs run to run).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def full():
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification

    dev = _dev()
    tcfg = LTXTransformerConfig(num_layers=28)
    spec = MI355XLTXVideoModelSpecification(transformer_config=tcfg)
    model = spec.load_diffusion_models(device=dev, random_init_seed=0)["transformer"]
    model.add_adapter(r=64, lora_alpha=64.0)
    g = torch.Generator(device=dev).manual_seed(3)
    with torch.no_grad():
        model.lora_flat.copy_(torch.randn(model.lora_flat.shape, generator=g, device=dev) * 0.01)
    B, C = 2, tcfg.in_channels
    latents = torch.randn((B, C, 7, 16, 24), generator=g, device=dev).to(torch.bfloat16)
    noise = torch.randn((B, C, 7, 16, 24), generator=g, device=dev).to(torch.bfloat16)
    text = torch.randn((B, 128, tcfg.caption_channels), generator=g, device=dev).to(torch.bfloat16)
    mask = torch.zeros((B, 128), dtype=torch.bfloat16, device=dev)
    mask[0, :32] = 1
    mask[1, :96] = 1
    sig = torch.tensor([0.25, 0.7], device=dev)
    return spec, model, latents, noise, text, mask, sig


def _forward(spec, model, latents, noise, text, mask, sig, rows=slice(None)):
    C = latents.shape[1]
    dev = latents.device
    return spec.forward(
        transformer=model,
        condition_model_conditions={"encoder_hidden_states": text[rows].contiguous(), "encoder_attention_mask": mask[rows].contiguous()},
        latent_model_conditions={"latents": latents[rows].contiguous(), "latents_mean": torch.zeros(C, device=dev), "latents_std": torch.ones(C, device=dev)},
        sigmas=sig[rows].contiguous(), noise=noise[rows].contiguous(), force_first_frame_branch=False,
    )


def test_full_size_forward_is_deterministic_and_batch_independent(full):
    spec, model, latents, noise, text, mask, sig = full
    with torch.no_grad():
        p1, t1, _ = _forward(spec, model, latents, noise, text, mask, sig)
        p1 = p1.clone()
        p2, _, _ = _forward(spec, model, latents, noise, text, mask, sig)
        assert p1.shape == (2, 2688, 128) and torch.isfinite(p1.float()).all()
        assert torch.equal(p1, p2), "forward is not deterministic"
        p2 = p2.clone()
        ps, ts, _ = _forward(spec, model, latents, noise, text, mask, sig, rows=slice(0, 1))
        assert torch.equal(ps[0], p2[0]), "sample 0 changes with its batch neighbour"
        assert torch.equal(ts[0], t1[0])
        # the prediction is a real function of the inputs (not a constant / zero buffer)
        assert p2.float().std() > 1e-3 and not torch.equal(p2[0], p2[1])


def test_full_size_zero_lora_b(full):
    spec, model, latents, noise, text, mask, sig = full
    saved = model.lora_flat.detach().clone()
    try:
        with torch.no_grad():
            model.lora_B.zero_()
        model.refresh_lora_copies() if hasattr(model, "refresh_lora_copies") else None
        pred, target, _ = _forward(spec, model, latents, noise, text, mask, sig)
        loss = (pred.float() - target.float()).pow(2).mean()
        loss.backward()
        torch.cuda.synchronize()
        ga, gb = model.lora_A.grad, model.lora_B.grad
        assert torch.count_nonzero(ga) == 0, "dA must vanish when B == 0"
        assert torch.isfinite(gb).all() and gb.abs().max() > 0
        # every one of the 224 adapters receives a gradient
        per_adapter = gb.reshape(28, 8, -1).abs().amax(dim=2)
        assert (per_adapter > 0).all()
        # and the prediction is the adapter-free model's
        model.lora_A.grad = None
        model.lora_B.grad = None
        with torch.no_grad():
            model.lora_A.mul_(3.0)  # any A: the LoRA branch stays exactly zero while B == 0
        model.refresh_lora_copies() if hasattr(model, "refresh_lora_copies") else None
        with torch.no_grad():
            pred2, _, _ = _forward(spec, model, latents, noise, text, mask, sig)
        assert torch.equal(pred.detach(), pred2)
    finally:
        with torch.no_grad():
            model.lora_flat.copy_(saved)
        model.lora_A.grad = None
        model.lora_B.grad = None
        if hasattr(model, "_lora_versions"):
            model._lora_versions = None


def test_full_size_backward_is_linear_in_dpred(full):
    spec, model, latents, noise, text, mask, sig = full
    g = torch.Generator(device=_dev()).manual_seed(9)
    w = torch.randn((2, 2688, 128), generator=g, device=_dev()) * 1e-3
    grads = []
    for scale in (1.0, 2.0):
        model.lora_A.grad = None
        model.lora_B.grad = None
        pred, _, _ = _forward(spec, model, latents, noise, text, mask, sig)
        (pred.float() * (w * scale)).sum().backward()
        torch.cuda.synchronize()
        grads.append((model.lora_A.grad.detach().clone(), model.lora_B.grad.detach().clone()))
    model.lora_A.grad = None
    model.lora_B.grad = None
    for g1, g2, name in ((grads[0][0], grads[1][0], "dA"), (grads[0][1], grads[1][1], "dB")):
        assert torch.isfinite(g1).all() and g1.abs().max() > 0
        rel = ((g2 - 2.0 * g1).float().norm() / (2.0 * g1).float().norm()).item()
        print(f"[linearity] {name}: rel {rel:.3e}")
        assert rel < 1e-5, f"{name} is not linear in dL/dpred ({rel:.3e})"
