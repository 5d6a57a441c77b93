"""Data-parallel step on the GPU: the block-range backward behind the bucketed gradient exchange, the exchange itself on RCCL (a
one-rank communicator on a single-GPU box; two real ranks when two MI355X are visible), gradient accumulation.  pytest -m gpu."""

import os

import pytest
import torch

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16


def _dev():
    return torch.device("cuda", 0)


def _model_and_batch(num_layers, B, F_, H_, W_, seed=3, dev=None, data_seed=None):
    """Same weights for every caller (CPU generator, fixed seed); data from `data_seed`."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification

    dev = dev or _dev()
    spec = MI355XLTXVideoModelSpecification(transformer_config=LTXTransformerConfig(num_layers=num_layers))
    model = spec.load_diffusion_models(device=dev, random_init_seed=0)["transformer"]
    model.add_adapter(r=64, lora_alpha=64)
    with torch.no_grad():
        g = torch.Generator().manual_seed(seed)
        model.lora_flat.copy_((torch.randn(model.lora_flat.shape, generator=g) * 0.02).to(dev))
    gd = torch.Generator().manual_seed(100 if data_seed is None else data_seed)
    lat = torch.randn((B, 128, F_, H_, W_), generator=gd).to(bf16).to(dev)
    text = torch.randn((B, 128, 4096), generator=gd).to(bf16).to(dev)
    noise = torch.randn((B, 128, F_, H_, W_), generator=gd).to(bf16).to(dev)
    mask = torch.zeros((B, 128), dtype=bf16, device=dev)
    for i in range(B):
        mask[i, : (32 if i % 2 == 0 else 96)] = 1
    cond = {"encoder_hidden_states": text, "encoder_attention_mask": mask}
    latd = {"latents": lat, "latents_mean": torch.zeros(128, device=dev), "latents_std": torch.ones(128, device=dev)}
    sig = torch.tensor([0.25, 0.7, 0.4, 0.9][:B], device=dev)
    return spec, model, cond, latd, sig, noise


def _grads(spec, model, cond, latd, sig, noise, hook=None, bucket=0):
    model.lora_A.grad = None
    model.lora_B.grad = None
    model._grad_bucket_hook, model.grad_bucket_blocks = hook, bucket
    try:
        pred, target, _ = spec.forward(transformer=model, condition_model_conditions=dict(cond), latent_model_conditions=dict(latd), sigmas=sig,
                                       noise=noise, force_first_frame_branch=False)
        ((pred.float() - target.float()) ** 2).mean().backward()
        torch.cuda.synchronize()
    finally:
        model._grad_bucket_hook = None
    return model.lora_A.grad.detach().clone(), model.lora_B.grad.detach().clone()


def test_gradient_checkpointing_gives_the_same_gradients_from_one_block_slot():
    """--gradient_checkpointing (the reference's own LTX example sets it: examples/training/sft/ltx_video/crush_smol_lora/train.sh:79; trainer.py:155-157 ->
    utils/activation_checkpoint.py:24-49 wraps every block).  Here the workspace then holds ONE block slot instead of L, the forward keeps the residual stream
    only and the backward re-runs a block's forward kernels before its gradient kernels: the prediction is bit-identical, the LoRA gradients agree to the
    fp32-atomics order of the weight-gradient GEMMs (batched over the range without checkpointing, per block with it), the workspace is the smaller one, and
    the block-range backward (bucketed exchange) works on top of it."""
    import ctypes

    from finetrainers_amd import _lib

    spec, model, cond, latd, sig, noise = _model_and_batch(5, 2, 2, 4, 6)
    ga0, gb0 = _grads(spec, model, cond, latd, sig, noise)
    pred0, _, _ = spec.forward(transformer=model, condition_model_conditions=dict(cond), latent_model_conditions=dict(latd), sigmas=sig, noise=noise,
                               force_first_frame_branch=False)
    full = _lib.load().ftmi_ltx_workspace_bytes(ctypes.byref(model._c_config(2, 2688, 128)))  # (sized at config 2's token count; this model has 5 blocks)
    one = _lib.load().ftmi_ltx_workspace_bytes(ctypes.byref(model._c_config(2, 2688, 128, checkpoint=True)))
    print(f"[ckpt] workspace at 2 x 2688 tokens, 5 blocks: {full / 2**30:.2f} GiB kept, {one / 2**30:.2f} GiB checkpointed")
    assert one < 0.5 * full, (one, full)  # 5 block slots -> 1 (the rest: the residual stream, the all-block text-side arrays, the backward's scratch)
    model.enable_gradient_checkpointing()
    assert model.is_gradient_checkpointing
    ga1, gb1 = _grads(spec, model, cond, latd, sig, noise)
    pred1, _, _ = spec.forward(transformer=model, condition_model_conditions=dict(cond), latent_model_conditions=dict(latd), sigmas=sig, noise=noise,
                               force_first_frame_branch=False)
    assert torch.equal(pred0, pred1)
    for name, a, b in (("A", ga0, ga1), ("B", gb0, gb1)):
        rel = ((a - b).norm() / a.norm()).item()
        print(f"[ckpt] lora_{name}.grad checkpointed vs kept: rel {rel:.2e}")
        assert rel < 2e-6, (name, rel)
    seen = []
    ga2, gb2 = _grads(spec, model, cond, latd, sig, noise, hook=lambda lo, hi, a, b: seen.append((lo, hi)), bucket=2)
    assert seen == [(3, 5), (1, 3), (0, 1)]
    assert ((ga2 - ga1).norm() / ga1.norm()).item() < 2e-6 and ((gb2 - gb1).norm() / gb1.norm()).item() < 2e-6
    model.disable_gradient_checkpointing()
    with pytest.raises(ValueError):
        model.apply_activation_checkpointing("block_skip", 2)


def test_fused_down_projection_launch_gives_the_bits_of_the_two_launches():
    """Round 6 (FTMI_FUSE_DOWN=1): x A^T / dY B computed by the leading workgroups of the projection GEMM's own launch, the GEMM's tiles waiting on per-row-tile
    counters two K stages before their K-extension.  Same kernels' arithmetic, so with 192- / 256-row tiles on both sides the prediction and every activation
    gradient are THE SAME BITS as with the two launches; the single-round launches move from the 192 x 128 kernel to the 192 x 256 pipeline when fused (bit-
    identical GEMM kernels by construction), so the LoRA gradients agree to the fp32 atomic order of the weight-gradient GEMMs.  No wait may time out."""
    import ctypes

    from finetrainers_amd import _lib

    lib = _lib.load()
    if not hasattr(lib, "ftmi_gemm_sk_status"):  # (an entry point only the FTMI_EXPERIMENTAL build exports)
        pytest.skip("the fused down-projection launch is compiled in FTMI_EXPERIMENTAL builds only (measured break-even: profiles/r06_instep_ab_fused_down_*.txt)")
    spec, model, cond, latd, sig, noise = _model_and_batch(3, 2, 7, 16, 24)  # config 2's token count: the fused launch needs >= 84 row tiles x column tiles
    res = {}
    for fuse in ("0", "3"):
        os.environ["FTMI_FUSE_DOWN"] = fuse
        lib.ftmi_reload_switches()
        try:
            ga, gb = _grads(spec, model, cond, latd, sig, noise)
            pred, _, _ = spec.forward(transformer=model, condition_model_conditions=dict(cond), latent_model_conditions=dict(latd), sigmas=sig, noise=noise,
                                      force_first_frame_branch=False)
            res[fuse] = (pred.detach().clone(), ga, gb)
        finally:
            os.environ.pop("FTMI_FUSE_DOWN", None)
            lib.ftmi_reload_switches()
    assert lib.ftmi_fused_status() == 0, "a fused launch gave up waiting for its down-projection"
    assert torch.equal(res["0"][0], res["3"][0]), f"prediction differs: {(res['0'][0].float() - res['3'][0].float()).abs().max().item():.3e}"
    for name, a, b in (("A", res["0"][1], res["3"][1]), ("B", res["0"][2], res["3"][2])):
        rel = ((a - b).norm() / a.norm()).item()
        print(f"[fused] lora_{name}.grad fused vs two launches: rel {rel:.2e}")
        assert rel < 2e-6, (name, rel)


def test_block_range_backward_matches_single_call():
    """ftmi_ltx_backward_range over [3,5) [1,3) [0,1) == one ftmi_ltx_backward over [0,5): same gradients (fp32 atomics order aside),
    ranges reported in backward order, each exactly once."""
    spec, model, cond, latd, sig, noise = _model_and_batch(5, 2, 2, 4, 6)
    ga0, gb0 = _grads(spec, model, cond, latd, sig, noise)
    seen = []

    def hook(lo, hi, ga, gb):
        assert ga.shape[0] == hi - lo and gb.shape[0] == hi - lo
        seen.append((lo, hi))

    ga1, gb1 = _grads(spec, model, cond, latd, sig, noise, hook=hook, bucket=2)
    assert seen == [(3, 5), (1, 3), (0, 1)]
    for a, b, n in ((ga1, ga0, "dA"), (gb1, gb0, "dB")):
        rel = ((a - b).norm() / b.norm()).item()
        print(f"[ranges] {n} rel {rel:.2e}")
        assert rel < 1e-5 and torch.isfinite(a).all()
    assert (ga0[:, 5:7].abs().amax(dim=(1, 2, 3)) > 0).all()  # the text-side adapters (attn2.to_k / to_v) of every block got gradients


def test_dp_step_on_rccl_single_rank():
    """The multi-GPU step's code path on one GPU: a one-rank RCCL communicator, LoRA broadcast, bucketed ReduceOp.AVG all-reduces issued
    from the backward on RCCL's stream, clip + AdamW after the join.  Must reproduce the non-distributed step."""
    from finetrainers_amd.parallel import DataParallelBackend
    from finetrainers_amd.trainer import MI355XSFTStep

    os.environ.setdefault("MASTER_PORT", str(29900 + os.getpid() % 90))
    par = DataParallelBackend(backend="nccl", exercise_collectives=True)
    try:
        assert par.active and par.world_size == 1
        outs = []
        for use_par in (False, True):
            spec, model, cond, latd, sig, noise = _model_and_batch(4, 2, 2, 4, 4)
            step = MI355XSFTStep(model, spec, lr=5e-5, betas=(0.9, 0.99), parallel=par if use_par else None, grad_bucket_blocks=1)
            first = None
            for _ in range(2):
                o = step.step(cond, latd, sigmas=sig, noise=noise, force_first_frame_branch=False)
                first = first or (o["loss"].item(), o["grad_norm"].item())
            torch.cuda.synchronize()
            outs.append((first, o["loss"].item(), o["grad_norm"].item(), model.lora_flat.detach().clone(), step.reducer))
        (f0, l0, g0, p0, _), (f1, l1, g1, p1, red) = outs
        print(f"[dp-1rank] step 1 loss {f0[0]:.6f} / {f1[0]:.6f} grad_norm {f0[1]:.6e} / {f1[1]:.6e};  step 2 loss {l0:.6f} / {l1:.6f} grad_norm {g0:.6e} / {g1:.6e}  buckets {red.buckets_issued}")
        assert red is not None and red.buckets_issued == 8  # 4 buckets per step x 2 steps
        # step 1: same weights, same data -> the same loss up to the order of its fp32 atomic sum; gradients differ only by the fp32 atomic order of the per-bucket
        # weight-gradient launches.  Step 2 starts from parameters that differ where AdamW's sign-like first update saw a near-zero
        # gradient with the other sign (an update of +-lr either way), so it agrees to ~1e-4, not to rounding.
        assert abs(f0[0] - f1[0]) <= 1e-6 * abs(f0[0]) and abs(f0[1] - f1[1]) <= 1e-5 * f0[1]  # (the loss scalar is an atomic fp32 sum)
        assert abs(l0 - l1) <= 2e-4 * abs(l0) and abs(g0 - g1) <= 1e-3 * g0
        assert ((p0 - p1).norm() / p0.norm()).item() < 1e-4
        m = par.reduce_step_metrics(o["loss"], o["grad_norm"])
        assert abs(m["global_avg_loss"].item() - l1) < 1e-7
    finally:
        par.destroy()


def test_dp_step_through_the_in_library_exchange_single_rank(monkeypatch):
    """SURVEY 8(b)'s ftmi_allreduce_{init,bucket,wait}: the gradient exchange of replicate(bucket_cap_mb=100) (parallel/ptd.py:462-463) INSIDE the library --
    its own RCCL communicator (dlopen), its own communication stream, event hand-over with the compute stream.  On this one-GPU box a one-rank communicator:
    the step that sends its buckets through ftmi_allreduce_bucket / _wait must reproduce the step that sends them through torch.distributed, bucket count
    included, and a mean all-reduce over one rank must leave the data untouched."""
    import ctypes

    from finetrainers_amd import _lib
    from finetrainers_amd.parallel import DataParallelBackend, GradBucketReducer
    from finetrainers_amd.trainer import MI355XSFTStep

    lib = _lib.load()
    assert lib.ftmi_allreduce_version() > 20000, "librccl not found by dlopen"
    os.environ.setdefault("MASTER_PORT", str(29800 + os.getpid() % 90))
    par = DataParallelBackend(backend="nccl", exercise_collectives=True)
    try:
        ex = par.native_exchange()
        assert par.native_exchange() is ex  # cached
        buf = torch.randn(1 << 20, device=_dev())
        ref = buf.clone()
        st = torch.cuda.current_stream().cuda_stream
        assert lib.ftmi_allreduce_bucket(ex, buf.data_ptr(), buf.numel(), 1, st) == 0
        assert lib.ftmi_allreduce_wait(ex, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(buf, ref) and lib.ftmi_allreduce_buckets_issued(ex) == 1
        outs = []
        for native in (False, True):
            spec, model, cond, latd, sig, noise = _model_and_batch(4, 2, 2, 4, 4)
            step = MI355XSFTStep(model, spec, lr=5e-5, betas=(0.9, 0.99), parallel=par, grad_bucket_blocks=1)
            if native:
                step.reducer = GradBucketReducer(par, native=True)
                step.reducer.measure_exposed = True
            for _ in range(2):
                o = step.step(cond, latd, sigmas=sig, noise=noise, force_first_frame_branch=False)
            torch.cuda.synchronize()
            outs.append((o["loss"].item(), o["grad_norm"].item(), model.lora_flat.detach().clone(), step.reducer))
        (l0, g0, p0, r0), (l1, g1, p1, r1) = outs
        print(f"[dp-native] loss {l0:.6f} / {l1:.6f} grad_norm {g0:.6e} / {g1:.6e} buckets {r0.buckets_issued} / {r1.buckets_issued} exposed {r1.exposed_ms()}")
        assert r1.native and r0.buckets_issued == r1.buckets_issued == 8
        assert lib.ftmi_allreduce_buckets_issued(ex) == 1 + 2 * 8  # two slices (A, B) per bucket
        assert abs(l0 - l1) <= 2e-4 * abs(l0) and abs(g0 - g1) <= 1e-3 * g0 and ((p0 - p1).norm() / p0.norm()).item() < 1e-4
    finally:
        par.destroy()


def test_gradient_accumulation_matches_oracle():
    """gradient_accumulation_steps = 2 (trainer.py:476-503): two micro-batches, each loss / 2, gradients summed, ONE clip + AdamW.  The
    accumulated LoRA gradient is compared with the oracle's; the scale must be 1/gas, not 1/gas^2."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification
    from finetrainers_amd.trainer import MI355XSFTStep, sft_loss
    from oracle import ltx

    dev = _dev()
    cfg = ltx.LTXConfig.production(num_layers=1)
    omodel = ltx.build_model(cfg, seed=0, rank=64, alpha=64.0, lora_b_std=0.02)
    micro = [ltx.synth_inputs(cfg, 1, 2, 4, 4, seed=31 + i, mask_lens=[32 + 64 * i], sigmas=[0.25 + 0.45 * i]) for i in range(2)]
    for p in omodel.parameters():
        p.grad = None
    losses_ref = []
    for inp in micro:
        loss, _, _ = ltx.forward_loss(omodel, inp, contiguous_hidden_states=True)
        (loss / 2).backward()
        losses_ref.append(loss.item() / 2)
    g_ref = {n.replace(".default", ""): p.grad.detach().clone() for n, p in ltx.lora_parameters(omodel)}
    gn_ref = torch.linalg.vector_norm(torch.stack([g.norm() for g in g_ref.values()])).item()

    spec = MI355XLTXVideoModelSpecification(transformer_config=LTXTransformerConfig(num_layers=1))
    gmodel = spec.load_diffusion_models(state_dict=omodel.state_dict(), device=dev)["transformer"]
    gmodel.add_adapter(r=64, lora_alpha=64)
    gmodel.load_state_dict({k: v for k, v in omodel.state_dict().items() if "lora_" in k}, strict=False)

    def fwd(inp):
        return spec.forward(
            transformer=gmodel,
            condition_model_conditions={"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)},
            latent_model_conditions={"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std},
            sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False)

    # (a) the reference loop's form: sft_loss(..., gradient_accumulation_steps=2).backward() twice
    for i, inp in enumerate(micro):
        pred, target, sig = fwd(inp)
        loss = sft_loss(pred, target, sig, "none", gradient_accumulation_steps=2)
        loss.backward()
        assert abs(loss.item() - losses_ref[i]) <= 1e-3 * abs(losses_ref[i])
    torch.cuda.synchronize()
    glob, worst = ltx.grads_rel_l2({k: v.float().cpu() for k, v in gmodel.lora_grad_views().items()}, g_ref)
    print(f"[accumulate] sft_loss x2: grad rel_l2 {glob:.3e} (worst adapter {worst:.3e}); |g| oracle {gn_ref:.4e}")
    assert glob < 1.5e-2  # 1/gas^2 instead of 1/gas would be 0.5
    gmodel.lora_A.grad = None
    gmodel.lora_B.grad = None

    # (b) the fused step: micro-step 1 accumulates only, micro-step 2 clips and steps
    step = MI355XSFTStep(gmodel, spec, lr=5e-5, betas=(0.9, 0.99), max_grad_norm=1.0, gradient_accumulation_steps=2)
    before = gmodel.lora_flat.detach().clone()
    outs = []
    for inp in micro:
        outs.append(step.step({"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)},
                              {"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std},
                              sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False))
        if len(outs) == 1:
            assert outs[0]["grad_norm"] is None and torch.equal(gmodel.lora_flat, before) and step.step_count == 0
    torch.cuda.synchronize()
    gn = outs[1]["grad_norm"].item()
    print(f"[accumulate] fused step: grad_norm {gn:.5e} vs oracle {gn_ref:.5e}; losses {[o['loss'].item() for o in outs]} vs {losses_ref}")
    assert abs(gn - gn_ref) <= 5e-3 * gn_ref
    assert step.step_count == 1 and not torch.equal(gmodel.lora_flat, before) and gmodel.lora_A.grad is None


def test_gradient_accumulation_at_config2_clip_size():
    """The same accumulation window at BASELINE config 2's clip size (49x512x768 -> latents [1, 128, 7, 16, 24] = 2 688 tokens per micro-batch), 4 of the 28
    blocks, micro-batches with different text lengths and sigmas: the accumulated LoRA gradient after two micro-steps of the fused step against the oracle's
    two backward passes.  At this size the summation-order floor of the graph is ~1.1e-3 (profiles/r04_parity.txt); asserted: 2.5e-3."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification
    from finetrainers_amd.trainer import MI355XSFTStep
    from oracle import ltx

    dev = _dev()
    L = 4
    cfg = ltx.LTXConfig.production(num_layers=L)
    omodel = ltx.build_model(cfg, seed=0, rank=64, alpha=64.0, lora_b_std=0.02)
    micro = [ltx.synth_inputs(cfg, 1, 7, 16, 24, seed=41 + i, mask_lens=[32 + 64 * i], sigmas=[0.7 - 0.45 * i]) for i in range(2)]
    for p in omodel.parameters():
        p.grad = None
    for inp in micro:
        loss, _, _ = ltx.forward_loss(omodel, inp, contiguous_hidden_states=True)
        (loss / 2).backward()
    g_ref = {n.replace(".default", ""): p.grad.detach().clone() for n, p in ltx.lora_parameters(omodel)}
    gn_ref = torch.linalg.vector_norm(torch.stack([g.norm() for g in g_ref.values()])).item()

    spec = MI355XLTXVideoModelSpecification(transformer_config=LTXTransformerConfig(num_layers=L))
    gmodel = spec.load_diffusion_models(state_dict=omodel.state_dict(), device=dev)["transformer"]
    gmodel.add_adapter(r=64, lora_alpha=64)
    gmodel.load_state_dict({k: v for k, v in omodel.state_dict().items() if "lora_" in k}, strict=False)
    step = MI355XSFTStep(gmodel, spec, lr=0.0, max_grad_norm=1e9, gradient_accumulation_steps=2)  # lr 0, no clipping: .grad after the window is the plain sum
    outs = []
    for inp in micro:
        outs.append(step.step({"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)},
                              {"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std},
                              sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False))
    torch.cuda.synchronize()
    got = gmodel._grad_flat.detach().clone()
    n = gmodel.lora_A.numel()
    views = {}
    A = got[:n].view_as(gmodel.lora_A).float().cpu()
    Bm = got[n:n + gmodel.lora_B.numel()].view_as(gmodel.lora_B).float().cpu()
    names = ("attn1.to_q", "attn1.to_k", "attn1.to_v", "attn1.to_out.0", "attn2.to_q", "attn2.to_k", "attn2.to_v", "attn2.to_out.0")
    for l in range(L):
        for j, nm in enumerate(names):
            views[f"transformer_blocks.{l}.{nm}.lora_A.weight"] = A[l, j]
            views[f"transformer_blocks.{l}.{nm}.lora_B.weight"] = Bm[l, j]
    assert set(views) == set(g_ref)
    glob, worst = ltx.grads_rel_l2(views, g_ref)
    gn = outs[1]["grad_norm"].item()
    print(f"[accumulate-cfg2] {L} blocks, 2 x 2688 tokens: accumulated LoRA-grad rel_l2 {glob:.3e} (worst adapter {worst:.3e}); grad_norm {gn:.5e} vs oracle {gn_ref:.5e}")
    assert glob < 2.5e-3 and worst < 1.2e-2
    assert abs(gn - gn_ref) <= 2e-3 * gn_ref


def test_accumulation_clips_after_every_backward():
    """The reference clips after every backward (trainer.py:487-492), also on the micro-steps of an accumulation window: with a bound the
    partial sum exceeds, .grad is rescaled in place after micro-step 1, and the norm reported at the stepping micro-step is that of
    clip(g/2) + g/2, not of g."""
    from finetrainers_amd.trainer import MI355XSFTStep

    spec, model, cond, latd, sig, noise = _model_and_batch(2, 1, 2, 4, 4, seed=3)
    kw = dict(sigmas=sig, noise=noise, force_first_frame_branch=False)
    raw_step = MI355XSFTStep(model, spec, lr=0.0, max_grad_norm=1e9, gradient_accumulation_steps=2)
    raw_step.step(cond, latd, **kw)
    raw = model._grad_flat.clone()  # g / 2, unclipped
    raw_norm = raw.norm().item()
    model.lora_A.grad = model.lora_B.grad = None

    bound = 0.25 * raw_norm
    step = MI355XSFTStep(model, spec, lr=0.0, max_grad_norm=bound, gradient_accumulation_steps=2)
    o1 = step.step(cond, latd, **kw)
    torch.cuda.synchronize()
    assert o1["grad_norm"] is None and step.step_count == 0
    coef = bound / (raw_norm + 1e-6)
    after1 = model._grad_flat.clone()
    assert abs(after1.norm().item() - coef * raw_norm) <= 1e-5 * bound  # rescaled in place to the bound
    o2 = step.step(cond, latd, **kw)  # same batch, lr = 0: the second backward adds the same g / 2
    torch.cuda.synchronize()
    want = (1.0 + coef) * raw_norm
    got = o2["grad_norm"].item()
    print(f"[accumulate-clip] reported norm {got:.6e} vs clip(g/2) + g/2 = {want:.6e} (raw g/2 norm {raw_norm:.6e})")
    assert abs(got - want) <= 2e-3 * want and step.step_count == 1


def _two_rank_worker(rank, world, port, q, backend="nccl", one_gpu=False):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(0 if one_gpu else rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch

    from finetrainers_amd.parallel import DataParallelBackend
    from finetrainers_amd.trainer import MI355XSFTStep

    par = DataParallelBackend(backend=backend, device=torch.device("cuda", 0) if one_gpu else None)
    try:
        spec, model, cond, latd, sig, noise = _model_and_batch(4, 1, 2, 4, 4, seed=3 + rank, dev=par.device, data_seed=200 + rank)  # different LoRA init per rank
        step = MI355XSFTStep(model, spec, lr=5e-5, betas=(0.9, 0.99), parallel=par, grad_bucket_blocks=2)  # rank 0's adapter is broadcast
        o = step.step(cond, latd, sigmas=sig, noise=noise, force_first_frame_branch=False)
        torch.cuda.synchronize()
        q.put((rank, o["loss"].item(), o["grad_norm"].item(), model.lora_flat.detach().cpu().numpy(), step.reducer.buckets_issued))  # numpy: pickled by value (a torch CPU tensor travels as a file descriptor of a process that may be gone)
    finally:
        par.destroy()


def _run_two_ranks(backend, one_gpu):
    """Two ranks, same weights, different data: after one step both hold the parameters a single rank gets from the concatenated batch."""
    import torch.multiprocessing as mp

    from finetrainers_amd.trainer import MI355XSFTStep

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200 + (50 if one_gpu else 0)
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, q, backend, one_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, l0, g0, p0, nb0), (_, l1, g1, p1, nb1) = res
    p0, p1 = torch.from_numpy(p0), torch.from_numpy(p1)
    assert torch.equal(p0, p1) and g0 == g1  # replicas stay bit-identical (the norm reduction is order-fixed)
    assert nb0 == nb1 == 2  # 4 blocks in buckets of 2: two bucketed exchanges issued from inside the backward
    # single rank, both samples in one batch
    spec, model, cond0, lat0, sig0, n0 = _model_and_batch(4, 1, 2, 4, 4, seed=3, data_seed=200)
    _, _, cond1, lat1, sig1, n1 = _model_and_batch(4, 1, 2, 4, 4, seed=3, data_seed=201)
    cond = {k: torch.cat([cond0[k], cond1[k]]) for k in cond0}
    latd = dict(lat0, latents=torch.cat([lat0["latents"], lat1["latents"]]))
    step = MI355XSFTStep(model, spec, lr=5e-5, betas=(0.9, 0.99))
    before = model.lora_flat.detach().cpu().clone()
    o = step.step(cond, latd, sigmas=torch.cat([sig0, sig1]), noise=torch.cat([n0, n1]), force_first_frame_branch=False)
    torch.cuda.synchronize()
    assert abs(o["loss"].item() - (l0 + l1) / 2) < 1e-5 * abs(o["loss"].item())
    assert abs(o["grad_norm"].item() - g0) < 1e-3 * g0
    # the first AdamW step moves every entry by ~lr * sign(g): entries whose tiny gradient changes sign between the two summation orders
    # differ by 2 lr, so compare the UPDATES (relative to the update's own norm), not the parameters to 1e-5
    upd_ref, upd = model.lora_flat.cpu() - before, p0 - before
    rel = ((upd - upd_ref).norm() / upd_ref.norm()).item()
    print(f"[dp-2rank {backend}{' one GPU' if one_gpu else ''}] update rel diff vs concatenated batch: {rel:.2e}; grad_norm {g0:.5e} vs {o['grad_norm'].item():.5e}")
    assert rel < 2e-2


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two MI355X (the round-end driver's multi-GPU node); single-GPU boxes run the gloo variant below")
def test_dp_step_two_ranks_equals_concatenated_batch():
    _run_two_ranks("nccl", one_gpu=False)


def test_dp_step_two_ranks_sharing_one_gpu_gloo():
    """The same world-size-2 step on ONE MI355X: two processes share cuda:0 and exchange through gloo (RCCL refuses two ranks on one
    device).  Everything but the transport is the product path: LoRA broadcast, block-range backward, bucket hooks, async handles,
    finish(), metric reduction."""
    _run_two_ranks("gloo", one_gpu=True)


def test_unmodified_trainer_loop_drives_the_drop_in_classes():
    """The way SFTTrainer._train calls the plug-ins (finetrainers/trainer/sft_trainer/trainer.py:436-528, restated statement by statement with the
    reference's own names -- /root/reference does not exist on the GPU box): MI355XParallelBackend.apply_ddp on the transformer,
    model_specification.forward(transformer=, scheduler=, condition_model_conditions=, latent_model_conditions=, sigmas=, compute_posterior=),
    the torch loss, loss.backward(), clip over transformer.parameters(), a torch AdamW(fused=False) built over ALL parameters, optimizer.step /
    lr_scheduler.step / zero_grad, the backend's properties and log().  Three steps must give the losses and gradient norms of MI355XSFTStep
    (the fused path) on the same inputs: same kernels up to the clip, torch's clip + AdamW against the fused kernel after it."""
    from finetrainers_amd.parallel import MI355XParallelBackend
    from finetrainers_amd.trainer import MI355XSFTStep
    from finetrainers_amd.utils import diffusion as diffusion_utils

    steps, lr, max_grad_norm = 3, 1e-3, 1.0
    gen_data = torch.Generator().manual_seed(7)
    noises = [torch.randn((2, 128, 2, 4, 6), generator=gen_data).to(bf16).to(_dev()) for _ in range(steps)]
    sigs = [torch.tensor([0.3 + 0.1 * i, 0.8 - 0.1 * i], device=_dev()) for i in range(steps)]

    # ---- (a) the fused MI355X step --------------------------------------------------------------------------------------------------------
    spec, model, cond, latd, _, _ = _model_and_batch(3, 2, 2, 4, 6)
    stepper = MI355XSFTStep(model, spec, lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4, max_grad_norm=max_grad_norm)
    fused = []
    for i in range(steps):
        out = stepper.step(dict(cond), dict(latd), sigmas=sigs[i], noise=noises[i], force_first_frame_branch=False)
        fused.append((float(out["loss"]), float(out["grad_norm"])))
    fused_params = model.lora_flat.detach().clone()

    # ---- (b) the reference loop, its statements in its order -------------------------------------------------------------------------------
    spec, model, cond, latd, _, _ = _model_and_batch(3, 2, 2, 4, 6)
    os.environ.setdefault("MASTER_PORT", "29533")
    parallel_backend = MI355XParallelBackend(world_size=1, dp_degree=1, backend="nccl", exercise_collectives=True)  # a one-rank RCCL communicator
    try:
        transformer = parallel_backend.apply_ddp(model, parallel_backend.get_mesh())  # trainer.py:185-189
        assert transformer._grad_bucket_hook is not None and parallel_backend.reducer is not None
        model_parts = [transformer]
        optimizer = torch.optim.AdamW([p for m in model_parts for p in m.parameters()], lr=lr, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-4, fused=False)
        lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lambda s: 1.0)
        optimizer, lr_scheduler = parallel_backend.prepare_optimizer(optimizer, lr_scheduler)  # trainer.py:228
        parallel_backend.initialize_trackers([], "t", {}, ".")
        scheduler = spec.load_diffusion_models(device=_dev(), random_init_seed=0)["scheduler"] if False else type("S", (), {"config": type("C", (), {"num_train_timesteps": 1000})()})()
        device, dtype = _dev(), bf16
        loop = []
        for i in range(steps):
            sigmas = sigs[i]  # trainer.py:436-448 (prepare_sigmas is pinned to the reference's fixtures in tests/test_host.py)
            sigmas = sigmas.reshape(-1, 1, 1, 1, 1)  # utils.expand_tensor_dims(sigmas, latents.ndim)
            pred, target, sigmas = spec.forward(transformer=transformer, scheduler=scheduler, condition_model_conditions=dict(cond),
                                                latent_model_conditions=dict(latd), sigmas=sigmas, compute_posterior=True,
                                                noise=noises[i], force_first_frame_branch=False)  # trainer.py:452-461 (+ the parity hooks)
            weights = diffusion_utils.compute_loss_weighting_for_sd3("none", sigmas)  # utils.prepare_loss_weights
            while weights.ndim < pred.ndim:
                weights = weights.unsqueeze(-1)
            loss = weights.float() * (pred.float() - target.float()).pow(2)  # trainer.py:473-481
            loss = loss.mean(list(range(1, loss.ndim)))
            loss = loss.mean()
            loss.backward()
            accumulated_loss = loss.detach().item()
            grad_norm = torch.nn.utils.clip_grad_norm_([p for m in model_parts for p in m.parameters()], max_grad_norm, foreach=True)  # trainer.py:487-492
            optimizer.step()  # trainer.py:498-503
            lr_scheduler.step()
            optimizer.zero_grad()
            # (nothing tells the transformer that torch changed the adapters: refresh_lora_copies() sees the parameters' _version move, as it must in the
            #  unmodified SFTTrainer, which never writes a private attribute)
            grad_norm = grad_norm.detach().item()
            logs = {"train/global_avg_loss": accumulated_loss, "train/global_max_loss": accumulated_loss, "train/grad_norm": grad_norm}
            assert not (parallel_backend.data_replication_enabled or parallel_backend.data_sharding_enabled or parallel_backend.context_parallel_enabled)
            parallel_backend.log(logs, step=i + 1)  # trainer.py:548
            loop.append((accumulated_loss, grad_norm))
        assert parallel_backend.reducer.buckets_issued == steps * 1 and not parallel_backend.reducer._pending  # 3 blocks < 7: one bucket per backward, ended by it
        assert [s for s, _ in parallel_backend.tracker.records] == [1, 2, 3]
        loop_params = transformer.lora_flat.detach().clone()
    finally:
        parallel_backend.destroy()
    for i, ((lf, gf), (ll, gl)) in enumerate(zip(fused, loop)):
        print(f"[trainer-loop] step {i + 1}: loss fused {lf:.6f} loop {ll:.6f} | grad norm fused {gf:.6f} loop {gl:.6f}")
        # step 1 is bit-identical (same kernels up to the clip); afterwards torch's AdamW and the fused kernel round differently in the last fp32 bit,
        # which the bf16 working copies of the adapters turn into ~1e-5 of the loss
        assert abs(lf - ll) <= (1e-7 if i == 0 else 2e-4) * max(1.0, abs(lf)) and abs(gf - gl) <= (1e-6 if i == 0 else 2e-3) * max(1.0, abs(gf))
    rel = ((fused_params - loop_params).norm() / fused_params.norm()).item()
    print(f"[trainer-loop] parameters after {steps} steps: rel {rel:.2e}")
    assert rel < 1e-3


def test_apply_ddp_after_the_step_object_was_built_still_exchanges():
    """Round-5 advice: the owner of the gradient exchange is decided per step.  A MI355XSFTStep built on a plain model (parallel=None) whose model is handed
    to apply_ddp() AFTERWARDS must exchange through the model's hooks from the next step on (it used to blank them for its backward: replicas diverging in
    silence); a step that has its own reducer must refuse such a model loudly."""
    from finetrainers_amd.parallel import MI355XParallelBackend
    from finetrainers_amd.trainer import MI355XSFTStep

    spec, model, cond, latd, _, _ = _model_and_batch(3, 2, 2, 4, 6)
    os.environ.setdefault("MASTER_PORT", "29539")
    step = MI355XSFTStep(model, spec, lr=1e-3, parallel=None)  # built BEFORE apply_ddp
    sig = torch.tensor([0.3, 0.8], device=_dev())
    step.step(dict(cond), dict(latd), sigmas=sig, force_first_frame_branch=False)
    backend = MI355XParallelBackend(world_size=1, dp_degree=1, backend="nccl", exercise_collectives=True)
    try:
        model = backend.apply_ddp(model, backend.get_mesh())
        hook, fin = model._grad_bucket_hook, model._grad_bucket_finish
        before = backend.reducer.buckets_issued
        out = step.step(dict(cond), dict(latd), sigmas=sig, force_first_frame_branch=False)
        assert torch.isfinite(out["loss"]) and torch.isfinite(out["grad_norm"])
        assert backend.reducer.buckets_issued == before + 1 and not backend.reducer._pending
        assert model._grad_bucket_hook == hook and model._grad_bucket_finish == fin
        # a step that brought its own reducer, on a model that acquired hooks later: refuse (the gradients would be averaged twice)
        own = MI355XSFTStep.__new__(MI355XSFTStep)
        own.__dict__.update(step.__dict__)
        own.reducer = backend.reducer
        with pytest.raises(RuntimeError, match="averaged twice"):
            own.step(dict(cond), dict(latd), sigmas=sig, force_first_frame_branch=False)
    finally:
        backend.destroy()


def test_fused_step_on_an_apply_ddp_model_keeps_the_models_exchange():
    """Round-4 advice: MI355XCheckpointer wants the MI355XSFTStep, MI355XParallelBackend.apply_ddp() wires the exchange on the MODEL -- mixing the two is
    the documented use.  A fused step on such a model must (a) leave the hooks apply_ddp installed in place, step after step, (b) exchange through them
    (buckets are issued, the backward ends the exchange), (c) refuse a second exchange of its own, and (d) foreign .grad tensors must raise instead of
    silently skipping the all-reduce."""
    from finetrainers_amd.parallel import DataParallelBackend, MI355XParallelBackend
    from finetrainers_amd.trainer import MI355XSFTStep

    spec, model, cond, latd, _, _ = _model_and_batch(3, 2, 2, 4, 6)
    os.environ.setdefault("MASTER_PORT", "29537")
    backend = MI355XParallelBackend(world_size=1, dp_degree=1, backend="nccl", exercise_collectives=True)  # a one-rank RCCL communicator
    try:
        model = backend.apply_ddp(model, backend.get_mesh())
        hook, fin = model._grad_bucket_hook, model._grad_bucket_finish
        assert hook is not None and fin is not None
        with pytest.raises(ValueError, match="averaged twice"):
            MI355XSFTStep(model, spec, parallel=backend._dp)
        step = MI355XSFTStep(model, spec, lr=1e-3, parallel=None)
        sig = torch.tensor([0.3, 0.8], device=_dev())
        for i in range(2):
            before = backend.reducer.buckets_issued
            out = step.step(dict(cond), dict(latd), sigmas=sig, force_first_frame_branch=False)
            assert torch.isfinite(out["loss"]) and torch.isfinite(out["grad_norm"])
            assert backend.reducer.buckets_issued == before + 1 and not backend.reducer._pending   # 3 blocks < 7: one bucket, ended by the backward
            assert model._grad_bucket_hook == hook and model._grad_bucket_finish == fin            # still wired after the step
        # foreign .grad tensors under an installed exchange: loud
        model.lora_A.grad = torch.zeros_like(model.lora_A)
        model.lora_B.grad = torch.zeros_like(model.lora_B)
        pred, target, sigmas = spec.forward(transformer=model, condition_model_conditions=dict(cond), latent_model_conditions=dict(latd),
                                            sigmas=sig.reshape(-1, 1, 1, 1, 1), force_first_frame_branch=False)
        with pytest.raises(RuntimeError, match="foreign .grad"):
            (pred.float() - target.float()).pow(2).mean().backward()
        model.lora_A.grad = model.lora_B.grad = None
    finally:
        backend.destroy()
