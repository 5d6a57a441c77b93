"""End-to-end parity of the MI355X LTX-Video DiT (forward, loss, LoRA gradients, optimiser step) against
the CPU oracle on identical noised latents + timesteps.  Also dumps every stashed activation's error so a
failure localises to one kernel.  Run on the MI355X box: pytest -m gpu."""

import json
import os
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

bf16 = torch.bfloat16

# Stated tolerances (bf16 storage, fp32 accumulation; rounding points mirror the eager reference graph; the oracle's attention is the
# real torch SDPA and its LoRA branch is fp32, as the reference runs them):
LOSS_RTOL = 1e-3           # north_star: loss within 1e-3 relative (measured: ~1e-5)
# LoRA gradients.  The north star says 1e-3; tests/test_oracle.py::test_lora_gradient_tolerance_triangle shows that the bf16 oracle moves
# by ~4e-3 when only the fp32 SUMMATION ORDER of its frozen linears changes (all rounding points identical) -- the floor for any two
# implementations of the reference's bf16 graph.  Every case below measures that floor ON THE SAME INPUTS (oracle vs
# ltx.accumulation_order_variant) and requires  kernel-vs-oracle <= FLOOR_FACTOR x floor  (global) and <= FLOOR_FACTOR_WORST x floor (the
# worst single adapter tensor, a max over 16-448 noisy values).
FLOOR_FACTOR = 1.5          # full-depth cfg 2 (offline yardsticks, linear-order floor only)
FLOOR_FACTOR_WORST = 2.0
# Round 4: the small cases measure the floor with ALL order-type freedoms exercised at once (linear summation order, attention key order, fused
# q|k|v input gradient: singly or together they saturate at the same level) and account for the rounding noise the fused row-wise backward
# kernels REMOVE (r^2 = |oracle - fp32|^2 - |kernel - fp32|^2 > 0: the kernel sits closer to exact arithmetic).  kernel-vs-oracle is then bounded
# by EXPLAINED_FACTOR x sqrt(floor^2 + r^2) (observed 1.14-1.18, profiles/r04_parity.txt) and the worst adapter by WORST_FACTOR x its floor
# (observed 1.16-1.19).
EXPLAINED_FACTOR = 1.30
WORST_FACTOR = 1.50
# full-depth cfg 2 (minutes of oracle time per evaluation): the two yardsticks were measured ONCE on the host CPU by
# tools/measure_cfg2_yardsticks.py (profiles/r03_cfg2_yardsticks.json: the oracle's own summation-order floor, and the bf16 oracle's distance
# from the fp32 evaluation of the same graph) together with a strided sample of the fp32 oracle's gradients
# (tests/golden/cfg2_fp32_grad_sample.safetensors); the bounds below are derived from THEM, never from the kernel's own residual:
#   kernel vs bf16 oracle  <= FLOOR_FACTOR x floor   (global)   and   <= FLOOR_FACTOR_WORST x floor (worst adapter)
#   kernel vs fp32 oracle  <= FP32_FACTOR x (bf16 oracle vs fp32 oracle)      -- the kernels may not be further from exact arithmetic
#                                                                                than the reference's own bf16 path is
FP32_FACTOR = 1.10
# ... and the same claim as ABSOLUTE numbers at BASELINE config 2, full depth (measured 1.64e-3 / 8.5e-3; north_star's 1e-3 is below the 1.17e-3 by which
# the reference's own bf16 graph moves when only its fp32 summation order changes -- BASELINE.md states this as the claimed tolerance)
CFG2_GLOBAL_ABS = 2.0e-3
CFG2_WORST_ABS = 1.0e-2


def _dev():
    return torch.device("cuda", 0)


def rel_l2(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _oracle_trace(model, inp):
    """Oracle forward with named intermediates of every block (restated from the oracle's own block forward)."""
    from oracle import ltx

    tr = {}
    sig5 = inp.sigmas.view(-1, 1, 1, 1, 1)
    lat = ltx.normalize_latents(inp.latents, inp.latents_mean, inp.latents_std)
    if inp.first_frame_sigma is not None:
        ffs = torch.min(inp.first_frame_sigma.view(-1, 1, 1, 1, 1), sig5.new_full(sig5.shape, 0.25))
        noisy = torch.cat([ltx.flow_match_xt(lat[:, :, :1], inp.noise[:, :, :1], ffs), ltx.flow_match_xt(lat[:, :, 1:], inp.noise[:, :, 1:], sig5)], 2)
    else:
        noisy = ltx.flow_match_xt(lat, inp.noise, sig5)
    x = ltx.pack_latents(noisy).to(lat).contiguous()
    B, S, _ = x.shape
    F_, H_, W_ = inp.latents.shape[2:]
    rope = model.rope(x, F_, H_, W_, [1 / (25 / 8), 32, 32])
    mask = ((1 - inp.encoder_attention_mask.to(x.dtype)) * -10000.0).unsqueeze(1)
    t = (inp.sigmas.view(-1, 1, 1).expand(-1, S, -1) * 1000.0).long()
    temb, emb = model.time_embed(t.flatten(), batch_size=B, hidden_dtype=x.dtype)
    temb, emb = temb.view(B, S, -1), emb.view(B, S, -1)
    tr["temb"], tr["emb"] = temb[:, 0], emb[:, 0]
    tr["tsin"] = model.time_embed.emb.time_proj(t[:, 0, 0]).to(x.dtype)
    tr["xt"] = x
    h = model.proj_in(x)
    e = model.caption_projection(inp.encoder_hidden_states).view(B, -1, h.size(-1))
    tr["e"] = e
    for l, blk in enumerate(model.transformer_blocks):
        tr[f"{l}.hs"] = h
        n = blk.norm1(h)
        ada = blk.scale_shift_table[None, None] + temb.reshape(B, S, 6, -1)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = ada.unbind(dim=2)
        n = n * (1 + scale_msa) + shift_msa
        tr[f"{l}.n1"] = n
        a1 = blk.attn1
        qr, kr, vr = a1.to_q(n), a1.to_k(n), a1.to_v(n)
        tr[f"{l}.qkv"] = torch.cat([qr, kr, vr], dim=-1)
        q = ltx.apply_rotary_emb(a1.norm_q(qr), rope)
        k = ltx.apply_rotary_emb(a1.norm_k(kr), rope)
        tr[f"{l}.qrot"], tr[f"{l}.krot"] = q, k
        sp = lambda z: z.unflatten(2, (a1.heads, -1)).transpose(1, 2)
        o = ltx.native_sdpa(sp(q), sp(k), sp(vr), None).transpose(1, 2).flatten(2, 3)
        tr[f"{l}.o1"] = o
        h = h + a1.to_out[0](o) * gate_msa
        tr[f"{l}.h1"] = h
        a2 = blk.attn2
        q2r = a2.to_q(h)
        tr[f"{l}.q2raw"] = q2r
        q2 = a2.norm_q(q2r)
        tr[f"{l}.q2n"] = q2
        k2r, v2r = a2.to_k(e), a2.to_v(e)
        tr[f"{l}.kv2raw"] = torch.cat([k2r, v2r], dim=-1)
        k2 = a2.norm_k(k2r)
        tr[f"{l}.k2n"] = k2
        am = mask.repeat_interleave(a2.heads, dim=0).view(B, a2.heads, -1, mask.shape[-1])
        o2 = ltx.native_sdpa(sp(q2), sp(k2), sp(v2r), am).transpose(1, 2).flatten(2, 3)
        tr[f"{l}.o2"] = o2
        h = h + a2.to_out[0](o2)
        tr[f"{l}.h2"] = h
        n2 = blk.norm2(h) * (1 + scale_mlp) + shift_mlp
        z = blk.ff.net[0].proj(n2)
        tr[f"{l}.z"] = z
        h = h + blk.ff.net[2](torch.nn.functional.gelu(z, approximate="tanh")) * gate_mlp
    tr[f"{len(model.transformer_blocks)}.hs"] = h
    return tr


def _build(num_layers, B, F_, H_, W_, first_frame, seed, rank=64, alpha=64.0, sigma_scheme=None):
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification
    from oracle import ltx

    cfg = ltx.LTXConfig.production(num_layers=num_layers)
    omodel = ltx.build_model(cfg, seed=0, rank=rank, alpha=float(alpha), lora_b_std=0.02)
    mask_lens = [32, 96][:B]
    inp = ltx.synth_inputs(cfg, B, F_, H_, W_, seed=seed, mask_lens=mask_lens, sigmas=[0.25, 0.7][:B])
    inp.latents_mean = torch.randn(cfg.in_channels, generator=torch.Generator().manual_seed(5)) * 0.1
    inp.latents_std = 1.0 + 0.2 * torch.rand(cfg.in_channels, generator=torch.Generator().manual_seed(6))
    if first_frame:
        inp.first_frame_sigma = torch.tensor([0.1, 0.6][:B])
    if sigma_scheme == "logit_normal":  # (only the 4-block cfg-2-size case: the full-depth cfg-2 test keeps sigma {0.25, 0.7}, which its committed fp32 sample was made with)
        # sigma draws of --flow_weighting_scheme logit_normal (utils/diffusion.py:38-63 -> finetrainers_amd.utils.diffusion, pinned to the reference's
        # fixtures in tests/test_host.py): u = sigmoid(N(0, 1)), indices into the 1000-entry table
        from finetrainers_amd.utils import diffusion as du
        u = du.compute_density_for_timestep_sampling("logit_normal", B, logit_mean=0.0, logit_std=1.0, generator=torch.Generator().manual_seed(11))
        inp.sigmas = torch.linspace(1, 1000, 1000).flip(0)[(u * 1000).long().clamp(max=999)] / 1000.0
    spec = MI355XLTXVideoModelSpecification(transformer_config=LTXTransformerConfig(num_layers=num_layers))
    gmodel = spec.load_diffusion_models(state_dict=omodel.state_dict(), device=_dev())["transformer"]
    gmodel.add_adapter(r=rank, lora_alpha=alpha)
    gmodel.load_lora_state_dict({k: v for k, v in omodel.state_dict().items() if "lora_" in k})
    return cfg, omodel, inp, spec, gmodel


def _gpu_forward(spec, gmodel, inp):
    dev = _dev()
    pred, target, sig = spec.forward(
        transformer=gmodel,
        condition_model_conditions={"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)},
        latent_model_conditions={"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std,
                                 "num_frames": inp.latents.shape[2], "height": inp.latents.shape[3], "width": inp.latents.shape[4]},
        sigmas=inp.sigmas.view(-1, 1, 1, 1, 1).to(dev),
        noise=inp.noise.to(dev),
        first_frame_sigma=None if inp.first_frame_sigma is None else inp.first_frame_sigma.to(dev),
        force_first_frame_branch=inp.first_frame_sigma is not None,
    )
    return pred, target, sig


CASES = [
    # layers, B, F, H, W, first_frame
    (2, 1, 2, 4, 4, False),   # BASELINE config 1 geometry (9x128x128 clip -> 32 tokens), 2 blocks
    (2, 2, 3, 4, 6, True),    # batch 2, ragged text masks {32, 96}, first-frame conditioning branch
    (1, 2, 2, 8, 10, False),  # 160 tokens: spans two 128-row GEMM tiles and several attention tiles
    (28, 1, 2, 4, 4, False),  # BASELINE config 1 EXACTLY: 28 blocks, batch 1, latents [1,128,2,4,4]
    (28, 1, 2, 4, 4, True),   # ... with the first-frame conditioning branch
    (4, 2, 7, 16, 24, True),  # BASELINE config 2's clip (2 x 2688 tokens) on 4 blocks, first-frame branch ON, sigmas drawn by the logit-normal scheme
]


def _run_parity_case(num_layers, B, F_, H_, W_, first_frame, measure_floor, trace_activations, tag, rank=64, alpha=64.0, keep_gpu_grads=None, sigma_scheme=None):
    from finetrainers_amd.trainer import sft_loss
    from oracle import ltx

    cfg, omodel, inp, spec, gmodel = _build(num_layers, B, F_, H_, W_, first_frame, seed=3, rank=rank, alpha=alpha, sigma_scheme=sigma_scheme)
    S = F_ * H_ * W_
    D = 2048

    # ---- oracle ----
    t0 = time.time()
    loss_ref, pred_ref, target_ref = ltx.forward_loss(omodel, inp, contiguous_hidden_states=True)
    loss_ref.backward()
    t_oracle = time.time() - t0
    grads_ref = {n.replace(".default", ""): p.grad.detach().clone() for n, p in ltx.lora_parameters(omodel)}
    pred_ref, target_ref, loss_ref_v = pred_ref.detach(), target_ref.detach(), loss_ref.item()
    floor_glob = floor_worst = None
    grads_32 = None
    if measure_floor:
        # the floor = the oracle against itself with every ORDER-type freedom an implementation has exercised at once: fp32 summation order of
        # the frozen linears, key order of the attention (tile / rescale order), and the one documented rounding-point difference (q|k|v input
        # gradients in one fp32 accumulator).  Measured singly they are each about as large as all three together (a one-ulp change anywhere
        # decorrelates every downstream bf16 rounding: the noise saturates), so the floor is a property of the graph, not of one kernel.
        import contextlib
        parts = {}
        for name, mk in (("linear order", lambda: [ltx.accumulation_order_variant(512)]), ("attention order", lambda: [ltx.attention_order_variant()]),
                         ("fused qkv dgrad", lambda: [ltx.fused_qkv_dgrad_variant()]),
                         ("all three", lambda: [ltx.accumulation_order_variant(512), ltx.attention_order_variant(), ltx.fused_qkv_dgrad_variant()])):
            if name != "all three" and num_layers > 2:
                continue  # the single-effect rows only where an oracle evaluation is cheap
            with contextlib.ExitStack() as st:
                for c in mk():
                    st.enter_context(c)
                g_ord, _ = ltx.lora_grads(omodel, inp)
            parts[name] = ltx.grads_rel_l2(g_ord, grads_ref)
            print(f"[dit-floor] {tag}: oracle vs itself, {name:16s}: {parts[name][0]:.3e} / {parts[name][1]:.3e}")
            del g_ord
        floor_glob, floor_worst = parts["all three"]
        # the third corner of the triangle: the SAME graph on the same (bf16-valued) weights and inputs evaluated in fp32
        m32 = ltx.build_model(cfg, seed=0, rank=rank, alpha=float(alpha), lora_b_std=0.02, dtype=torch.float32)
        m32.load_state_dict({k: v.float() for k, v in omodel.state_dict().items()})
        inp32 = ltx.StepInputs(**{k: (getattr(inp, k).float() if torch.is_tensor(getattr(inp, k)) else getattr(inp, k)) for k in
                                  ("latents", "latents_mean", "latents_std", "encoder_hidden_states", "encoder_attention_mask", "sigmas", "noise", "first_frame_sigma")})
        grads_32, _ = ltx.lora_grads(m32, inp32)
        del m32
    trace = None
    if trace_activations:
        with torch.no_grad():
            trace = _oracle_trace(omodel, inp)

    # ---- MI355X ----
    pred, target, sig = _gpu_forward(spec, gmodel, inp)
    loss = sft_loss(pred, target, sig, "none")
    loss.backward()
    torch.cuda.synchronize()

    rows = []
    worst = 0.0
    if trace is not None:
        T = cfg.text_seq_len
        shapes = {"n1": (B * S, D), "qkv": (B * S, 3 * D), "qrot": (B * S, D), "krot": (B * S, D), "o1": (B * S, D), "h1": (B * S, D),
                  "q2raw": (B * S, D), "q2n": (B * S, D), "o2": (B * S, D), "h2": (B * S, D),
                  "z": (B * S, 4 * D)}
        for name, shp in (("tsin", (B, 256)), ("temb", (B, 6 * D)), ("emb", (B, D)), ("e", (B * T, D))):
            err = rel_l2(gmodel.workspace_tensor(name, 0, shp), trace[name].reshape(shp))
            rows.append((name, err))
        for l in range(num_layers):
            err = rel_l2(gmodel.workspace_tensor("hs", l, (B * S, D)), trace[f"{l}.hs"].reshape(B * S, D))
            rows.append((f"{l}.hs", err))
            for name, shp in shapes.items():
                err = rel_l2(gmodel.workspace_tensor(name, l, shp), trace[f"{l}.{name}"].reshape(shp))
                rows.append((f"{l}.{name}", err))
        # text-side K/V of every block live in all-block arrays: kv2_all [B*T, L*2D], k2n_all [B*T, L, D]
        kv_all = gmodel.workspace_tensor("kv2_all", 0, (B * T, num_layers * 2 * D))
        k2n_all = gmodel.workspace_tensor("k2n_all", 0, (B * T, num_layers, D))
        for l in range(num_layers):
            rows.append((f"{l}.kv2raw", rel_l2(kv_all[:, l * 2 * D:(l + 1) * 2 * D], trace[f"{l}.kv2raw"].reshape(B * T, 2 * D))))
            rows.append((f"{l}.k2n", rel_l2(k2n_all[:, l], trace[f"{l}.k2n"].reshape(B * T, D))))
        rows.append((f"{num_layers}.hs", rel_l2(gmodel.workspace_tensor("hs", num_layers, (B * S, D)), trace[f"{num_layers}.hs"].reshape(B * S, D))))
        for n, e in rows:
            if num_layers <= 2 or n.endswith(".hs"):
                print(f"[dit-trace] {n:12s} rel_l2={e:.3e}")
            worst = max(worst, e)

        from finetrainers_amd import ops as _ops
        dev = _dev()
        ffd = None if inp.first_frame_sigma is None else torch.min(inp.first_frame_sigma, torch.full_like(inp.first_frame_sigma, 0.25)).to(dev)
        xt_g, _ = _ops.noise_pack(inp.latents.to(dev), inp.noise.to(dev), inp.latents_mean.to(dev), inp.latents_std.to(dev), inp.sigmas.to(dev), ffd,
                                  H_ * W_ if ffd is not None else 0)
        xt_equal = torch.equal(xt_g.cpu(), trace["xt"])
        print(f"[dit] x_t equal={xt_equal} rel_l2={rel_l2(xt_g, trace['xt']):.3e}")
        assert xt_equal, "noised, packed latents must be bit-exact"
    pred_err = rel_l2(pred, pred_ref)
    tgt_equal = torch.equal(target.cpu(), target_ref)
    loss_rel = abs(loss.item() - loss_ref_v) / abs(loss_ref_v)
    print(f"[dit] {tag}: pred rel_l2={pred_err:.3e} target_equal={tgt_equal} loss={loss.item():.6f} ref={loss_ref_v:.6f} rel={loss_rel:.3e} (oracle {t_oracle:.1f} s)")

    gv = {k: v.float().cpu() for k, v in gmodel.lora_grad_views().items()}
    if keep_gpu_grads is not None:
        keep_gpu_grads["grads"] = gv
    glob, worst_adapter = ltx.grads_rel_l2(gv, grads_ref)
    per = {k: rel_l2(gv[k], g) for k, g in grads_ref.items()}
    if num_layers <= 2:
        for k, g in grads_ref.items():
            print(f"[dit-grad] {k:58s} rel_l2={per[k]:.3e} |g|={g.norm().item():.3e}")
    floor_s = "" if floor_glob is None else f"; order floor {floor_glob:.3e} / {floor_worst:.3e} -> ratio {glob / floor_glob:.2f} / {worst_adapter / floor_worst:.2f}"
    print(f"[dit] {tag}: global LoRA-grad rel_l2={glob:.3e} worst adapter={worst_adapter:.3e}{floor_s}")
    k32 = o32 = None
    explained = None
    if grads_32 is not None:
        k32, o32 = ltx.grads_rel_l2(gv, grads_32), ltx.grads_rel_l2(grads_ref, grads_32)
        # What is left over the floor: the fused row-wise backward kernels keep fp32 between ops where eager autograd rounds every intermediate
        # gradient to bf16 -- rounding noise REMOVED, which shows as the kernel sitting closer to the fp32 evaluation than the bf16 oracle does.
        # Noise removed r^2 = |O - F|^2 - |K - F|^2, and it separates K from O in quadrature with the order floor.
        r2 = max(0.0, o32[0] ** 2 - k32[0] ** 2)
        explained = (floor_glob ** 2 + r2) ** 0.5
        print(f"[dit] {tag}: vs the fp32 evaluation of the graph: kernel {k32[0]:.3e} / {k32[1]:.3e}, bf16 oracle {o32[0]:.3e} / {o32[1]:.3e}; "
              f"rounding noise removed r = {r2 ** 0.5:.3e} -> sqrt(floor^2 + r^2) = {explained:.3e}, kernel / explained = {glob / explained:.2f}")

    out_dir = os.environ.get("FTMI_REPORT_DIR", "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"parity_{tag}.json"), "w") as f:
            json.dump({"case": {"layers": num_layers, "B": B, "S": S, "first_frame": first_frame}, "trace": rows, "pred_rel_l2": pred_err,
                       "loss": loss.item(), "loss_ref": loss_ref_v, "loss_rel": loss_rel, "grad_global_rel_l2": glob, "grad_worst_adapter": worst_adapter,
                       "floor_global": floor_glob, "floor_worst_adapter": floor_worst, "oracle_seconds": t_oracle,
                       "kernel_vs_fp32_oracle": k32, "bf16_oracle_vs_fp32_oracle": o32,
                       "grad_per_adapter": per if num_layers <= 2 else None}, f, indent=1)
    except OSError:
        pass

    assert tgt_equal, "flow-match target must be bit-exact"
    if trace is not None:
        # rounding noise of independent bf16 roundings grows like a random walk over the depth: 1.0e-2 * sqrt(L / 2) (observed on MI355X: 4.2e-3 at
        # 2 blocks, 3.3e-3 at 4 blocks of the cfg-2 clip, 8.1e-3 at 28 blocks -- the rotated keys of the last block) -- the round-3 bound 2e-2 * L / 4 (0.14 at 28 blocks) could not localise anything
        worst_name = max(rows, key=lambda r: r[1])[0]
        print(f"[dit-trace-worst] {tag}: {worst_name} rel_l2={worst:.3e} (bound {1.0e-2 * max(1.0, num_layers / 2) ** 0.5:.3e})")
        assert worst < 1.0e-2 * max(1.0, num_layers / 2) ** 0.5, f"an activation diverged ({worst_name}: rel_l2 {worst:.3e})"
    assert pred_err < 1e-2 * max(1.0, num_layers / 7)
    assert loss_rel < LOSS_RTOL, f"loss {loss.item()} vs oracle {loss_ref_v}"
    if k32 is not None:
        # against exact (fp32) arithmetic the kernels may not be further away than the reference's own bf16 path (+ 15 %: the two bf16
        # evaluations differ from each other by the floor)
        assert k32[0] < 1.15 * o32[0] + 2e-4, f"kernel vs fp32 oracle {k32[0]:.3e}, bf16 oracle vs fp32 oracle {o32[0]:.3e}"
        assert k32[1] < 1.30 * o32[1] + 5e-4
    _run_parity_case.last_explained = explained
    return glob, worst_adapter, floor_glob, floor_worst


@pytest.mark.parametrize("num_layers,B,F_,H_,W_,first_frame", CASES)
def test_dit_forward_backward_parity(num_layers, B, F_, H_, W_, first_frame):
    tag = f"L{num_layers}_B{B}_S{F_ * H_ * W_}{'_ff' if first_frame else ''}"
    scheme = "logit_normal" if F_ * H_ * W_ == 2688 else None  # the cfg-2-size case draws its sigmas the way --flow_weighting_scheme logit_normal does
    glob, worst_adapter, floor_glob, floor_worst = _run_parity_case(num_layers, B, F_, H_, W_, first_frame, True, True, tag, sigma_scheme=scheme)
    explained = _run_parity_case.last_explained
    assert glob < EXPLAINED_FACTOR * explained, f"global LoRA gradient error {glob:.3e} vs sqrt(floor^2 + removed rounding noise^2) = {explained:.3e}"
    assert worst_adapter < WORST_FACTOR * floor_worst, f"worst adapter {worst_adapter:.3e} vs floor {floor_worst:.3e}"


@pytest.mark.parametrize("rank,alpha", [(128, 128.0), (64, 32.0), (32, 32.0)])
def test_dit_parity_other_ranks(rank, alpha):
    """Rank 128 (two K-extension steps per plane);
    alpha != rank exercises the LoRA scale (alpha / rank) that rank 64 / alpha 64 leaves at 1; rank 32 (the reference's LTX example,
    examples/training/sft/ltx_video/crush_smol_lora/train.sh:75-76) runs on zero-padded rank-64 storage."""
    tag = f"L2_B2_S72_r{rank}_a{int(alpha)}"
    glob, worst_adapter, floor_glob, floor_worst = _run_parity_case(2, 2, 3, 4, 6, False, True, False, tag, rank=rank, alpha=alpha)
    explained = _run_parity_case.last_explained
    assert glob < EXPLAINED_FACTOR * explained, f"global LoRA gradient error {glob:.3e} vs sqrt(floor^2 + removed rounding noise^2) = {explained:.3e}"
    assert worst_adapter < WORST_FACTOR * floor_worst, f"worst adapter {worst_adapter:.3e} vs floor {floor_worst:.3e}"


def test_rank32_padding_stays_zero_through_optimiser_steps():
    """Rank 32 on rank-64 storage: after fused clip + AdamW steps (weight decay on) the padded rows of A / columns of B are still exact
    zeros, the parameters' gradients have the rank-32 shape, and the padded gradient entries the kernels produced are exact zeros."""
    from finetrainers_amd.trainer import MI355XSFTStep

    cfg, omodel, inp, spec, gmodel = _build(1, 2, 2, 4, 4, False, seed=5, rank=32, alpha=32.0)
    dev = _dev()
    step = MI355XSFTStep(gmodel, spec, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0)
    cond = {"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)}
    lat = {"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std,
           "num_frames": inp.latents.shape[2], "height": inp.latents.shape[3], "width": inp.latents.shape[4]}
    before = gmodel.lora_A.detach().clone()
    for _ in range(3):
        pred, target, sig = spec.forward(transformer=gmodel, condition_model_conditions=dict(cond), latent_model_conditions=dict(lat),
                                         sigmas=inp.sigmas.view(-1, 1, 1, 1, 1).to(dev), noise=inp.noise.to(dev))
        ((pred.float() - target.float()) ** 2).mean().backward()  # the reference loop's own loss.backward(): autograd edge, foreign-free .grad
        assert gmodel.lora_A.grad.shape == (1, 8, 32, 2048) and gmodel.lora_B.grad.shape == (1, 8, 2048, 32)
        gfull = gmodel._grad_flat
        n = gmodel._lora_A_full.numel()
        assert gfull[:n].view_as(gmodel._lora_A_full)[:, :, 32:, :].abs().max().item() == 0.0
        assert gfull[n:].view_as(gmodel._lora_B_full)[:, :, :, 32:].abs().max().item() == 0.0
        gmodel.lora_A.grad = gmodel.lora_B.grad = None
        out = step.step(cond, lat, sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev))
    torch.cuda.synchronize()
    assert torch.isfinite(out["loss"]).item() and out["grad_norm"].item() > 0
    assert gmodel._lora_A_full[:, :, 32:, :].abs().max().item() == 0.0 and gmodel._lora_B_full[:, :, :, 32:].abs().max().item() == 0.0
    assert not torch.equal(gmodel.lora_A.detach(), before)


def test_step_state_resume_and_lr_schedule():
    """Save after two optimiser steps (parameters via transformer.state_dict(), AdamW moments / counters / schedule clock via
    MI355XSFTStep.state_dict()), rebuild everything from the saved state, take the third step: the same update as the run that
    never stopped (to fp32-atomic rounding).  The schedule is the reference example's constant_with_warmup: step 1 runs at lr = 0 and must leave the parameters alone."""
    import copy

    from finetrainers_amd.trainer import MI355XSFTStep
    from finetrainers_amd.utils.lr_schedule import LRSchedule

    def fresh():
        cfg, omodel, inp, spec, gmodel = _build(1, 2, 2, 4, 4, False, seed=5)
        sched = LRSchedule.from_args(1e-3, "constant_with_warmup", num_warmup_steps=2)
        return inp, spec, gmodel, MI355XSFTStep(gmodel, spec, lr=1e-3, betas=(0.9, 0.99), weight_decay=1e-2, lr_scheduler=sched)

    dev = _dev()
    inp, spec, gmodel, step = fresh()
    cond = {"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)}
    lat = {"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std,
           "num_frames": inp.latents.shape[2], "height": inp.latents.shape[3], "width": inp.latents.shape[4]}
    kw = dict(sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False)
    p0 = gmodel.lora_flat.clone()
    step.step(cond, lat, **kw)
    assert torch.equal(gmodel.lora_flat, p0) and step.lr_scheduler.current_lr() == 5e-4  # warm-up: lr(0) = 0, then 0.5 * base
    step.step(cond, lat, **kw)
    saved_model = {k: v.clone() for k, v in gmodel.state_dict().items() if "lora_" in k}
    saved_step = copy.deepcopy(step.state_dict())
    step.step(cond, lat, **kw)
    torch.cuda.synchronize()
    want = gmodel.lora_flat.clone()
    assert step.lr_scheduler.current_lr() == 1e-3 and not torch.equal(want, p0)

    _, spec2, gmodel2, step2 = fresh()
    gmodel2.load_state_dict(saved_model, strict=False)
    step2.load_state_dict(saved_step)
    assert step2.step_count == 2 and step2.lr_scheduler.current_lr() == 1e-3
    step2.step(cond, lat, **kw)
    torch.cuda.synchronize()
    # the weight-gradient GEMMs add their token splits with fp32 atomics: two runs agree to rounding, not bit for bit
    rel = ((gmodel2.lora_flat - want).norm() / (want - p0).norm()).item()
    print(f"[resume] third step after reload vs uninterrupted: update rel diff {rel:.2e}")
    assert rel < 2e-2


def test_precomputed_posterior_path_equals_sampled_latents():
    """compute_posterior = False (what --enable_precomputation runs, trainer.py:374): the spec draws the latents from the stored moments,
    then proceeds exactly as with ready-made latents."""
    from finetrainers_amd import ops

    cfg, omodel, inp, spec, gmodel = _build(1, 2, 2, 4, 4, False, seed=9)
    dev = _dev()
    g = torch.Generator().manual_seed(4)
    mean = inp.latents.to(bf16)
    logvar = (torch.randn(mean.shape, generator=g) * 0.5 - 3.0).to(bf16)
    moments = torch.cat([mean, logvar], dim=1).to(dev)
    eps = torch.randn(mean.shape, generator=g).to(bf16).to(dev)
    cond = {"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)}
    lat = {"latents_mean": inp.latents_mean, "latents_std": inp.latents_std, "num_frames": mean.shape[2], "height": mean.shape[3], "width": mean.shape[4]}
    kw = dict(sigmas=inp.sigmas.view(-1, 1, 1, 1, 1).to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False)
    with torch.no_grad():
        p1, t1, _ = spec.forward(transformer=gmodel, condition_model_conditions=dict(cond), latent_model_conditions=dict(lat, latents=moments),
                                 compute_posterior=False, posterior_noise=eps, **kw)
        sampled = ops.posterior_sample(moments, eps)
        p2, t2, _ = spec.forward(transformer=gmodel, condition_model_conditions=dict(cond), latent_model_conditions=dict(lat, latents=sampled),
                                 compute_posterior=True, **kw)
    assert torch.equal(t1, t2) and torch.equal(p1, p2)
    assert not torch.equal(sampled.cpu(), mean)


def test_layerwise_upcasting_fp8_storage_semantics():
    """--layerwise_upcasting_modules transformer with float8_e4m3fn storage (trainer.py:111-118): the same Linears as the reference's skip patterns
    leave (args.py:395) hold fp8-representable weights and biases afterwards, bit-identical to the oracle's restatement of the cast; the step
    still matches the (equally cast) oracle."""
    from oracle import ltx

    cfg = ltx.LTXConfig.production(num_layers=1)
    torch.manual_seed(0)
    omodel = ltx.LTXVideoTransformer3DModel(cfg).to(bf16)
    plain = {k: v.clone() for k, v in omodel.state_dict().items()}
    names = ltx.apply_layerwise_casting(omodel)
    assert "transformer_blocks.0.attn1.to_q" in names and "transformer_blocks.0.ff.net.2" in names and "caption_projection.linear_1" in names
    assert not any(n.startswith(("proj_in", "proj_out", "time_embed")) or "norm" in n for n in names)
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification

    spec = MI355XLTXVideoModelSpecification(transformer_config=LTXTransformerConfig(num_layers=1))
    gmodel = spec.load_diffusion_models(state_dict=plain, device=_dev())["transformer"]
    cast = gmodel.apply_layerwise_casting(torch.float8_e4m3fn, torch.bfloat16)
    assert sorted(cast) == sorted(names)
    osd = omodel.state_dict()
    changed = 0
    for k, v in gmodel.state_dict().items():
        assert torch.equal(v.cpu(), osd[k]), k
        changed += int(not torch.equal(osd[k], plain[k]))
    assert changed >= 2 * len(names) - 2  # weights and biases of the cast Linears moved (a bias may survive by chance), nothing else did
    assert torch.equal(gmodel.w_o_t[0].cpu(), osd["transformer_blocks.0.attn1.to_out.0.weight"].t())  # the dgrad copies follow


def _cfg2_yardsticks():
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "..", "profiles", "r03_cfg2_yardsticks.json")) as f:
        y = json.load(f)
    from safetensors.torch import load_file

    sample = load_file(os.path.join(here, "golden", "cfg2_fp32_grad_sample.safetensors"))
    return y, sample


def test_full_depth_config2_parity():
    """BASELINE config 2 EXACTLY: 28 blocks, batch 2, latents [2,128,7,16,24] (2688 tokens), text masks {32, 96}, sigma {0.25, 0.7},
    LoRA rank 64 -- loss and every LoRA gradient against the CPU oracle run on this box's host cores (a few minutes with torch's default
    thread count), and against the fp32 evaluation of the same graph through the committed gradient sample.  Guard: the oracle is first
    timed on ONE block; if 28 blocks would not fit ORACLE_BUDGET_S the depth is reduced to what fits (never below 8), a sanity bound is checked and
    the test SKIPS with that message -- a slow or oversubscribed host must not hang the suite, and a reduced run must not read as a pass."""
    from oracle import ltx

    budget = float(os.environ.get("FTMI_ORACLE_BUDGET_S", "480"))
    cfg1 = ltx.LTXConfig.production(num_layers=1)
    m1 = ltx.build_model(cfg1, seed=0, rank=64, alpha=64.0, lora_b_std=0.02)
    inp1 = ltx.synth_inputs(cfg1, 2, 7, 16, 24, seed=3, mask_lens=[32, 96], sigmas=[0.25, 0.7])
    ltx.lora_grads(m1, inp1)  # warm-up (thread pool, allocator)
    t0 = time.time()
    ltx.lora_grads(m1, inp1)
    per_block = time.time() - t0
    del m1, inp1
    depth = 28 if 28 * per_block <= budget else max(8, int(budget / per_block))
    print(f"[dit] full_cfg2: oracle {per_block:.1f} s per block (fwd+bwd, batch 2) on {torch.get_num_threads()} threads -> {depth} blocks")
    keep = {}
    glob, worst_adapter, _, _ = _run_parity_case(depth, 2, 7, 16, 24, False, False, False, "full_cfg2" if depth == 28 else f"full_cfg2_REDUCED_to_L{depth}",
                                                 keep_gpu_grads=keep)
    if depth != 28:
        # a reduced run is NOT the configuration's parity: it must never count as a pass of this test (round-4 review).  The sanity bound still guards
        # the kernels, then the test reports itself as skipped with the depth it ran.
        assert glob < 2.5e-3 and worst_adapter < 1.3e-2, f"reduced-depth ({depth} blocks) sanity bound: {glob:.3e} / {worst_adapter:.3e}"
        pytest.skip(f"full-depth cfg-2 parity NOT run: the host oracle needs {per_block:.1f} s per block, 28 blocks exceed FTMI_ORACLE_BUDGET_S={budget:.0f} s; "
                    f"ran {depth} blocks instead (global {glob:.3e}, worst adapter {worst_adapter:.3e}: sanity bound only)")
    y, sample = _cfg2_yardsticks()
    stride = int(y["stride"])
    got = {k: v.flatten()[::stride] for k, v in keep["grads"].items()}
    assert set(got) == set(sample)
    g32, w32 = ltx.grads_rel_l2(got, sample)
    print(f"[dit] full_cfg2: vs bf16 oracle {glob:.3e} / worst {worst_adapter:.3e} (summation-order floor {y['floor_global']:.3e} / {y['floor_worst_adapter']:.3e}: "
          f"ratio {glob / y['floor_global']:.2f} / {worst_adapter / y['floor_worst_adapter']:.2f}); vs fp32 oracle (sampled) {g32:.3e} / {w32:.3e} "
          f"(bf16 oracle vs fp32 oracle {y['bf16_vs_fp32_global_sampled']:.3e} / {y['bf16_vs_fp32_worst_adapter_sampled']:.3e})")
    try:
        with open(os.path.join(os.environ.get("FTMI_REPORT_DIR", "gpurun_out"), "parity_full_cfg2_yardsticks.json"), "w") as f:
            json.dump({"kernel_vs_bf16_oracle": [glob, worst_adapter], "floor": [y["floor_global"], y["floor_worst_adapter"]],
                       "kernel_vs_fp32_oracle_sampled": [g32, w32],
                       "bf16_oracle_vs_fp32_oracle_sampled": [y["bf16_vs_fp32_global_sampled"], y["bf16_vs_fp32_worst_adapter_sampled"]]}, f, indent=1)
    except OSError:
        pass
    # the CLAIMED tolerance, as absolute numbers ("claimed tolerance"): a regression cannot hide behind a moving yardstick
    assert glob <= CFG2_GLOBAL_ABS, f"global LoRA gradient error {glob:.3e} above the claimed {CFG2_GLOBAL_ABS:.1e} (north_star asks 1e-3; floor of the bf16 graph {y['floor_global']:.3e})"
    assert worst_adapter <= CFG2_WORST_ABS, f"worst adapter {worst_adapter:.3e} above the claimed {CFG2_WORST_ABS:.1e}"
    assert glob < FLOOR_FACTOR * y["floor_global"], f"global LoRA gradient error {glob:.3e} vs {FLOOR_FACTOR} x floor {y['floor_global']:.3e}"
    assert worst_adapter < FLOOR_FACTOR_WORST * y["floor_worst_adapter"]
    assert g32 < FP32_FACTOR * y["bf16_vs_fp32_global_sampled"], f"kernel vs fp32 oracle {g32:.3e} vs the bf16 oracle's own distance {y['bf16_vs_fp32_global_sampled']:.3e}"
    assert w32 < 1.25 * y["bf16_vs_fp32_worst_adapter_sampled"]


def test_north_star_gradient_tolerance_1e3_is_reported_not_claimed():
    """north_star states "grads within 1e-3 rel of reference".  The full-depth test above asserts the tolerance this project CLAIMS (2.0e-3 global /
    1.0e-2 worst adapter at config 2, 1.5 x the bf16 graph's own summation-order floor of 1.17e-3) -- this test keeps the north-star number itself visible: it
    reads what that test measured in this session and is an EXPECTED FAILURE with the measured figure wherever 1e-3 is not met (it cannot be met by two bf16
    evaluations of this graph: the oracle moves by 1.17e-3 against itself), a pass if it ever is."""
    path = os.path.join(os.environ.get("FTMI_REPORT_DIR", "gpurun_out"), "parity_full_cfg2_yardsticks.json")
    if not os.path.exists(path):
        pytest.skip("the full-depth cfg-2 parity did not run in this session (no report to read)")
    rec = json.load(open(path))
    glob, floor = rec["kernel_vs_bf16_oracle"][0], rec["floor"][0]
    print(f"[dit] north star: LoRA gradients {glob:.3e} from the bf16 oracle at config 2 (target 1e-3; floor of the bf16 graph {floor:.3e})")
    if glob > 1e-3:
        pytest.xfail(f"north_star's 1e-3 is not met and not claimed: measured {glob:.3e} (the reference's own bf16 graph moves by {floor:.3e} under a pure fp32 reordering)")


def test_full_step_matches_oracle_step():
    """forward + loss + backward + clip + AdamW: updated LoRA parameters vs the oracle's torch.optim.AdamW step."""
    from finetrainers_amd.trainer import MI355XSFTStep
    from oracle import ltx

    cfg, omodel, inp, spec, gmodel = _build(1, 2, 2, 4, 4, False, seed=11)
    opt = ltx.make_optimizer(omodel, lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4)
    before = {n.replace(".default", ""): p.detach().clone() for n, p in ltx.lora_parameters(omodel)}
    loss_ref, gn_ref, grads_ref = ltx.sft_step(omodel, opt, inp, max_grad_norm=1.0, contiguous_hidden_states=True)
    grads_ref = {k.replace(".default", ""): v for k, v in grads_ref.items()}

    step = MI355XSFTStep(gmodel, spec, lr=5e-5, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-4, max_grad_norm=1.0)
    dev = _dev()
    out = step.step(
        condition_model_conditions={"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)},
        latent_model_conditions={"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std},
        sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False,
    )
    torch.cuda.synchronize()
    loss, gn = out["loss"].item(), out["grad_norm"].item()
    print(f"[step] loss {loss:.6f} vs {loss_ref.item():.6f}; grad_norm {gn:.6e} vs {gn_ref.item():.6e}")
    assert abs(loss - loss_ref.item()) / abs(loss_ref.item()) < LOSS_RTOL
    assert abs(gn - gn_ref.item()) / gn_ref.item() < 5e-3
    after = gmodel.lora_state_dict()
    # AdamW's first step is sign-like (update = -lr * g / (|g| + eps') ~ -+lr for every entry), so an entry of the update differs only
    # where the gradient's SIGN differs -- i.e. where |g| is below the gradient noise (floor ~4e-3 of the typical |g|).  Checked:
    # (a) the fraction of entries whose update differs in sign is tiny, (b) entries whose oracle gradient is not tiny agree closely.
    flips = total = 0
    num = den = 0.0
    for n, p in ltx.lora_parameters(omodel):
        k = n.replace(".default", "")
        d_ref = (p.detach() - before[k]).float()
        d_got = (after[k].detach().cpu() - before[k]).float()
        flips += (torch.sign(d_ref) != torch.sign(d_got)).sum().item()
        total += d_ref.numel()
        big = grads_ref[k].abs() > 0.05 * grads_ref[k].abs().mean()   # gradient entries well above the noise
        num += (d_got[big] - d_ref[big]).pow(2).sum().item()
        den += d_ref[big].pow(2).sum().item()
    upd = (num / max(den, 1e-30)) ** 0.5
    print(f"[step] update sign flips {flips}/{total} = {flips / total:.2e}; update rel_l2 on entries with a non-negligible gradient = {upd:.3e}")
    assert flips / total < 3e-3
    assert upd < 2e-2


def test_gradient_accumulation_and_buffer_recycling():
    """`.grad` semantics of the flat LoRA-gradient buffer: a second backward without zero_grad ADDS (the reference's
    gradient_accumulation_steps > 1), and after `grad = None` the recycled buffer holds exactly one step's gradient again."""
    cfg, omodel, inp, spec, gmodel = _build(1, 1, 2, 4, 4, False, seed=21)

    def one_backward():
        pred, target, _ = _gpu_forward(spec, gmodel, inp)
        ((pred.float() - target.float()) ** 2).mean().backward()
        torch.cuda.synchronize()

    gmodel.lora_A.grad = None
    gmodel.lora_B.grad = None
    one_backward()
    ga1, gb1 = gmodel.lora_A.grad.detach().clone(), gmodel.lora_B.grad.detach().clone()
    assert ga1.abs().max() > 0 and gb1.abs().max() > 0
    one_backward()  # accumulates into the live .grad
    rel_a = ((gmodel.lora_A.grad - 2 * ga1).norm() / (2 * ga1).norm()).item()
    rel_b = ((gmodel.lora_B.grad - 2 * gb1).norm() / (2 * gb1).norm()).item()
    print(f"[accumulate] dA rel {rel_a:.2e}  dB rel {rel_b:.2e}")
    assert rel_a < 1e-5 and rel_b < 1e-5
    gmodel.lora_A.grad = None
    gmodel.lora_B.grad = None
    one_backward()  # recycled buffer
    rel_a = ((gmodel.lora_A.grad - ga1).norm() / ga1.norm()).item()
    rel_b = ((gmodel.lora_B.grad - gb1).norm() / gb1.norm()).item()
    assert rel_a < 1e-5 and rel_b < 1e-5
    gmodel.lora_A.grad = None
    gmodel.lora_B.grad = None


def test_reference_dummy_model_runs_on_the_production_kernels():
    """The reference's own trainer tests run ``DummyLTXVideoModelSpecification`` (tests/models/ltx_video/base_specification.py:47-58: 4 heads x 8, one block, 8
    latent channels, caption width 32) -- until round 6 ``load_diffusion_models`` of this backend answered that geometry with "unsupported".  It now runs embedded
    in the 2048-wide layout by zero padding (finetrainers_amd/ltx_video/narrow.py) on the SAME kernels: prediction, loss and the rank-4 LoRA gradients against the
    oracle at that geometry, and every padded LoRA entry's gradient an exact zero (which is what keeps the padded model the narrow model over optimiser steps)."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification, MI355XNarrowLTXVideoTransformer3DModel
    from finetrainers_amd.trainer import sft_loss
    from oracle import ltx

    cfg = ltx.LTXConfig.dummy()
    rank, alpha = 4, 4.0
    omodel = ltx.build_model(cfg, seed=0, rank=rank, alpha=alpha, lora_b_std=0.02)
    B, F_, H_, W_ = 2, 2, 4, 4
    inp = ltx.synth_inputs(cfg, B, F_, H_, W_, seed=3, mask_lens=[32, 96], sigmas=[0.25, 0.7])
    loss_ref, pred_ref, target_ref = ltx.forward_loss(omodel, inp, contiguous_hidden_states=True)
    loss_ref.backward()
    grads_ref = {n.replace(".default", ""): p.grad.detach().clone() for n, p in ltx.lora_parameters(omodel)}

    tcfg = LTXTransformerConfig(in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_attention_heads=cfg.num_attention_heads,
                                attention_head_dim=cfg.attention_head_dim, cross_attention_dim=cfg.cross_attention_dim, num_layers=cfg.num_layers,
                                caption_channels=cfg.caption_channels)
    spec = MI355XLTXVideoModelSpecification(transformer_config=tcfg)
    gmodel = spec.load_diffusion_models(state_dict=omodel.state_dict(), device=_dev())["transformer"]
    assert isinstance(gmodel, MI355XNarrowLTXVideoTransformer3DModel)
    gmodel.add_adapter(r=rank, lora_alpha=alpha)
    gmodel.load_lora_state_dict({k: v for k, v in omodel.state_dict().items() if "lora_" in k})
    pred, target, sig = _gpu_forward(spec, gmodel, inp)
    loss = sft_loss(pred, target, sig, "none")
    loss.backward()
    torch.cuda.synchronize()

    assert pred.shape == pred_ref.shape
    pred_err = rel_l2(pred, pred_ref.detach())
    loss_rel = abs(loss.item() - loss_ref.item()) / abs(loss_ref.item())
    gv = {k: v.float().cpu() for k, v in gmodel.lora_grad_state_dict().items()}
    glob, worst = ltx.grads_rel_l2(gv, grads_ref)
    print(f"[dit] dummy 4x8: pred rel_l2={pred_err:.3e} target_equal={torch.equal(target.cpu(), target_ref)} loss rel={loss_rel:.3e} LoRA-grad rel_l2={glob:.3e} worst adapter={worst:.3e}")
    assert torch.equal(target.cpu(), target_ref)
    assert pred_err < 2e-2 and loss_rel < 2e-3
    assert glob < 3e-2 and worst < 6e-2
    # state dict in the reference's (narrow) shapes and names, both ways: what the save / resume paths read and write
    osd = omodel.state_dict()
    gsd = gmodel.state_dict()
    assert set(gsd) == set(osd)
    for k, v in osd.items():
        assert gsd[k].shape == v.shape and torch.equal(gsd[k].float().cpu(), v.float() if "lora_" in k else v.to(bf16).float()), k
    gmodel.load_state_dict(gsd)
    assert all(torch.equal(a.cpu(), gsd[k].cpu()) for k, a in gmodel.state_dict().items())
    # --gradient_checkpointing reaches the wide module through the wrapper: the same gradients from one block slot
    g_keep = {k: v.clone() for k, v in gv.items()}
    gmodel.inner.lora_A.grad = None
    gmodel.inner.lora_B.grad = None
    gmodel.enable_gradient_checkpointing()
    assert gmodel.inner.is_gradient_checkpointing
    pred2, target2, sig2 = _gpu_forward(spec, gmodel, inp)
    sft_loss(pred2, target2, sig2, "none").backward()
    torch.cuda.synchronize()
    g_ck = {k: v.float().cpu() for k, v in gmodel.lora_grad_state_dict().items()}
    ck_glob, _ = ltx.grads_rel_l2(g_ck, g_keep)
    print(f"[dit] dummy 4x8: gradients with checkpointing vs without: rel_l2={ck_glob:.2e}")
    assert torch.equal(pred2, pred) and ck_glob < 1e-6
    # the padding stays padding: gradients of every padded LoRA entry are exact zeros, activations of padded channels too
    lay = gmodel.layout
    ga, gb = gmodel.inner.lora_A.grad.clone(), gmodel.inner.lora_B.grad.clone()
    for i in range(8):
        a_sp, b_sp = lay.lora_spaces(i)
        ga[:, i][:, :, lay._index(a_sp)[0].to(ga.device)] = 0
        gb[:, i][:, lay._index(b_sp)[0].to(gb.device)] = 0
    assert ga.abs().max().item() == 0.0 and gb.abs().max().item() == 0.0
    full = gmodel.inner._lora_A_full.grad if gmodel.inner._lora_A_full.grad is not None else None
    assert full is None or full[:, :, rank:].abs().max().item() == 0.0


def test_reference_dummy_model_takes_fused_optimisation_steps():
    """Two optimisation steps of ``MI355XSFTStep`` on the reference's dummy geometry (zero-padded into the wide layout) against two steps of the oracle's torch
    AdamW at that geometry: losses, gradient norms, the updated rank-4 adapters -- and the padding is still padding afterwards (every padded LoRA entry an exact
    zero after weight decay, clipping and two AdamW updates), i.e. the padded model IS the narrow model, not an approximation that drifts."""
    from finetrainers_amd.ltx_video import LTXTransformerConfig, MI355XLTXVideoModelSpecification
    from finetrainers_amd.trainer import MI355XSFTStep
    from oracle import ltx

    cfg = ltx.LTXConfig.dummy()
    rank, alpha = 4, 4.0
    omodel = ltx.build_model(cfg, seed=0, rank=rank, alpha=alpha, lora_b_std=0.02)
    inp = ltx.synth_inputs(cfg, 2, 2, 4, 4, seed=11, mask_lens=[32, 96], sigmas=[0.25, 0.7])
    tcfg = LTXTransformerConfig(in_channels=cfg.in_channels, out_channels=cfg.out_channels, num_attention_heads=cfg.num_attention_heads,
                                attention_head_dim=cfg.attention_head_dim, cross_attention_dim=cfg.cross_attention_dim, num_layers=cfg.num_layers,
                                caption_channels=cfg.caption_channels)
    spec = MI355XLTXVideoModelSpecification(transformer_config=tcfg)
    gmodel = spec.load_diffusion_models(state_dict=omodel.state_dict(), device=_dev())["transformer"]
    gmodel.add_adapter(r=rank, lora_alpha=alpha)
    gmodel.load_lora_state_dict({k: v for k, v in omodel.state_dict().items() if "lora_" in k})
    opt = ltx.make_optimizer(omodel, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2)
    before = {n.replace(".default", ""): p.detach().clone() for n, p in ltx.lora_parameters(omodel)}
    step = MI355XSFTStep(gmodel, spec, lr=1e-3, betas=(0.9, 0.99), eps=1e-8, weight_decay=1e-2, max_grad_norm=1.0)
    dev = _dev()
    for it in range(2):
        loss_ref, gn_ref, _ = ltx.sft_step(omodel, opt, inp, max_grad_norm=1.0, contiguous_hidden_states=True)
        out = step.step(
            condition_model_conditions={"encoder_hidden_states": inp.encoder_hidden_states.to(dev), "encoder_attention_mask": inp.encoder_attention_mask.to(dev)},
            latent_model_conditions={"latents": inp.latents.to(dev), "latents_mean": inp.latents_mean, "latents_std": inp.latents_std},
            sigmas=inp.sigmas.to(dev), noise=inp.noise.to(dev), force_first_frame_branch=False,
        )
        torch.cuda.synchronize()
        loss, gn = out["loss"].item(), out["grad_norm"].item()
        print(f"[step] dummy 4x8, step {it}: loss {loss:.6f} vs {loss_ref.item():.6f}; grad_norm {gn:.6e} vs {gn_ref.item():.6e}")
        assert abs(loss - loss_ref.item()) / abs(loss_ref.item()) < 5e-3
        assert abs(gn - gn_ref.item()) / gn_ref.item() < 2e-2
    after = gmodel.lora_state_dict()
    num = den = 0.0
    for n, p in ltx.lora_parameters(omodel):
        k = n.replace(".default", "")
        d_ref = (p.detach() - before[k]).float()
        d_got = (after[k].detach().cpu() - before[k]).float()
        num += (d_got - d_ref).pow(2).sum().item()
        den += d_ref.pow(2).sum().item()
    upd = (num / max(den, 1e-30)) ** 0.5
    print(f"[step] dummy 4x8: update after two steps rel_l2 = {upd:.3e}")
    assert upd < 0.15  # (AdamW's early steps are sign-like: entries whose gradient is inside the bf16 noise flip)
    lay = gmodel.layout
    A, Bm = gmodel.inner._lora_A_full.detach().clone(), gmodel.inner._lora_B_full.detach().clone()
    assert A[:, :, rank:].abs().max().item() == 0.0 and Bm[:, :, :, rank:].abs().max().item() == 0.0
    for i in range(8):
        a_sp, b_sp = lay.lora_spaces(i)
        A[:, i][:, :, lay._index(a_sp)[0].to(A.device)] = 0
        Bm[:, i][:, lay._index(b_sp)[0].to(Bm.device)] = 0
    assert A.abs().max().item() == 0.0 and Bm.abs().max().item() == 0.0
