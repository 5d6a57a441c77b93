"""Close the "parity unpinned" caveat of oracle/ (DESIGN.md section 5): diff the restated DiTs against upstream diffusers on the reference's own dummy
configurations.  The block internals of LTX-Video, CogVideoX, Wan and HunyuanVideo live in ``diffusers`` (pinned 0.33 by the reference), which is absent
from the reference tree and from the build image, so oracle/*.py restate them from the published algorithm.  On any machine that has
``diffusers==0.33.*`` installed:

    python tools/diff_oracle_vs_diffusers.py [ltx] [cogvideox] [wan] [hunyuan]

For each model: build the upstream class with the dummy configuration of the reference's tests (tests/models/<model>/base_specification.py), copy its
weights into the oracle model through the name map below, run both in fp32 on the same seeded inputs and print the largest absolute difference of the
outputs and of every parameter gradient.  Expected: 0 or float-rounding (<= 1e-5); anything larger is a restatement bug in oracle/.
CPU only; nothing here is on the product path (tools/ may import oracle/ as the checker)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

try:
    import diffusers  # noqa: F401
except ImportError:
    sys.exit("diffusers is not installed here (the build image has no network): run this on a machine with diffusers==0.33 -- see the module docstring")


def ff_names(k: str, prefixes=("ff", "ff_context", "ffn")) -> str:
    """oracle FeedForward (proj_in / proj_out) -> diffusers FeedForward (net.0.proj / net.2)."""
    for a in prefixes:
        k = k.replace(f"{a}.proj_in.", f"{a}.net.0.proj.").replace(f"{a}.proj_out.", f"{a}.net.2.")
    return k


def hunyuan_names(k: str) -> str:
    k = k.replace("context_embedder.refiner_blocks.", "context_embedder.token_refiner.refiner_blocks.").replace(".norm_out_linear.", ".norm_out.linear.")
    if k.startswith("norm_out_linear."):
        k = "norm_out.linear." + k[len("norm_out_linear."):]
    if k.startswith("x_embedder."):
        k = "x_embedder.proj." + k[len("x_embedder."):]
    return ff_names(k)


def load_into_oracle(omodel, upstream, name_map):
    up = upstream.state_dict()
    sd = {}
    for k, v in omodel.state_dict().items():
        src = name_map(k)
        if src not in up:
            raise KeyError(f"oracle parameter {k} -> {src} not found upstream (first upstream keys: {list(up)[:6]})")
        sd[k] = up[src].reshape(v.shape).to(v.dtype)
    omodel.load_state_dict(sd)
    extra = set(up) - {name_map(k) for k in omodel.state_dict()}
    if extra:
        print(f"  note: {len(extra)} upstream tensors have no oracle counterpart, e.g. {sorted(extra)[:4]}")


def compare(name, out_o, out_u, omodel, upstream, name_map):
    out_o.square().mean().backward()
    out_u.square().mean().backward()
    worst = (out_o.detach() - out_u.detach()).abs().max().item()
    up = dict(upstream.named_parameters())
    gworst, gname = 0.0, ""
    for k, p in omodel.named_parameters():
        q = up[name_map(k)]
        if p.grad is None or q.grad is None:
            continue
        d = (p.grad - q.grad.reshape(p.grad.shape)).abs().max().item()
        if d > gworst:
            gworst, gname = d, k
    print(f"{name}: output max |diff| {worst:.3e}; parameter gradients max |diff| {gworst:.3e} ({gname})")
    return worst, gworst


def run_ltx():
    from diffusers import LTXVideoTransformer3DModel

    from oracle import ltx

    torch.manual_seed(0)
    up = LTXVideoTransformer3DModel(in_channels=8, out_channels=8, num_attention_heads=4, attention_head_dim=8, cross_attention_dim=32, num_layers=1,
                                    caption_channels=32).float()
    om = ltx.LTXVideoTransformer3DModel(ltx.LTXConfig.dummy()).float()
    load_into_oracle(om, up, lambda k: k)
    g = torch.Generator().manual_seed(1)
    B, F_, H, W, T = 2, 2, 4, 4, 6
    x = torch.randn(B, F_ * H * W, 8, generator=g)
    text, mask = torch.randn(B, T, 32, generator=g), torch.tensor([[1] * 4 + [0] * 2, [1] * 6])
    t = torch.tensor([310.0, 840.0]).view(B, 1, 1).expand(B, F_ * H * W, 1).long()
    kw = dict(hidden_states=x, encoder_hidden_states=text, timestep=t, encoder_attention_mask=mask, num_frames=F_, height=H, width=W,
              rope_interpolation_scale=[1 / (8 / 25), 8, 8], return_dict=False)
    return compare("LTX-Video", om(**kw)[0], up(**kw)[0], om, up, lambda k: k)


def run_cogvideox():
    from diffusers import CogVideoXTransformer3DModel

    from oracle import cogvideox as cvx

    cfg = cvx.CogVideoXConfig.dummy()
    torch.manual_seed(0)
    up = CogVideoXTransformer3DModel(num_attention_heads=4, attention_head_dim=16, in_channels=4, out_channels=4, time_embed_dim=2, text_embed_dim=32,
                                     num_layers=2, sample_width=24, sample_height=24, sample_frames=9, patch_size=2, temporal_compression_ratio=4,
                                     max_text_seq_length=16, use_rotary_positional_embeddings=True).float()
    om = cvx.CogVideoXTransformer3DModel(cfg).float()
    load_into_oracle(om, up, ff_names)
    g = torch.Generator().manual_seed(1)
    B, F_, H, W = 2, 3, 6, 6
    x = torch.randn(B, F_, 4, H, W, generator=g)
    text = torch.randn(B, 16, 32, generator=g)
    rope = cvx.prepare_rotary_positional_embeddings(H * 8, W * 8, F_, 8, 2, None, cfg.attention_head_dim, cfg.sample_width, cfg.sample_height)
    kw = dict(hidden_states=x, encoder_hidden_states=text, timestep=torch.tensor([310, 840]), image_rotary_emb=rope, return_dict=False)
    rc = compare("CogVideoX", om(**kw)[0], up(**kw)[0], om, up, ff_names)
    # the 1.5 architecture: patches over two latent frames (Linear patch embedding without bias), ofs embedding, integer-position rotary tables
    import dataclasses

    cfg15 = dataclasses.replace(cfg, patch_size_t=2, ofs_embed_dim=2, patch_bias=False)
    torch.manual_seed(0)
    up15 = CogVideoXTransformer3DModel(num_attention_heads=4, attention_head_dim=16, in_channels=4, out_channels=4, time_embed_dim=2, text_embed_dim=32,
                                       num_layers=2, sample_width=24, sample_height=24, sample_frames=9, patch_size=2, patch_size_t=2, patch_bias=False,
                                       ofs_embed_dim=2, temporal_compression_ratio=4, max_text_seq_length=16, use_rotary_positional_embeddings=True).float()
    om15 = cvx.CogVideoXTransformer3DModel(cfg15).float()
    load_into_oracle(om15, up15, ff_names)
    x15 = torch.randn(B, 4, 4, H, W, generator=g)
    rope15 = cvx.prepare_rotary_positional_embeddings(H * 8, W * 8, 4, 8, 2, 2, cfg.attention_head_dim, cfg.sample_height * 8, cfg.sample_width * 8)
    kw15 = dict(hidden_states=x15, encoder_hidden_states=text, timestep=torch.tensor([310, 840]), image_rotary_emb=rope15, ofs=torch.full((B,), 2.0), return_dict=False)
    rc15 = compare("CogVideoX 1.5", om15(**kw15)[0], up15(**kw15)[0], om15, up15, ff_names)
    return max(rc[0], rc15[0]), max(rc[1], rc15[1])


def run_wan():
    from diffusers import WanTransformer3DModel

    from oracle import wan

    torch.manual_seed(0)
    up = WanTransformer3DModel(patch_size=(1, 2, 2), num_attention_heads=2, attention_head_dim=12, in_channels=16, out_channels=16, text_dim=32, freq_dim=256,
                               ffn_dim=32, num_layers=2, cross_attn_norm=True, qk_norm="rms_norm_across_heads", rope_max_seq_len=32).float()
    om = wan.WanTransformer3DModel(wan.WanConfig.dummy()).float()
    load_into_oracle(om, up, ff_names)
    g = torch.Generator().manual_seed(1)
    x, text = torch.randn(2, 16, 3, 4, 6, generator=g), torch.randn(2, 5, 32, generator=g)
    kw = dict(hidden_states=x, timestep=torch.tensor([310, 840]), encoder_hidden_states=text, return_dict=False)
    return compare("Wan", om(**kw)[0], up(**kw)[0], om, up, ff_names)


def run_hunyuan():
    from diffusers import HunyuanVideoTransformer3DModel

    from oracle import hunyuan as hy

    torch.manual_seed(0)
    up = HunyuanVideoTransformer3DModel(in_channels=4, out_channels=4, num_attention_heads=2, attention_head_dim=10, num_layers=2, num_single_layers=2,
                                        num_refiner_layers=1, patch_size=1, patch_size_t=1, guidance_embeds=True, text_embed_dim=16, pooled_projection_dim=8,
                                        rope_axes_dim=(2, 4, 4)).float()
    om = hy.HunyuanVideoTransformer3DModel(hy.HunyuanVideoConfig.dummy()).float()
    load_into_oracle(om, up, hunyuan_names)
    g = torch.Generator().manual_seed(1)
    x, text = torch.randn(2, 4, 3, 4, 6, generator=g), torch.randn(2, 5, 16, generator=g)
    kw = dict(hidden_states=x, timestep=torch.tensor([300, 800]), encoder_hidden_states=text, encoder_attention_mask=torch.tensor([[1, 1, 1, 0, 0], [1] * 5]),
              pooled_projections=torch.randn(2, 8, generator=g), guidance=torch.tensor([6000.0, 6000.0]), return_dict=False)
    return compare("HunyuanVideo", om(**kw)[0], up(**kw)[0], om, up, hunyuan_names)


if __name__ == "__main__":
    runners = {"ltx": run_ltx, "cogvideox": run_cogvideox, "wan": run_wan, "hunyuan": run_hunyuan}
    which = [a for a in sys.argv[1:] if a in runners] or list(runners)
    print(f"diffusers {diffusers.__version__}")
    bad = []
    for w in which:
        try:
            o, gdiff = runners[w]()
            if o > 1e-4 or gdiff > 1e-4:
                bad.append(w)
        except Exception as e:  # a key that does not map or a signature that moved is itself a finding
            print(f"{w}: FAILED to run -- {type(e).__name__}: {e}")
            bad.append(w)
    sys.exit(1 if bad else 0)
