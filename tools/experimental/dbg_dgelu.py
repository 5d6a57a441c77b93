import torch, math, sys
sys.path.insert(0, '.')
from finetrainers_amd import _lib, ops
dev = torch.device('cuda:0')
bf16 = torch.bfloat16
for (M, N, K) in [(300, 256, 512), (448, 256, 512), (5376, 8192, 2048)]:
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((M, K), generator=g)).to(bf16).to(dev); w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(bf16).to(dev)
    z = torch.randn((M, N), generator=g).to(bf16).to(dev); b = torch.randn((N,), generator=g).to(bf16).to(dev)
    for epi, kw in (("dgelu", dict(epilogue=_lib.EPI_DGELU, aux=z)), ("resid", dict(epilogue=_lib.EPI_RESID, resid=z)), ("store", {})):
        for bias in (None, b):
            ref = ops.gemm_nt(x, w, bias, variant=87, **kw)
            for v in (1387, 1287):
                out = ops.gemm_nt(x, w, bias, variant=v, **kw)
                bad = (out != ref)
                rows = bad.any(dim=1).nonzero().flatten().tolist()
                cols = bad.any(dim=0).nonzero().flatten().tolist()
                print(M, N, K, epi, "bias" if bias is not None else "nobias", v, "bad", int(bad.sum()), "rows", rows[:12], len(rows), "cols", cols[:6], cols[-3:], len(cols), flush=True)
