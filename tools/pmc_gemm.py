"""PMC target: a few launches of chosen NT GEMM variants on one shape.  usage: pmc_gemm.py M N K v1,v2,..."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from finetrainers_amd import ops
dev = torch.device("cuda", 0)
M, N, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
variants = [int(v) for v in sys.argv[4].split(",")]
x = torch.randn((M, K), device=dev).to(torch.bfloat16)
w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
for v in variants:
    for _ in range(6):
        ops.gemm_nt(x, w, None, variant=v)
torch.cuda.synchronize()
