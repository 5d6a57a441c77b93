"""Calibration only: which vendor-library GEMM kernels run for the LTX step shapes (names encode tile/wave configuration)."""
import torch, math
dev = torch.device("cuda", 0)
for (M, N, K) in [(5376, 2048, 2048), (5376, 6144, 2048), (5376, 8192, 2048), (5376, 2048, 8192), (4096, 4096, 4096), (8192, 8192, 8192)]:
    x = torch.randn((M, K), device=dev).to(torch.bfloat16)
    w = (torch.randn((N, K), device=dev) / math.sqrt(K)).to(torch.bfloat16)
    for _ in range(5):
        torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
