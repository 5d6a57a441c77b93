import subprocess, sys
CASE = '''
import torch, math, sys
sys.path.insert(0, '.')
from finetrainers_amd import _lib, ops
dev = torch.device('cuda:0'); bf16 = torch.bfloat16
M, N, K, v = %d, %d, %d, %d
g = torch.Generator().manual_seed(5)
x = (torch.randn((M, K), generator=g)).to(bf16).to(dev); w = (torch.randn((N, K), generator=g) / math.sqrt(K)).to(bf16).to(dev)
b = torch.randn((N,), generator=g).to(bf16).to(dev)
A = (torch.randn(64, K, generator=g) / math.sqrt(K)).to(dev); Bm = (torch.randn(N, 64, generator=g) * 0.05).to(dev)
ref = ops.linear_lora_fwd(x, w, b, A, Bm, 0.5, variant=%d)[0]
out = ops.linear_lora_fwd(x, w, b, A, Bm, 0.5, variant=v)[0]
torch.cuda.synchronize()
print("equal" if torch.equal(out, ref) else "DIFF %%d" %% int((out != ref).sum()))
'''
for (M, N, K) in [(1024, 256, 256), (1024, 256, 320), (5376, 2048, 2048)]:
    for v, base in ((1386, 86), (1286, 86), (2286, 86), (1387, 87), (1287, 87), (1380, 80)):
        r = subprocess.run([sys.executable, "-c", CASE % (M, N, K, v, base)], capture_output=True, text=True)
        tail = (r.stdout.strip().splitlines() or ["-"])[-1]
        err = "FAULT" if "Memory access fault" in r.stderr or r.returncode < 0 else ("rc %d %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:100]) if r.returncode else "")
        print(M, N, K, v, tail, err, flush=True)
