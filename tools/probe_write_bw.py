import torch
dev=torch.device("cuda",0)
for mb in (22, 88, 352):
    x=torch.empty(mb*1024*1024//2, dtype=torch.bfloat16, device=dev)
    for _ in range(5): x.fill_(1.0)
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): x.fill_(1.0)
    e.record(); torch.cuda.synchronize()
    us=s.elapsed_time(e)/20*1e3
    print(f"fill {mb} MB: {us:.1f} us  {mb*1.048576/us*1e3/1e3:.2f} TB/s")
