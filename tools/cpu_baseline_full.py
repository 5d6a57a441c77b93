"""The CPU baseline of SURVEY 8(d) measured ONCE at full size instead of extrapolated: BASELINE configs[1] exactly (batch 2, 49x512x768 clip =
2 x 2688 tokens, all 28 blocks, LoRA rank 64, bf16 like the reference), the oracle's SFT step (forward + backward + clip + AdamW) on all host
threads, 1 warm-up + 2 timed steps (~6 min on the gpurun box's 128 threads).  Writes profiles/r04_cpu_baseline_full.json, which bench.py quotes next
to its scaled live sample ("measured_full": true).   python tools/cpu_baseline_full.py [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from oracle import ltx  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r04_cpu_baseline_full.json")
torch.manual_seed(0)
t0 = time.time()
dt, ts = bench._cpu_steps(ltx, ltx.LTXConfig.production(num_layers=28), 64, torch.bfloat16, 2, 7, 16, 24, 1, 2)
res = {"workload": "LTX-Video LoRA rank=64 bf16, 49x512x768, batch 2, 28 blocks (BASELINE configs[1])", "kind": "port", "cores": torch.get_num_threads(),
       "step_s_measured": ts, "step_s": dt, "samples_per_s": 2.0 / dt, "warmup_steps": 1, "timed_steps": 2, "wall_s_total": time.time() - t0,
       "what": "oracle/ltx.py sft_step (CPU restatement of the reference SFTTrainer step: forward, weighted MSE, backward, clip, AdamW), no activation checkpointing"}
with open(out_path, "w") as f:
    json.dump(res, f, indent=1)
print(json.dumps(res))
