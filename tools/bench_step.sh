#!/bin/bash
# usage: bench_step.sh [extra bench args]; prints ms/step and per-kernel-class ms
python -m finetrainers_amd.csrc.build >/dev/null 2>&1 || { echo BUILD FAILED; exit 1; }
python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f  samples/s %.2f  mfma_util %.3f'%(d['ms_per_step'], d['value'], d['mfma_utilisation_step']))
for k,v in d.get('kernels',{}).items(): print('  %-9s %7.2f ms/step  %6.1f us avg  %7.1f TF/s  (%d launches)'%(k, v['ms_per_step'], v['avg_us'], v['tflops'], v['launches_per_step']))
"
