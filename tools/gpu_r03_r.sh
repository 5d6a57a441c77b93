#!/bin/bash
# same-box A/B: Wan and HunyuanVideo steps with the blocks issued from Python (FTMI_NATIVE_BLOCKS=0) and as C calls
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
for wl in wan hunyuan; do
  st=6; [ $wl = hunyuan ] && st=3
  for nb in 0 1 0 1; do
    echo -n "$wl FTMI_NATIVE_BLOCKS=$nb  "
    FTMI_NATIVE_BLOCKS=$nb timeout 600 python bench.py --workload $wl --steps $st --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.1f'%d['ms_per_step'], 'min/med/max', d.get('step_ms_min_median_max'), 'peak GiB %.1f'%d['peak_memory_gib'])"
    [ $wl = hunyuan ] && [ $nb = 1 ] && break
  done
done > $O/r03r_ab.log 2>&1
cat $O/r03r_ab.log
