#!/bin/bash
# per-kernel time of the LoRA weight-gradient launches for settings of FTMI_TN_WIDE / FTMI_TN_XCD (rocprofv3 --kernel-trace --stats of a short bench run)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  rm -rf /tmp/tnprof
  FTMI_TN_WIDE=$1 FTMI_TN_XCD=$2 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tnprof -o tn -- python $R/bench.py --steps 4 --warmup 1 --no-prof --no-cpu-baseline > /tmp/tn.log 2>&1
  echo "== FTMI_TN_WIDE=$1 FTMI_TN_XCD=$2  rc=$?"
  grep "gemm_tn" /tmp/tnprof/tn_kernel_stats.csv | cut -c1-200 || tail -5 /tmp/tn.log
done
