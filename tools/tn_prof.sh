#!/bin/bash
# per-kernel time of the LoRA weight-gradient launches for settings of an environment switch (rocprofv3 --kernel-trace --stats of a short bench run)
#   tools/tn_prof.sh                      -> FTMI_TN_WIDE x FTMI_TN_XCD (profiles/r05_tn_wgrad.txt)
#   tools/tn_prof.sh FTMI_TN_TARGET_WGS "56 128 448"
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {
  rm -rf /tmp/tnprof
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tnprof -o tn -- python $R/bench.py --steps 4 --warmup 1 --no-prof --no-cpu-baseline > /tmp/tn.log 2>&1
  echo "== $*  rc=$?"
  grep "gemm_tn" /tmp/tnprof/tn_kernel_stats.csv | cut -c1-200 || tail -5 /tmp/tn.log
}
if [ -n "${1:-}" ]; then
  for v in $2; do run $1=$v; done
else
  for cfg in "0 0" "1 0" "0 1" "1 1"; do set -- $cfg; run FTMI_TN_WIDE=$1 FTMI_TN_XCD=$2; done
fi
