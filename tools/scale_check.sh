#!/bin/bash
# First multi-GPU run on an 8 x MI355X node: the scaling lines of all four workloads and, per workload, how much of the gradient / parameter
# exchange is NOT hidden behind compute (rocprofv3 kernel trace of every rank; RCCL kernel time with no compute kernel running on that GPU).
#   bash tools/scale_check.sh [out_dir]        -- needs >= 2 visible GPUs; never run by the tests
# The traced runs are short (few steps); the untraced bench lines are the numbers to quote.
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-$R/gpurun_out/scale}
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
unset NCCL_P2P_DISABLE
NG=$(python -c "import torch; print(torch.cuda.device_count())")
echo "visible GPUs: $NG"
for wl in ltx cogvideox wan hunyuan; do
  for n in 1 2 4 8; do
    [ "$n" -le "$NG" ] || continue
    steps=10; [ "$wl" = ltx ] && steps=20
    python "$R/bench.py" --workload $wl --gpus $n --steps $steps --warmup 3 --no-cpu-baseline > "$OUT/${wl}_n${n}.json" 2> "$OUT/${wl}_n${n}.err"
    echo "$wl n=$n: $(python -c "import json,sys; d=json.loads(open('$OUT/${wl}_n${n}.json').read().strip().splitlines()[-1]); print(round(d['ms_per_step'],1),'ms/step', round(d['value'],3), d['unit'])" 2>/dev/null || echo FAILED)"
  done
  n=$NG; [ "$n" -ge 2 ] || continue
  # traced run on all ranks (kernel trace only -- never combined with counters), 3 + 3 steps
  cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/sc_$wl && mkdir -p /tmp/sc_$wl
  rocprofv3 --kernel-trace --output-format csv -d /tmp/sc_$wl -o t -- python "$R/bench.py" --workload $wl --gpus $n --steps 3 --warmup 3 --no-cpu-baseline --no-prof > "$OUT/${wl}_n${n}_traced.json" 2> "$OUT/${wl}_n${n}_traced.err"
  for f in $(find /tmp/sc_$wl -name "*kernel_trace.csv" | sort); do
    echo "$wl n=$n $(basename $(dirname $f))/$(basename $f): $(python "$R/tools/exposed_comm.py" "$f" 3 2>&1 | tr '\n' ' ')"
  done | tee "$OUT/${wl}_n${n}_exposed_comm.txt"
  cd "$R"
done
