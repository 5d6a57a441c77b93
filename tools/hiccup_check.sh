# how often does a 10-step bench run contain a stalled step?  usage: hiccup_check.sh [runs] [extra bench args]
N=${1:-8}; shift
for i in $(seq $N); do python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], d.get('step_ms_min_median_max'), d.get('step_ms_outliers'))"; done
