# in-step A/B of an environment switch: ab_env.sh VAR "v1 v2 ..." [rounds]
VAR=$1; V=$2; R=${3:-2}
for r in $(seq $R); do for v in $V; do echo -n "$VAR=$v  "; env $VAR=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], 'min/med/max', d.get('step_ms_min_median_max'), ' '.join('%s %.2f'%(k,v['ms_per_step']) for k,v in d['kernels'].items()))"; done; done
