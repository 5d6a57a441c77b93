# interleaved in-step A/B of whole configurations: ab_cfg.sh rounds "ENV=.. ENV=.." "ENV=.." ...   (each argument = one configuration's environment; "-" = none)
R=$1; shift
for r in $(seq $R); do for c in "$@"; do [ "$c" = "-" ] && e="" || e="$c"; echo -n "[$c]  "; env $e python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], 'min/med/max', d.get('step_ms_min_median_max'), ' '.join('%s %.2f'%(k,v['ms_per_step']) for k,v in d['kernels'].items()))"; done; done
