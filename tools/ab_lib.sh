# in-step A/B of two builds of the library (interleaved, one box): ab_lib.sh "finetrainers_amd/libftmi355_prev.so finetrainers_amd/libftmi355.so" [rounds] [extra env]
LIBS=$1; R=${2:-2}
for r in $(seq $R); do for l in $LIBS; do echo -n "$l  "; env FTMI_LIB_PATH=$l ${3:-} python bench.py --steps 12 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], 'min/med/max', d.get('step_ms_min_median_max'), ' '.join('%s %.2f'%(k,v['ms_per_step']) for k,v in d['kernels'].items()))"; done; done
