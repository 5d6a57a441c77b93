# in-step A/B of the 256x256-tile choice: ab_nt256.sh "variant:thr ..." [rounds]
V=${1:-"0:0.05 47:0.05 47:-0.01"}; R=${2:-2}
for r in $(seq $R); do for c in $V; do v=${c%%:*}; t=${c##*:}; echo -n "NT256=$v THR=$t  "; FTMI_NT256=$v FTMI_NT256_THR=$t python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], ' gemm_nt %.2f ms'%d['kernels']['gemm_nt']['ms_per_step'])"; done; done
