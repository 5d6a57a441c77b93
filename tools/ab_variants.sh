# in-step A/B of NT GEMM K-loop variants (one box, interleaved); usage: ab_variants.sh "v1 v2 ..." [rounds]
V=${1:-"30 36 37 38 39"}; R=${2:-2}
for r in $(seq $R); do for v in $V; do echo -n "NT192=$v  "; FTMI_NT192=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], ' gemm_nt %.2f ms'%d['kernels']['gemm_nt']['ms_per_step'])"; done; done
