for r in 1 2; do
for cfg in "0 0.05" "33 -0.01" "5 -0.01" "31 -0.01"; do set -- $cfg; echo -n "NT256=$1 THR=$2  "; FTMI_NT256=$1 FTMI_NT256_THR=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ms/step %.2f'%d['ms_per_step'], ' gemm_nt %.2f ms'%d['kernels']['gemm_nt']['ms_per_step'])"; done; done
