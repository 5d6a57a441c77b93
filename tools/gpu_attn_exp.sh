#!/bin/bash
# attention A/B (experimental build); writes gpurun_out/<tag>_attn.txt
tag=${1:-r02}
mkdir -p gpurun_out
{
echo "== product build (attention.hip with -amdgpu-mfma-vgpr-form)"
python tools/bench_attn.py fwd "FTMI_ATTN_GEN=1" "FTMI_ATTN_GEN=1,FTMI_ATTN_FWD=16"
python tools/bench_attn.py bwd "FTMI_ATTN_GEN=1"
echo "== alt build (compiler's choice: accumulators in AGPRs)"
FTMI_LIB_PATH=$PWD/finetrainers_amd/libftmi355_alt.so python tools/bench_attn.py fwd "FTMI_ATTN_GEN=1" "FTMI_ATTN_GEN=1,FTMI_ATTN_FWD=16"
FTMI_LIB_PATH=$PWD/finetrainers_amd/libftmi355_alt.so python tools/bench_attn.py bwd "FTMI_ATTN_GEN=1"
} > gpurun_out/${tag}_attn.txt 2>&1
cat gpurun_out/${tag}_attn.txt
