#!/bin/bash
# One gpurun call: probes + GPU parity tests, everything logged under gpurun_out/.
mkdir -p gpurun_out
export FTMI_REPORT_DIR=gpurun_out
python -m finetrainers_amd.csrc.build > gpurun_out/build.log 2>&1 || { echo BUILD FAILED; tail -20 gpurun_out/build.log; exit 1; }
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -6 > gpurun_out/rocminfo.txt
nproc > gpurun_out/nproc.txt
timeout 60 ./tools/probe_tr16 > gpurun_out/probe_tr16.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -400 > gpurun_out/test_kernels.log
echo "kernels exit: ${PIPESTATUS[0]}" >> gpurun_out/test_kernels.log
timeout 1500 python -m pytest tests/test_gpu_dit.py -m gpu -q -s -p no:cacheprovider 2>&1 | tail -400 > gpurun_out/test_dit.log
echo "dit exit: ${PIPESTATUS[0]}" >> gpurun_out/test_dit.log
grep -E "passed|failed|error" gpurun_out/test_kernels.log | tail -3
grep -E "passed|failed|error" gpurun_out/test_dit.log | tail -3
