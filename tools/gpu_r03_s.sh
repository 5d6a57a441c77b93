#!/bin/bash
# CogVideoX 1.5 architecture (patch_size_t, ofs) against the oracle, next to the 2b / 5b cases of the same test
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 900 python -m pytest tests/test_gpu_cogvideox.py -q -x -s -k "model_step_parity and (1.5 or True-2-True or False-2-True)" > $O/r03s_cog.log 2>&1; echo "cog rc=$?"
grep -n "cog-model\|cog-step\|passed\|failed\|Error" $O/r03s_cog.log | tail -n 14 | cut -c1-300
