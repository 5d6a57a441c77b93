#!/bin/bash
# attention ablations / variants (experimental build) + atomic probe; writes gpurun_out/<tag>_attn.txt
tag=${1:-r02}
mkdir -p gpurun_out
{ python tools/bench_attn.py fwd 0 1 2 3 4 5 6 7 8 10 11 12 13 14 15; tools/probe_atomic; } > gpurun_out/${tag}_attn.txt 2>&1
cat gpurun_out/${tag}_attn.txt
