#!/bin/bash
# Round-5 GPU visits (run through gpurun from the repo root).  usage: tools/gpu_r05.sh <letter> [args]
#   a  attention lab: hand-placed backward pipelines vs the compiler-scheduled kernels (bit compare + timing)      -> gpurun_out/r05_attn_lab_<tag>.txt
set -u
mkdir -p gpurun_out
case "${1:-}" in
  a)
    tag="${2:-1}"; shapes="${3:-2x32x2688}"; cfgs="${4:-0,0x01,0x11,0x21}"
    timeout 600 tools/bin/attn_lab "$shapes" "$cfgs" > gpurun_out/r05_attn_lab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r05_attn_lab_$tag.txt
    tail -40 gpurun_out/r05_attn_lab_$tag.txt
    ;;
  *) echo "unknown visit"; exit 1;;
esac
