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
  f)  # attention lab, forward: pipelined forward (FTMI_ATTN_PL bit 2) vs attn_fwd_kernel, bit compare + timing; also with keys that grow along the sequence (rare rescale path)
    tag="${2:-1}"; shapes="${3:-2x32x2688}"; cfgs="${4:-0,0x4}"
    LAB_FWD=1 LAB_FWD_ONLY=1 timeout 300 tools/bin/attn_lab "$shapes" "$cfgs" > gpurun_out/r05_attn_fwd_lab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r05_attn_fwd_lab_$tag.txt
    echo "# LAB_GROW=1 (keys grow along the sequence)" >> gpurun_out/r05_attn_fwd_lab_$tag.txt
    LAB_GROW=1 LAB_FWD=1 LAB_FWD_ONLY=1 timeout 300 tools/bin/attn_lab "$shapes" "$cfgs" >> gpurun_out/r05_attn_fwd_lab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r05_attn_fwd_lab_$tag.txt
    cat gpurun_out/r05_attn_fwd_lab_$tag.txt
    ;;
  h)  # attention lab at head_dim 128: the fused pipelined dK / dV kernel (FTMI_ATTN_PL bit 3) vs the two-pass kernels, bit compare + timing
    tag="${2:-1}"; shapes="${3:-1x12x4096,1x4x1000,1x12x21504}"; cfgs="${4:-0,0x8}"
    LAB_D=128 LAB_ONLY=02 timeout 600 tools/bin/attn_lab "$shapes" "$cfgs" > gpurun_out/r05_attn128_lab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r05_attn128_lab_$tag.txt
    cat gpurun_out/r05_attn128_lab_$tag.txt
    ;;
  p)  # the MFMA / VALU issue probe of round 2, whole output
    hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu.hip -o /tmp/probe_mfma_valu && timeout 120 /tmp/probe_mfma_valu > gpurun_out/r05_probe_mfma_valu.txt 2>&1
    cat gpurun_out/r05_probe_mfma_valu.txt
    ;;
  s)  # skinny lab: LoRA down-projection kernels
    tag="${2:-1}"; shapes="${3:-5376x2048x64,5376x2048x192,2688x2048x64}"; cfgs="${4:-0,1}"
    timeout 300 tools/bin/skinny_lab "$shapes" "$cfgs" > gpurun_out/r05_skinny_lab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r05_skinny_lab_$tag.txt
    cat gpurun_out/r05_skinny_lab_$tag.txt
    ;;
  e)  # evidence run: default bench line (with the CPU baseline), the driver's command, --no-prof, rocprofv3 stats + counter passes of the same command
    python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; echo "bench default rc=$?"; cut -c1-600 gpurun_out/r05_bench_default.json
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r05_bench_driver_cmd.json 2>/dev/null; cut -c1-300 gpurun_out/r05_bench_driver_cmd.json
    python bench.py --no-prof --no-cpu-baseline > gpurun_out/r05_bench_noprof.json 2>/dev/null; cut -c1-300 gpurun_out/r05_bench_noprof.json
    bash tools/gpu_profile_r05.sh r05 2>&1 | tail -60
    ;;
  *) echo "unknown visit"; exit 1;;
esac
