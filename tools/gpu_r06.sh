#!/bin/bash
# Round-6 GPU visits (run through gpurun from the repo root).  usage: tools/gpu_r06.sh <letter> [args]
set -u
mkdir -p gpurun_out
case "${1:-}" in
  t)  # phase timeline of the tiled NT GEMMs inside the step and stand-alone (trace build of the library)
    tag="${2:-1}"
    FTMI_LIB_PATH=finetrainers_amd/libftmi355_trace.so timeout 600 python tools/nt_trace.py gpurun_out/r06_nt_trace_$tag --lab > gpurun_out/r06_nt_trace_$tag.log 2>&1
    echo "exit $?" >> gpurun_out/r06_nt_trace_$tag.log
    tail -70 gpurun_out/r06_nt_trace_$tag.log
    rm -f gpurun_out/r06_nt_trace_${tag}_step.bin gpurun_out/r06_nt_trace_${tag}_lab.bin
    ;;
  g)  # gemm lab: stand-alone NT kernels (plain store), "shapes" "variants" [env...]
    tag="${2:-1}"; shapes="${3:-5376x8192x2048}"; vars="${4:-80,87}"
    timeout 600 tools/bin/gemm_lab "$shapes" "$vars" > gpurun_out/r06_gemm_lab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r06_gemm_lab_$tag.txt
    cat gpurun_out/r06_gemm_lab_$tag.txt
    ;;
  b)  # C-ABI A/B of GEMM variants with epilogues / LoRA extension: tag "variants" [SHAPES] [EPI] [LORA]
    tag="${2:-1}"; vars="${3:-8,80,87}"
    SHAPES="${4:-5376x8192x2048}" EPI="${5:-store}" LORA="${6:-0}" VENDOR=0 timeout 600 python tools/bench_gemm_ab.py "$vars" > gpurun_out/r06_gemm_ab_$tag.txt 2>&1
    echo "exit $?" >> gpurun_out/r06_gemm_ab_$tag.txt
    cat gpurun_out/r06_gemm_ab_$tag.txt
    ;;
  s)  # in-step A/B of an environment switch: tag VAR "v1 v2" [rounds]
    tag="${2:-1}"; var="$3"; vals="$4"; rounds="${5:-2}"
    bash tools/ab_env.sh "$var" "$vals" "$rounds" > gpurun_out/r06_instep_ab_$tag.txt 2>&1
    cat gpurun_out/r06_instep_ab_$tag.txt
    ;;
  l)  # in-step A/B of two library builds: tag "libA libB" [rounds]
    tag="${2:-1}"; libs="${3:-finetrainers_amd/libftmi355_prev.so finetrainers_amd/libftmi355.so}"; rounds="${4:-2}"
    bash tools/ab_lib.sh "$libs" "$rounds" > gpurun_out/r06_lib_ab_$tag.txt 2>&1
    cat gpurun_out/r06_lib_ab_$tag.txt
    ;;
  k)  # GPU kernel tests (GEMM + attention files)
    timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -15
    ;;
  *) echo "unknown visit"; exit 1;;
esac
