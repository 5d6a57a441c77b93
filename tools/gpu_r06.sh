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
  e)  # evidence run: default bench line (with the CPU baseline), the driver's command, --no-prof, the checkpointing line, rocprofv3 stats + counter passes
    python bench.py > gpurun_out/r06_bench_default.json 2> gpurun_out/r06_bench_default.err; echo "bench default rc=$?"; cut -c1-400 gpurun_out/r06_bench_default.json
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r06_bench_driver_cmd.json 2>/dev/null; cut -c1-300 gpurun_out/r06_bench_driver_cmd.json
    python bench.py --no-prof --no-cpu-baseline > gpurun_out/r06_bench_noprof.json 2>/dev/null; cut -c1-300 gpurun_out/r06_bench_noprof.json
    python bench.py --no-cpu-baseline --gradient-checkpointing > gpurun_out/r06_bench_ckpt.json 2>/dev/null; cut -c1-300 gpurun_out/r06_bench_ckpt.json
    bash tools/gpu_profile_r06.sh r06 2>&1 | tail -70
    ;;
  w)  # the other three workloads, plain bench lines (after the last kernel change)
    for wl in cogvideox wan hunyuan; do
      python bench.py --workload $wl --no-cpu-baseline > gpurun_out/r06_${wl}_bench.json 2> gpurun_out/r06_${wl}_bench.err; echo "$wl rc=$?"; cut -c1-330 gpurun_out/r06_${wl}_bench.json
    done
    ;;
  u)  # the whole -m gpu suite, one process per file
    shift
    bash tools/gpu_suite.sh r06 "$@" 2>&1 | tail -60
    ;;
  p)  # full-depth parity at config 2 (28 blocks, batch 2) against the bf16 oracle and the committed fp32 sample
    timeout 2400 python -m pytest tests/test_gpu_dit.py -m gpu -q -s -k "full_depth" -p no:cacheprovider > gpurun_out/r06_parity_full_cfg2.log 2>&1
    echo "rc=$?"; grep -E "parity|passed|failed|rel|floor" gpurun_out/r06_parity_full_cfg2.log | tail -20
    ;;
  *) echo "unknown visit"; exit 1;;
esac
