#!/bin/bash
# One GPU-box cycle (run through gpurun from the repo root):  bash tools/gpu_cycle.sh <tag> [steps...]
# steps: bf16tests alltests gemmbench bench benchfull stress   (default: alltests bench)
# A step that fails its assertions does not stop the cycle; a step that TIMES OUT or is killed does (nothing further touches the GPU).
set -o pipefail
TAG=${1:-cycle}; shift
STEPS=${@:-alltests bench}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
run() {  # run <name> <seconds> <cmd...>
  local name=$1 secs=$2; shift 2
  echo "== $name: $*" | tee -a $OUT/cycle.log
  timeout -k 10 $secs "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/cycle.log
  tail -n 6 $OUT/$name.log
  [ $rc -lt 124 ]
}
for s in $STEPS; do
  case $s in
    bf16tests) run bf16tests 900 python3 -m pytest tests/test_gpu_bf16.py -x -q -m gpu -s || exit 1 ;;
    berttests) run berttests 900 python3 -m pytest tests/test_gpu_bert.py tests/test_gpu_bf16.py tests/test_gpu_dropin.py tests/test_gpu_loaders.py tests/test_gpu_api_shim.py -x -q -m gpu || exit 1 ;;
    alltests)  run alltests 1100 python3 -m pytest tests -x -q -m gpu || exit 1 ;;
    gemmbench) run gemmbench 300 python3 tools/gemm_bf16_bench.py || exit 1 ;;
    phasesab)  run phasesab 300 python3 tools/gemm_bf16_phases_ab.py || exit 1 ;;
    stress)    run stress 400 python3 tools/gemm_bf16_stress.py 200 noise || exit 1 ;;
    bench)     run bench 400 python3 bench.py --steps 2 --warmup 1 --no-cpu || exit 1 ;;
    benchfull) run benchfull 600 python3 bench.py || exit 1 ;;
    bertprof)  rm -rf /tmp/bp_$TAG; run bertprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bp_$TAG -- python3 tools/bert_prof.py bf16 || exit 1
               f=$(find /tmp/bp_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bert_bf16_kernel_stats.csv && python3 tools/kstats.py $f "" 14
               t=$(find /tmp/bp_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$t" ] && python3 tools/bert_layer_times.py $t > $OUT/bert_layer_times.txt ;;
    benchprof) rm -rf /tmp/bn_$TAG; run benchprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bn_$TAG -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extra || exit 1
               f=$(find /tmp/bn_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/bench_kernel_stats.csv && python3 tools/kstats.py $f "" 14 ;;
    bertab)    run bertab 300 python3 tools/bert_ab.py bf16_gemm_tail 0 1 2 || exit 1 ;;
    f16tests)  run f16tests 600 python3 -m pytest tests/test_gpu_f16.py -x -q -m gpu -s || exit 1 ;;
    stamps)    run stamps 300 python3 tools/gemm_bf16_stamps.py || exit 1 ;;
    stampsame) run stampsame 300 python3 tools/gemm_bf16_stamps.py same_tile || exit 1 ;;
    attnctx)   rm -rf /tmp/ac_$TAG; run attnctx 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/ac_$TAG -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0 || exit 1
               f=$(find /tmp/ac_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/attn_vs_ctx.py $f > $OUT/attn_vs_ctx.txt && python3 tools/kernels_vs_ctx.py $f > $OUT/kernels_vs_ctx.txt; head -30 $OUT/attn_vs_ctx.txt ;;
    bertrev)   run bertrev 300 python3 tools/bert_ab.py bf16_gemm_reverse 0 1 || exit 1 ;;
    berttrace) rm -rf /tmp/bt_$TAG; run berttrace 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/bt_$TAG -- python3 tools/bert_prof.py bf16 || exit 1
               f=$(find /tmp/bt_$TAG -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python3 tools/bert_layer_times.py $f > $OUT/bert_layer_times.txt; cat $OUT/bert_layer_times.txt | head -70 ;;
    prefillab) run prefillab 300 python3 tools/prefill_ab.py decoder_prefill16_pages 0 1 || exit 1 ;;
    attnwide)  run attnwide 300 python3 tools/prefill_ab.py attn16_wide 0 1 || exit 1 ;;
    attnstamps) run attnstamps 300 python3 tools/attn_stamps.py || exit 1 ;;
    bertph)    run bertph 300 python3 tools/bert_ab.py bf16_gemm_phases 4 2 1 || exit 1 ;;
    newtests)  run newtests 900 python3 -m pytest tests/test_gpu_decoder.py tests/test_gpu_ops.py tests/test_gpu_bf16.py tests/test_gpu_bert.py -x -q -m gpu -s -k "long or half_tile_tail_whose or odd_batch or lm_head or device_resident" || exit 1 ;;
    prefillprof) rm -rf /tmp/pp_$TAG; run prefillprof 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$TAG -- python3 tools/prefill_bench.py ${PREFILL_DTYPE:-f32} logits || exit 1
               f=$(find /tmp/pp_$TAG -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/prefill_${PREFILL_DTYPE:-f32}_kernel_stats.csv && python3 tools/kstats.py $f "" 14 ;;
    attn16ab)  run attn16ab 300 python3 tools/attn16_ab.py ${AB_SWITCH:-attn16_pipe} ${AB_VALUES:-0 1} || exit 1 ;;
    attntests) run attntests 600 python3 -m pytest tests/test_gpu_bf16.py tests/test_gpu_f16.py -x -q -m gpu -k "attention or prefill" || exit 1 ;;
    sampler)   run samplertests 600 python3 -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "sampler or topk" || exit 1
               MGEA_SAMPLER_WAVE_SELECT=0 run sampler_old 200 python3 tools/sampler_bench.py || exit 1
               run sampler_new 200 python3 tools/sampler_bench.py || exit 1 ;;
    splittests) run splittests 900 python3 -m pytest tests/test_gpu_decoder.py -x -q -m gpu -s -k "split_context or full_context or single_stream or fused_path_step or generate_vs_oracle" || exit 1 ;;
    b1)        MGEA_ATTN_SPLIT=0 run b1_old 300 python3 tools/b1_prof.py || exit 1
               run b1_new 300 python3 tools/b1_prof.py || exit 1 ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
