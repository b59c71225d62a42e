#!/bin/bash
# All rocprofv3 evidence of a round in one GPU-box call (run from the repo root through gpurun):
#   bash tools/profile_round.sh r2
# Writes gpurun_out/prof_<tag>/: kernel stats of the f32 headline bench, of the fp16 mode and of the bf16 DistilBERT forward;
# PMC FETCH_SIZE of the eager twin of the headline generation; PMC MFMA utilisation of the DistilBERT GEMMs; and ONE attempt
# at --pmc under hipGraph replay with its output kept (it aborted in round 1; evidence, not a retry loop).
# Counter passes carry --pmc only (no trace domains): the pool refuses the combination.
set -o pipefail
TAG=${1:-r2}
OUT=gpurun_out/prof_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
T=/tmp/prof_$TAG; rm -rf $T; mkdir -p $T
stats() { f=$(find $1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $2; }
pmc() { f=$(find $1 -name "*counter_collection.csv" | head -1); echo $f; }

timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-extra > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err && stats $T/bench $OUT/bench_kernel_stats.csv
echo "bench kernel stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/f16 -- python3 tools/f16_prof.py both > $OUT/f16_prof.txt 2>&1 && stats $T/f16 $OUT/f16_kernel_stats.csv
echo "f16 kernel stats rc=$?"; cat $OUT/f16_prof.txt | grep tokens
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $T/bert -- python3 tools/bert_prof.py bf16 > $OUT/bert_bf16_prof.txt 2>&1 && stats $T/bert $OUT/bert_bf16_kernel_stats.csv
echo "bert kernel stats rc=$?"
MGEA_DECODER_NOGRAPH=1 timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0 > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch_bench.err
echo "pmc fetch rc=$?"
f=$(pmc $T/pmc_fetch); [ -n "$f" ] && python3 tools/pmc_traffic.py $f FETCH_SIZE > $OUT/pmc_fetch_size_full.json
timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_BF16 --output-format csv -d $T/pmc_mfma -- python3 tools/bert_prof.py bf16 > $OUT/pmc_mfma_prof.txt 2>&1
echo "pmc mfma rc=$?"
f=$(pmc $T/pmc_mfma); [ -n "$f" ] && python3 tools/pmc_mfma.py $f gemm > $OUT/pmc_mfma_util.json
# one attempt, output kept: --pmc with the decode step replayed from the hipGraph (aborted in round 1)
timeout -k 10 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/pmc_graph -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0 --total-len 72 > $OUT/pmc_under_graph_replay.out 2> $OUT/pmc_under_graph_replay.err
echo "pmc under graph replay rc=$?" | tee $OUT/pmc_under_graph_replay.rc
# the full-length generation under graph replay with --pmc (only meaningful if the short attempt above returned 0)
if grep -q "rc=0" $OUT/pmc_under_graph_replay.rc; then
  timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $T/pmc_graph_full -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0 > $OUT/pmc_graph_full_bench.json 2> $OUT/pmc_graph_full_bench.err
  echo "pmc under graph replay, full length rc=$?" | tee -a $OUT/pmc_under_graph_replay.rc
  f=$(pmc $T/pmc_graph_full); [ -n "$f" ] && python3 tools/pmc_traffic.py $f FETCH_SIZE > $OUT/pmc_fetch_size_graph_replay.json
fi
[ -x tools/micro/kernel_floor ] && (timeout -k 5 60 tools/micro/kernel_floor 256; timeout -k 5 60 tools/micro/kernel_floor 128) 2>&1 | grep -v amdgpu.ids > $OUT/kernel_floor.txt
ls -la $OUT
