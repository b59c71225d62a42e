#!/bin/bash
# FETCH_SIZE (HBM / fabric bytes into the L2s) of the decode step's skinny GEMMs, eager twin of the headline generation (VERDICT r3 #2c).
# rocprofv3 --pmc dies with SIGSEGV inside its dispatch interception of gemm_skinny_kernel<..., 8> (the FC2 launch, > 64 KB of dynamic
# LDS) -- profiles/r4_pmc_fetch_size_skinny_profiler_crash.err, four attempts -- so that instantiation is left out by the regex:
#   bash tools/pmc_skinny.sh <out dir> '<kernel regex>' <name>
set -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_skinny}; RE=${2:-gemm_skinny_kernel<.*, 2>}; NAME=${3:-skinny_nch2}
mkdir -p $OUT; rm -rf /tmp/pmc_$NAME
export MGEA_COMMIT=${MGEA_COMMIT:-unknown}
export MGEA_PMC_COMMAND="MGEA_DECODER_NOGRAPH=1 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex '$RE' --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0"
MGEA_DECODER_NOGRAPH=1 timeout -k 10 420 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "$RE" --output-format csv -d /tmp/pmc_$NAME -- python3 bench.py --steps 1 --warmup 0 --no-cpu --no-extra --profile-stride 0 > $OUT/$NAME.out 2> $OUT/$NAME.err
rc=$?; echo "$NAME rc=$rc"
f=$(find /tmp/pmc_$NAME -name "*counter_collection.csv" 2>/dev/null | head -1)
[ -n "$f" ] && python3 tools/pmc_traffic.py $f FETCH_SIZE > $OUT/pmc_fetch_size_$NAME.json && cat $OUT/pmc_fetch_size_$NAME.json
[ $rc -lt 124 ] || [ $rc -eq 139 ]
