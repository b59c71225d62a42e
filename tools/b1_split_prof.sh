#!/bin/bash
# B = 1 greedy generation under rocprofv3 --kernel-trace --stats for each value of the attn_split switch (GPU box; run from the repo root)
set -o pipefail
export TMPDIR=/tmp
OUT=${1:-gpurun_out/b1_split}; mkdir -p $OUT
for v in ${2:-0 2 1}; do
  rm -rf /tmp/b1s_$v
  MGEA_ATTN_SPLIT=$v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b1s_$v -- python3 tools/b1_prof.py > $OUT/b1_split_$v.log 2>&1 || exit 1
  f=$(find /tmp/b1s_$v -name "*kernel_stats.csv" | head -1)
  t=$(find /tmp/b1s_$v -name "*kernel_trace.csv" | head -1)
  [ -n "$t" ] && python3 tools/trace_gaps.py $t > $OUT/b1_split_${v}_gaps.txt
  [ -n "$f" ] && cp $f $OUT/b1_split_${v}_kernel_stats.csv && echo "== attn_split=$v" && grep "us per step" $OUT/b1_split_$v.log && python3 tools/kstats.py $f "" 8
done
