#!/bin/bash
# K-loop ablations of the persistent GEMM from compile-time variants of the tools-only build (tools/build_stamps.sh <bits> for bits in 1 2 4 3 5 6 7):
# prints the K-loop time of a whole QKV tile (12 K-tiles) per variant
mkdir -p gpurun_out/ablate
for a in ${ABLS:-0 1 2 4 3 5 6 7}; do
  lib=tools/libmgea_hip_stamps.so; [ $a != 0 ] && lib=tools/libmgea_hip_stamps_a$a.so
  [ -f $lib ] || continue
  MGEA_LIB_PATH=$PWD/$lib timeout -k 10 120 python3 tools/gemm_bf16_stamps.py > gpurun_out/ablate/abl_$a.log 2>&1 || exit 1
  echo "ablate $a: $(grep -A2 '== qkv0' gpurun_out/ablate/abl_$a.log | grep 'unit 1' | cut -c1-150)"
done
