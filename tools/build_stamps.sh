#!/bin/bash
# Tools-only build of the native library with the in-kernel time stamps of the persistent GEMM compiled in (-DMGEA_PH_STAMPS):
# tools/libmgea_hip_stamps.so, loaded by tools/gemm_bf16_stamps.py through MGEA_LIB_PATH.  The product library never has them.
# build_stamps.sh [ablate bits]: with bits, the whole-tile K loop is built without LDS-DMA (1) / fragment reads (2) / MFMAs (4) ->
# tools/libmgea_hip_stamps_a<bits>.so (timings only, wrong results)
set -e
ABL=${1:-0}
SUF=""; [ "$ABL" != "0" ] && SUF="_a$ABL"
cd "$(dirname "$0")/../music-generation-emotion-adaptive_amd/csrc"
mkdir -p build_stamps
for f in capi gemm_f32 gemm_skinny gemv_small bf16 rowops attn_paged attn_dense attn_cls sampler decoder bert; do
  if [ $f = bf16 ] || [ ! -f build/$f.o ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DMGEA_PH_STAMPS -DMGEA_PH_ABLATE=$ABL -c $f.hip -o build_stamps/$f.o
  else
    cp build/$f.o build_stamps/$f.o
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libmgea_hip_stamps$SUF.so build_stamps/*.o
ls -la ../../tools/libmgea_hip_stamps$SUF.so
