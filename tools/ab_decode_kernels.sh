#!/bin/bash
# A/B of two builds of the library on the decode-step kernels (tools/bench_kernels.py, graph replays, rotating weights):
#   tools/ab_decode_kernels.sh <old.so> [rows ...]
old=$1; shift
rows=${@:-64 8}
for B in $rows; do
  for lib in $old padt_amd/libpadt_hip.so; do
    echo "== rows=$B lib=$lib"
    PADT_HIP_LIB=$PWD/$lib B=$B GRAPH=1 PACK=1 SPLIT_DOWN=2 python tools/bench_kernels.py 2>&1 | grep -v amdgpu.ids
  done
done
