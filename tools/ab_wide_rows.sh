#!/bin/bash
# A/B of the wide-row-block gate/up launch at 64 decode rows (PADT_SKINNY_WIDE, gemm.hip) on tools/bench_kernels.py (graph replays,
# rotating weights): 3B bf16, 7B bf16, 7B fp8 weight images
run() { echo "== $*"; env "$@" B=64 GRAPH=1 PACK=1 GEMMS_ONLY=1 SPLIT_DOWN=2 python tools/bench_kernels.py 2>&1 | grep -E "^(qkv|o |gu|down)"; }
for g in "GEOM=3b" "GEOM=7b" "GEOM=7b FP8=1"; do
  for w in 0 1; do run PADT_SKINNY_WIDE=$w $g; done
done
