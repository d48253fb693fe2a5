# round 6, call 16: ablation builds of the prefill attention kernel (timing only, results wrong): 1 no DMA after tile 0, 2 no exp2, 3 no per-tile block barrier
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06o; mkdir -p $O
for n in 0 1 2 3; do
  echo "== ablation $n" >> $O/abl.log
  if [ $n = 0 ]; then timeout 300 python tools/bench_attn_all.py 2>/dev/null >> $O/abl.log; else PADT_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libpadt_abl$n.so timeout 300 python tools/bench_attn_all.py 2>/dev/null >> $O/abl.log; fi
done
cat $O/abl.log
