# round 6, call 15: attention tests + same-call A/B: old library vs (asm V^T reads + running DMA pointers)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06n; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > $O/tests_attn.log 2>&1; echo "rc=$?" >> $O/tests_attn.log; tail -n 3 $O/tests_attn.log
for i in 1 2; do
  echo "== old" >> $O/ab.log; PADT_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libpadt_old_attn.so timeout 300 python tools/bench_attn_all.py 2>/dev/null >> $O/ab.log
  echo "== new" >> $O/ab.log; timeout 300 python tools/bench_attn_all.py 2>/dev/null >> $O/ab.log
done
cat $O/ab.log
