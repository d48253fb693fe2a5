# round 6, call 13: same-call A/B of the prefill attention kernels: old library (builtin ds_read_tr, compiler-inserted vmcnt(0)) vs new (asm reads), twice each
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06l; mkdir -p $O
for i in 1 2; do
  echo "== old" >> $O/ab.log; PADT_HIP_LIB=$GRAFT_REPO_ROOT/tools/ab/libpadt_old_attn.so timeout 300 python tools/bench_attn_all.py 2>/dev/null >> $O/ab.log
  echo "== new" >> $O/ab.log; timeout 300 python tools/bench_attn_all.py 2>/dev/null >> $O/ab.log
done
cat $O/ab.log
