cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
echo "== default dispatch" > $O/resid.log; python tools/bench_resid32.py >> $O/resid.log 2>&1
echo "== PADT_GEMM256=0 (128^2 kernel, 2 blocks per CU)" >> $O/resid.log; PADT_GEMM256=0 python tools/bench_resid32.py >> $O/resid.log 2>&1
for mf in 2 3 4; do echo "== PADT_GEMM_MF=$mf" >> $O/resid.log; PADT_GEMM_MF=$mf python tools/bench_resid32.py >> $O/resid.log 2>&1; done
cat $O/resid.log
