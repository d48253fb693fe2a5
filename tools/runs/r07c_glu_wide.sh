# round 6, call 30: SwiGLU tiles through LDS as whole 256-byte rows (PADT_GEMM_GLU_WIDE=1, new default) against the 8-byte fragment stores: digests + timing, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r07c; mkdir -p $O
for i in 1 2; do for P in 0 1; do echo "== PADT_GEMM_GLU_WIDE=$P" >> $O/ab.log; PADT_GEMM_GLU_WIDE=$P timeout 300 python tools/bench_gemm_persist.py 2>/dev/null | grep -v PERSIST >> $O/ab.log; done; done
cat $O/ab.log
