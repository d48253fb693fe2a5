# round 6, call 29: persistent dynamic-queue SwiGLU GEMM (PADT_GEMM_PERSIST=1) against the shipped launch: digests + timing, alternating
cd $GRAFT_REPO_ROOT
O=gpurun_out/r07b; mkdir -p $O
for i in 1 2; do for P in 0 1; do PADT_GEMM_PERSIST=$P timeout 300 python tools/bench_gemm_persist.py 2>/dev/null >> $O/ab.log; done; done
cat $O/ab.log
