# round 6, call 28: what each fused epilogue piece of the ViT qkv GEMM costs (stand-alone)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r07a; mkdir -p $O
timeout 300 python tools/bench_gemm_qkv_rope.py > $O/qkv.log 2>&1; cat $O/qkv.log | tail -8
