# round 6, call 12: prefill attention with the V^T transpose reads as inline asm (no compiler-inserted vmcnt(0) in front of them): attention tests + the three prefill shapes
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06k; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "attn or attention" > $O/tests_attn.log 2>&1; echo "rc=$?" >> $O/tests_attn.log; tail -n 3 $O/tests_attn.log
timeout 300 python tools/bench_attn_all.py > $O/bench_attn_all.log 2>&1; cat $O/bench_attn_all.log | tail -n 12
