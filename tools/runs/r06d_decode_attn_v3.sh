# round 6, call 4: the single-launch decode attention prototype (v3) against the library kernel; the range-guard test after its fix
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06d; mkdir -p $O
timeout 600 python tools/bench_decode_attn_v2.py > $O/decode_attn_v3.log 2>&1; echo "rc=$?" >> $O/decode_attn_v3.log
cat $O/decode_attn_v3.log | cut -c1-400
timeout 300 python -m pytest tests/test_range_guard_gpu.py -q -m gpu > $O/tests_rg.log 2>&1; tail -3 $O/tests_rg.log
