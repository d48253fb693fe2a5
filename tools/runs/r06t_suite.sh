# round 6, call 21: the split-precision test file first (a test-side assertion stopped call 20), then the whole GPU suite + smoke
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06t; mkdir -p $O
timeout 300 python -m pytest tests/test_decoder_hp_gpu.py -x -q -m gpu > $O/hp.log 2>&1; echo "rc=$?" >> $O/hp.log; tail -n 3 $O/hp.log
if grep -q "rc=0" $O/hp.log; then
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
tail -14 $O/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
fi
