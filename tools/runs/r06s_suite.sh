# round 6, call 20: the whole GPU suite + smoke on the tree of call 19 (whose suite stopped at a test-side assertion: a peeled tail is summed in the few-row kernel's order)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06s; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
tail -14 $O/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
