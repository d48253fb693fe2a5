cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04j; mkdir -p $O
( timeout 2400 python -m pytest tests -x -q -m gpu -s --timeout 1200 -p no:cacheprovider --durations=10 2>&1 | grep -v "^\[Gloo\]" > $O/suite_full.log; grep "^\[\|passed\|failed\|FAILED\|ERROR\|s call" $O/suite_full.log > $O/gpu_suite.log; tail -30 $O/suite_full.log >> $O/gpu_suite.log; rm -f $O/suite_full.log )
grep "passed\|failed\|FAILED\|s call" $O/gpu_suite.log | head -16
