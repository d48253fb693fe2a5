cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04d
mkdir -p $O
( timeout 2400 python -m pytest tests -q -m gpu -s --timeout 1200 -p no:cacheprovider --durations=12 2>&1 | grep -v "^\[Gloo\]" > $O/suite_full.log; grep "^\[\|passed\|failed\|FAILED\|ERROR\|s call" $O/suite_full.log > $O/gpu_suite.log; tail -60 $O/suite_full.log >> $O/gpu_suite.log; rm -f $O/suite_full.log )
( python __graft_entry__.py smoke 2>&1 | tail -3 ) > $O/smoke.log
bash tools/collect_profiles.sh > $O/collect.log 2>&1
grep "passed\|failed\|FAILED" $O/gpu_suite.log | tail -5; cat $O/smoke.log; tail -30 $O/collect.log
