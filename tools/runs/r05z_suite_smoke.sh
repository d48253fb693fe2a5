# round 5, evidence call 2: the whole -m gpu suite on the end-of-round tree, then smoke()
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05z; mkdir -p $O
( timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 900 -p no:cacheprovider --durations=12 2>&1 | grep -v "^\[Gloo\]" | tail -40 ) > $O/gpu_suite.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 ) > $O/smoke.log
tail -25 $O/gpu_suite.log; cat $O/smoke.log
