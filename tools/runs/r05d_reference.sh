# round 5, call d: precision="reference" — small-config e2e first (fast), then the full-depth 3B test
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05d; mkdir -p $O
( timeout 300 python -m pytest tests/test_reference_mode_gpu.py -x -q -s -m gpu --timeout 280 -p no:cacheprovider -k small 2>&1 | grep -v "^  File\|^Extension" | tail -40 ) > $O/t_small.log
cat $O/t_small.log
if grep -q "1 passed" $O/t_small.log; then
( timeout 500 python -m pytest tests/test_reference_mode_gpu.py -x -q -s -m gpu --timeout 480 -p no:cacheprovider -k full_depth 2>&1 | grep -v "^  File\|^Extension" | tail -40 ) > $O/t_full.log
cat $O/t_full.log
fi
