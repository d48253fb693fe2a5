cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05m; mkdir -p $O
( timeout 900 python -m pytest tests/test_reference_mode_gpu.py -x -q -s -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension" | grep "reference precision\|passed\|failed\|Error\|assert" | tail -30 ) > $O/t_all.log
cat $O/t_all.log
