# round 5, call l: reference precision with fp32 LLM attention — kernel tests, decoder tests (shared kernel), then the reference-mode tests small → full
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05l; mkdir -p $O
( timeout 300 python -m pytest tests/test_decoder_hp_gpu.py -x -q -m gpu --timeout 280 -p no:cacheprovider 2>&1 | tail -15 ) > $O/t_dec.log
tail -15 $O/t_dec.log
( timeout 300 python -m pytest tests/test_reference_mode_gpu.py -x -q -s -m gpu --timeout 280 -p no:cacheprovider -k "small" 2>&1 | grep -v "^  File\|^Extension" | tail -40 ) > $O/t_small.log
tail -30 $O/t_small.log
if grep -q "1 passed" $O/t_small.log; then
( timeout 900 python -m pytest tests/test_reference_mode_gpu.py -x -q -s -m gpu --timeout 600 -p no:cacheprovider -k "full_depth" 2>&1 | grep -v "^  File\|^Extension" | grep "reference precision\|passed\|failed\|Error\|assert" | tail -20 ) > $O/t_full.log
cat $O/t_full.log
fi
