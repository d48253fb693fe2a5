# round 6, call 11: caller logits_processor / stopping_criteria on the hooked decode loop, bounded RLE scratch: both kernel test files + the e2e file
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06j; mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu > $O/tests_e2e.log 2>&1; echo "rc=$?" >> $O/tests_e2e.log; tail -n 3 $O/tests_e2e.log
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_decoder_hp_gpu.py -x -q -m gpu > $O/tests_k.log 2>&1; echo "rc=$?" >> $O/tests_k.log; tail -n 3 $O/tests_k.log
