# round 6, call 25: caller hooks on precision="reference" (small config)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06x; mkdir -p $O
timeout 600 python -m pytest tests/test_reference_mode_gpu.py -x -q -m gpu -k small_config > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log
