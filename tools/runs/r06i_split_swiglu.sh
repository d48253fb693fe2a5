# round 6, call 10: split SwiGLU as the gate/up GEMM's epilogue in precision="reference" (tile + few-row kernels): kernel tests, reference-mode suites, the reference leg alone
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06i; mkdir -p $O
timeout 600 python -m pytest tests/test_decoder_hp_gpu.py -x -q -m gpu > $O/tests_hp.log 2>&1; echo "rc=$?" >> $O/tests_hp.log; tail -n 3 $O/tests_hp.log
timeout 600 python tools/bench_reference_leg.py > $O/ref_leg.json 2> $O/ref_leg.err; cat $O/ref_leg.json; tail -n 3 $O/ref_leg.err
timeout 900 python -m pytest tests/test_reference_mode_gpu.py -x -q -m gpu > $O/tests_ref.log 2>&1; echo "rc=$?" >> $O/tests_ref.log; tail -n 3 $O/tests_ref.log
