# round 6, call 23: reference-precision decode projections over packed operands (padt_gemm_split_rows layout 3): kernel tests, the reference leg, the reference-mode suites
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06v; mkdir -p $O
timeout 600 python -m pytest tests/test_decoder_hp_gpu.py -x -q -m gpu > $O/tests_hp.log 2>&1; echo "rc=$?" >> $O/tests_hp.log; tail -n 3 $O/tests_hp.log
timeout 600 python tools/bench_reference_leg.py > $O/ref_leg.json 2> $O/ref_leg.err; cat $O/ref_leg.json; tail -n 3 $O/ref_leg.err
timeout 900 python -m pytest tests/test_reference_mode_gpu.py -x -q -m gpu > $O/tests_ref.log 2>&1; echo "rc=$?" >> $O/tests_ref.log; tail -n 3 $O/tests_ref.log
