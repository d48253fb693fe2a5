cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04a
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_f16_gpu.py tests/test_kernels_gpu.py tests/test_decoder_hp_gpu.py -q -m gpu --timeout 300 -p no:cacheprovider 2>&1 | tail -60 ) > $O/t_kernels.log
( timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -m gpu -s --timeout 900 -p no:cacheprovider 2>&1 | grep -v "^\[Gloo\]" | tail -150 ) > $O/t_e2e.log
( timeout 1500 python -m pytest tests/test_real_shape_gpu.py -q -m gpu -s --timeout 900 -p no:cacheprovider -k "not ovd_geometry and not 7b_full" --durations=8 2>&1 | tail -150 ) > $O/t_real.log
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line.json ) 2> $O/bench.err
tail -5 $O/t_kernels.log; tail -5 $O/t_e2e.log; tail -5 $O/t_real.log; tail -c 1500 $O/bench_line.json; tail -5 $O/bench.err
