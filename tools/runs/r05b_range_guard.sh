# round 5, call b: the fp16 range guard (rope_fin + padt_check_finite + operands="auto") — new tests, the kernel / e2e files the csrc edit touches, a light line
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05b; mkdir -p $O
( timeout 300 python -m pytest tests/test_range_guard_gpu.py -x -q -s -m gpu --timeout 280 -p no:cacheprovider 2>&1 | tail -30 ) > $O/t_guard.log
( timeout 300 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -m gpu --timeout 280 -p no:cacheprovider 2>&1 | tail -8 ) > $O/t_kernels.log
( timeout 400 python -m pytest tests/test_e2e_gpu.py tests/test_decoder_hp_gpu.py -x -q -m gpu --timeout 380 -p no:cacheprovider 2>&1 | grep -v "^\[Gloo\]" | tail -8 ) > $O/t_e2e.log
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-roofline"
( timeout 150 python bench.py --steps 20 --warmup 5 $LIGHT > $O/line_s20_auto.json ) 2> $O/err_auto.log
( timeout 150 python bench.py --steps 20 --warmup 5 --operands bf16 $LIGHT > $O/line_s20_bf16.json ) 2> $O/err_bf16.log
cat $O/t_guard.log; tail -3 $O/t_kernels.log; tail -3 $O/t_e2e.log
python -c "
import json
for n in ('auto','bf16'):
    try:
        d=json.load(open('$O/line_s20_%s.json'%n)); print(n, d['value'], d['dtype'])
    except Exception as e: print(n,'ERR',e)
"
