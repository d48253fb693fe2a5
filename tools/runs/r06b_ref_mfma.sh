# round 6, call 2: f32-MFMA attention kernel tests, the reference-precision suite on it (captured decode steps), range-guard test, bench with the reference leg
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06b; mkdir -p $O
timeout 600 python -m pytest tests/test_decoder_hp_gpu.py tests/test_range_guard_gpu.py -x -q -m gpu > $O/tests_hp.log 2>&1; echo "rc=$?" >> $O/tests_hp.log
tail -8 $O/tests_hp.log
timeout 1200 python -m pytest tests/test_reference_mode_gpu.py -x -q -m gpu -s > $O/tests_ref.log 2>&1; echo "rc=$?" >> $O/tests_ref.log
grep -E "reference precision|passed|failed|Error|rc=" $O/tests_ref.log | tail -30
( timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-from-images > $O/line.json ) 2> $O/err.log
tail -3 $O/err.log; python -c "
import json; d=json.load(open('$O/line.json')); print(d['value'], 'ref', d.get('reference_precision'), 'bf16', (d.get('operands_bf16') or {}).get('value'), {k:v.get('value') for k,v in (d.get('extra_workloads') or {}).items()})"
