# round 5, last call: the driver's bench command (reference_precision now with fp32 LLM attention), then the whole -m gpu suite + smoke on the final tree
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05n; mkdir -p $O
( timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver_style.json ) 2> $O/err_driver.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05n/line_driver_style.json'))
print('value', d['value'], 'steady', d['steady_state']['value'], 'to_rle', d['to_rle']['value'], 'from_images', d['from_images']['value'], 'bf16', d['operands_bf16']['value'], 'reference', d['reference_precision'].get('value', d['reference_precision']))
PY
( timeout 1500 python -m pytest tests/ -x -q -m gpu --timeout 900 -p no:cacheprovider --durations=8 2>&1 | grep -v "^\[Gloo\]" | tail -30 ) > $O/gpu_suite.log
( timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 ) > $O/smoke.log
tail -14 $O/gpu_suite.log; cat $O/smoke.log
