# round 6, call 9: padt_gemm_split_rows (reference-precision decode steps read each weight once): kernel tests, reference-mode suites, the reference leg of the bench
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06h; mkdir -p $O
timeout 600 python -m pytest tests/test_decoder_hp_gpu.py -x -q -m gpu > $O/tests_hp.log 2>&1; echo "rc=$?" >> $O/tests_hp.log; tail -3 $O/tests_hp.log
timeout 900 python -m pytest tests/test_reference_mode_gpu.py -x -q -m gpu > $O/tests_ref.log 2>&1; echo "rc=$?" >> $O/tests_ref.log; tail -3 $O/tests_ref.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady --no-bf16-twin > $O/line_light_ref.json ) 2> $O/err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06h/line_light_ref.json'))
print(d['value'], 'ref', (d.get('reference_precision') or {}).get('value'), (d.get('reference_precision') or {}).get('ms_per_step'))
PY
