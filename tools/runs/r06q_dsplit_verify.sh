# round 6, call 18: d-split decode attention in the product path: kernel + e2e tests, light bench line (decode step alone / in situ)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06q; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_e2e_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady > $O/line_light.json ) 2> $O/err.log
( PADT_DECODE_ATTN_DSPLIT=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady > $O/line_light_nosplit.json ) 2>> $O/err.log
python - <<'PY'
import json
for n in ('line_light','line_light_nosplit'):
    d=json.load(open('gpurun_out/r06q/%s.json'%n)); r=d['roofline_decode']
    print(n, d['value'], 'decode alone us', r['us_per_step_alone'], r['frac_alone'], 'in situ', r['us_per_step'])
PY
