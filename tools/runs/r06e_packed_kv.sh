# round 6, call 5: packed KV caches + one-launch decode attention in the library: kernel tests, e2e / real-shape LLM tests, light bench A/B (PADT_KV_PACKED=0)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06e; mkdir -p $O
timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -m gpu -k "decode or qkv_post or rows_do_not" -s > $O/tests_k.log 2>&1; echo "rc=$?" >> $O/tests_k.log
grep -E "decode attention, packed|passed|failed|rc=" $O/tests_k.log | tail -12
timeout 1500 python -m pytest tests/test_e2e_gpu.py tests/test_real_shape_gpu.py -x -q -m gpu > $O/tests_e2e.log 2>&1; echo "rc=$?" >> $O/tests_e2e.log
tail -5 $O/tests_e2e.log
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady"
( timeout 600 python bench.py --steps 20 --warmup 5 $LIGHT > $O/line_packed.json ) 2> $O/err_packed.log
( PADT_KV_PACKED=0 timeout 600 python bench.py --steps 20 --warmup 5 $LIGHT > $O/line_rowmajor.json ) 2> $O/err_rowmajor.log
( timeout 600 python bench.py --steps 20 --warmup 5 $LIGHT > $O/line_packed2.json ) 2> $O/err_packed2.log
python - <<'PY'
import json
for n in ('line_packed','line_rowmajor','line_packed2'):
    try:
        d=json.load(open('gpurun_out/r06e/%s.json'%n)); r=d.get('roofline_decode',{})
        print(n, d['value'], 'decode alone us', r.get('us_per_step_alone'), 'frac_alone', r.get('frac_alone'), 'in situ', r.get('us_per_step'))
    except Exception as e: print(n,'ERR',e)
PY
