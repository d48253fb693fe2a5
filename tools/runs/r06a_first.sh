# round 6, call 1: the e2e / range-guard suites on the new collect path + output_scores, smoke, a light bench line
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06a; mkdir -p $O
timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_range_guard_gpu.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -15 $O/tests.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -3 $O/smoke.log
( timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/line_light.json ) 2> $O/err_light.log
tail -3 $O/err_light.log; python -c "
import json; d=json.load(open('$O/line_light.json')); print(d['value'], d.get('roofline',{}).get('frac'), d.get('roofline_decode'))"
