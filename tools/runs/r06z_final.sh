# round 6, call 27 (final tree): the whole GPU suite + smoke, then the evidence passes: PMC re-stamp (csrc changed), kernel trace of the light bench, the driver's command with every leg
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=8 > $O/gpu_suite.log 2>&1; echo "rc=$?" >> $O/gpu_suite.log
tail -14 $O/gpu_suite.log
timeout 300 python __graft_entry__.py smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady"
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 $LIGHT --no-roofline --no-graph"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_fetch -o fetch -- $SEQ > $REPO/$O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_write -o write -- $SEQ > $REPO/$O/prof_write.log 2>&1
cd $REPO
python tools/pmc_sum.py $O/prof_fetch gemm_tile > $O/pmc_fetch.md 2>&1
python tools/pmc_sum.py $O/prof_write gemm_tile > $O/pmc_write.md 2>&1
python tools/pmc_traffic_json.py $O/prof_fetch $O/prof_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/r06_pmc_traffic.json
rm -rf $O/prof_fetch $O/prof_write
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/$O/prof_trace -o trace -- python $REPO/bench.py --steps 20 --warmup 5 $LIGHT > $REPO/$O/line_under_rocprof.json 2> $REPO/$O/prof_trace.err
cd $REPO
DB=$(find $O/prof_trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > $O/rocpd_stats.log 2>&1
rm -rf $O/prof_trace
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver_style.json ) 2> $O/err_driver.log
tail -3 $O/pmc_traffic.log; head -14 $O/kernel_stats.md | cut -c1-150
python - <<'PY'
import json
for n in ('line_under_rocprof','line_driver_style'):
    try:
        d=json.load(open('gpurun_out/r06z/%s.json'%n)); r=d.get('roofline',{})
        print(n, d['value'], 'steady', (d.get('steady_state') or {}).get('value'), 'frac', r.get('frac'), r.get('frac_replay'), 'traffic', r.get('traffic'), r.get('traffic_note'),
              'decode', (d.get('roofline_decode') or {}).get('frac_alone'), 'ref', (d.get('reference_precision') or {}).get('value'),
              {k:v.get('value') for k,v in (d.get('extra_workloads') or {}).items()})
    except Exception as e: print(n,'ERR',e)
PY
