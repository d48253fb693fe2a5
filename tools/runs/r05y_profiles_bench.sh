# round 5, evidence call 1: PMC re-stamp (csrc changed this round) → kernel trace of the light bench → the driver's command with every leg → the default 64-step line
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r05y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady"
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 $LIGHT --no-roofline --no-graph"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_fetch -o fetch -- $SEQ > $REPO/$O/prof_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_write -o write -- $SEQ > $REPO/$O/prof_write.log 2>&1
cd $REPO
python tools/pmc_sum.py $O/prof_fetch gemm_tile > $O/pmc_fetch.md 2>&1
python tools/pmc_sum.py $O/prof_write gemm_tile > $O/pmc_write.md 2>&1
python tools/pmc_traffic_json.py $O/prof_fetch $O/prof_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json profiles/r05_pmc_traffic.json
rm -rf $O/prof_fetch $O/prof_write
cd /tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/$O/prof_trace -o trace -- python $REPO/bench.py --steps 20 --warmup 5 $LIGHT > $REPO/$O/line_under_rocprof.json 2> $REPO/$O/prof_trace.err
cd $REPO
DB=$(find $O/prof_trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/kernel_stats.md > $O/rocpd_stats.log 2>&1
rm -rf $O/prof_trace
( timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver_style.json ) 2> $O/err_driver.log
( timeout 700 python bench.py --no-extras --no-cpu-baseline > $O/line_default_64.json ) 2> $O/err_default.log
cat $O/pmc_traffic.log | tail -3; head -16 $O/kernel_stats.md
python - <<'PY'
import json
for n in ('line_under_rocprof','line_driver_style','line_default_64'):
    try:
        d=json.load(open('gpurun_out/r05y/%s.json'%n)); r=d.get('roofline',{})
        print(n, d['value'], 'steady', (d.get('steady_state') or {}).get('value'), 'frac', r.get('frac'), r.get('frac_replay'), 'traffic', r.get('traffic'), r.get('traffic_note'),
              'to_rle', (d.get('to_rle') or {}).get('value'), 'from_images', (d.get('from_images') or {}).get('value'), 'guard', d.get('range_guard',{}).get('batches_rerun_on_bf16'))
    except Exception as e: print(n,'ERR',e)
PY
