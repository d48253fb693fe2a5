# round 5, call e: the driver's command as typed, all legs (timing of the whole run + every new key)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05e; mkdir -p $O
date +%s > $O/t0
( timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver_style.json ) 2> $O/err.log
date +%s > $O/t1
echo "bench wall: $(( $(cat $O/t1) - $(cat $O/t0) )) s"
tail -5 $O/err.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05e/line_driver_style.json'))
print('value', d['value'], d['dtype'], 'steady', d.get('steady_state'))
for k in ('operands_bf16','reference_precision','from_images','to_rle','unmerged_decode','single_batch_latency'):
    print(k, json.dumps(d.get(k))[:400])
r=d.get('roofline',{}); print('roofline', {k:r.get(k) for k in ('achieved','frac','frac_replay','traffic','traffic_note')})
print('top', r.get('top_shapes'))
rd=d.get('roofline_decode',{}); print('decode', {k:rd.get(k) for k in ('us_per_step','frac','us_per_step_alone','frac_alone')})
print('cpu', json.dumps(d.get('cpu_baseline'))[:600])
for k,v in (d.get('extra_workloads') or {}).items(): print(k, v if 'error' in v else (v['value'], v['workload'][:80]))
PY
