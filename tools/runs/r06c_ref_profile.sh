# round 6, call 3: attention tests after the tolerance fix, kernel trace of the reference-precision runner, the driver's command with every leg
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06c; mkdir -p $O
timeout 600 python -m pytest tests/test_decoder_hp_gpu.py tests/test_range_guard_gpu.py -q -m gpu > $O/tests_hp.log 2>&1; echo "rc=$?" >> $O/tests_hp.log
tail -4 $O/tests_hp.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/$O/prof_ref -o trace -- python $REPO/tools/profile_reference.py > $REPO/$O/prof_ref.log 2>&1
cd $REPO
DB=$(find $O/prof_ref -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/reference_kernel_stats.md > $O/rocpd_stats.log 2>&1
rm -rf $O/prof_ref
tail -2 $O/prof_ref.log; head -24 $O/reference_kernel_stats.md | cut -c1-150
( timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/line_driver_style.json ) 2> $O/err_driver.log
tail -2 $O/err_driver.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06c/line_driver_style.json'))
print(d['value'], 'ref', (d.get('reference_precision') or {}).get('value'), 'parity', json.dumps(d.get('parity_vs_oracle'))[:1500])
PY
