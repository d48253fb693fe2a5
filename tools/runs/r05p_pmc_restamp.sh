# round 5, after the last csrc edit (padt_attn_f32 arguments, padt_scatter_rows_f32): the PMC traffic stamp carries a hash of csrc/ — measure it again, then one light line quoting it
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r05p; mkdir -p $O
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
( timeout 200 python bench.py --steps 20 --warmup 5 $LIGHT > $O/line_light.json ) 2> $O/err.log
cat $O/pmc_traffic.log | tail -2
python -c "
import json; d=json.load(open('$O/line_light.json')); r=d['roofline']; print(d['value'], r['frac'], r['frac_replay'], r['traffic'], r['traffic_note'])"
