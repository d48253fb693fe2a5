# round 4, last call: after the fp16 rope rounding fix (csrc changed → the PMC stamp must be re-measured).  Order = value: the stamp first,
# then the tests that run decode steps on fp16 operands, then one light bench line quoting the new stamp.
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images"
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 $LIGHT --no-roofline --no-graph"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_fetch -o fetch -- $SEQ > $REPO/$O/prof_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_write -o write -- $SEQ > $REPO/$O/prof_write.log 2>&1
cd $REPO
python tools/pmc_sum.py $O/prof_fetch gemm_tile > $O/pmc_fetch.md 2>&1
python tools/pmc_sum.py $O/prof_write gemm_tile > $O/pmc_write.md 2>&1
python tools/pmc_traffic_json.py $O/prof_fetch $O/prof_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/prof_fetch $O/prof_write
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
date +%s > $O/t0
( timeout 300 python -m pytest tests/test_e2e_gpu.py tests/test_decoder_hp_gpu.py -x -q -m gpu --timeout 280 -p no:cacheprovider 2>&1 | grep -v "^\[Gloo\]" | tail -6 ) > $O/t_e2e.log
( timeout 330 python -m pytest tests/test_real_shape_gpu.py -x -q -s -m gpu --timeout 300 -p no:cacheprovider -k "full_depth_3b or batch8" 2>&1 | grep -v "^\[Gloo\]" | tail -40 ) > $O/t_real.log
( timeout 120 python bench.py --steps 20 --warmup 5 $LIGHT > $O/bench_line_20_light.json ) 2> $O/bench_20.err
cat $O/pmc_traffic.log; tail -3 $O/t_e2e.log; tail -4 $O/t_real.log; python -c "
import json; d=json.load(open('$O/bench_line_20_light.json')); r=d['roofline']; print(d['value'], r['frac'], r.get('frac_replay'), r['traffic'], r['traffic_note'])"
