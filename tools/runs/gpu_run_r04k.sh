cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
( timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py tests/test_decoder_hp_gpu.py tests/test_e2e_gpu.py -x -q -m gpu --timeout 600 -p no:cacheprovider 2>&1 | grep -v "^\[Gloo\]" | tail -8 ) > $O/t.log
cd /tmp && export TMPDIR=/tmp
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images"
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 $LIGHT --no-roofline --no-graph"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_fetch -o fetch -- $SEQ > $REPO/$O/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $REPO/$O/prof_write -o write -- $SEQ > $REPO/$O/prof_write.log 2>&1
cd $REPO
python tools/pmc_sum.py $O/prof_fetch gemm_tile > $O/pmc_fetch.md 2>&1
python tools/pmc_sum.py $O/prof_write gemm_tile > $O/pmc_write.md 2>&1
python tools/pmc_traffic_json.py $O/prof_fetch $O/prof_write $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1
rm -rf $O/prof_fetch $O/prof_write
cp $O/pmc_traffic.json profiles/r04_pmc_traffic.json
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_line_20.json ) 2> $O/bench_20.err
cat $O/t.log; cat $O/pmc_traffic.log; python -c "
import json; d=json.load(open('$O/bench_line_20.json')); r=d['roofline']; print(d['value'], r['frac'], r['frac_replay'], r['traffic'], r['traffic_note'], d['to_rle']['value'], d['extra_workloads']['ovd_3b']['value'], d['extra_workloads']['ric_7b_fp8']['value'])"
