# round 5, call c: range guard after the greedy_step NaN-row fix + device RLE (kernel test, e2e post-processing tests, the to_rle leg)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05c; mkdir -p $O
( timeout 300 python -m pytest tests/test_range_guard_gpu.py -x -q -s -m gpu --timeout 280 -p no:cacheprovider 2>&1 | grep -v "^  File\|^Extension" | tail -40 ) > $O/t_guard.log
( timeout 200 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu --timeout 180 -p no:cacheprovider -k "rle or mask_upsample" 2>&1 | tail -15 ) > $O/t_rle.log
( timeout 300 python -m pytest tests/test_e2e_gpu.py -x -q -m gpu --timeout 280 -p no:cacheprovider -k "postprocess or harness or massive" 2>&1 | grep -v "^\[Gloo\]" | tail -15 ) > $O/t_e2e.log
( timeout 200 python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-extras --no-roofline > $O/line_s20.json ) 2> $O/err.log
cat $O/t_guard.log; tail -15 $O/t_rle.log; tail -5 $O/t_e2e.log
python -c "
import json
d=json.load(open('$O/line_s20.json')); print(d['value'], d['dtype'], d.get('from_images',{}).get('value'), d.get('to_rle'))
"
