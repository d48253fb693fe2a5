cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04b
mkdir -p $O
( timeout 600 python -m pytest tests/test_kernels_f16_gpu.py -q -m gpu -s --timeout 300 -p no:cacheprovider -k "decode_attn_rope or quant_rows" 2>&1 | tail -30 ) > $O/t_kernels.log
( timeout 1500 python -m pytest tests/test_e2e_gpu.py -q -m gpu -s --timeout 900 -p no:cacheprovider --durations=6 2>&1 | grep -v "^\[Gloo\]" > $O/t_e2e_full.log; grep "^\[\|passed\|failed\|FAILED\|s call" $O/t_e2e_full.log > $O/t_e2e.log; tail -80 $O/t_e2e_full.log >> $O/t_e2e.log; rm $O/t_e2e_full.log )
( timeout 2400 python -m pytest tests/test_real_shape_gpu.py -q -m gpu -s --timeout 1200 -p no:cacheprovider -k "ovd_geometry or 7b_full or vit_block" --durations=8 2>&1 | tail -120 ) > $O/t_real.log
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-roofline"
for m in 4 5 8 10; do ( timeout 300 python bench.py --steps 20 --warmup 5 --merge $m $LIGHT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('merge', $m, d['value'], d['ms_per_step'])" ) >> $O/merge_ab.log 2>> $O/merge_ab.err; done
( timeout 300 python bench.py --steps 20 --warmup 5 --operands bf16 $LIGHT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 operands', d['value'], d['ms_per_step'])" ) >> $O/merge_ab.log 2>> $O/merge_ab.err
( timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/bench_legs.json ) 2> $O/bench_legs.err
tail -4 $O/t_kernels.log; grep "passed\|failed\|FAILED" $O/t_e2e.log; grep "^\[\|passed\|failed\|FAILED\|s call" $O/t_real.log; cat $O/merge_ab.log; python -c "
import json; d=json.load(open('$O/bench_legs.json')); print(d['value'], {k: d[k]['value'] for k in ('from_images','to_rle','unmerged_decode') if k in d})"
