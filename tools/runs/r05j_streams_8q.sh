# round 5, call j: with 8 hardware queues, do the stream variants that lost with 4 (ViT on its own stream, one prefill stream per lane, depth 3) behave differently?
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05j; mkdir -p $O
L="--no-alt --no-cpu-baseline --no-extras --no-roofline --no-steady --no-from-images"
for s in 20 64; do
  i=0
  for v in "" "--vit-stream 1" "--lane-streams" "--depth 3" "--vit-stream 1 --lane-streams"; do
    i=$((i+1))
    ( timeout 200 python bench.py --steps $s --warmup 5 $v $L > $O/line_s${s}_v$i.json ) 2> $O/err_s${s}_v$i.log
    python -c "
import json
try:
    d=json.load(open('$O/line_s${s}_v$i.json')); print('steps $s', '[$v]', d['value'])
except Exception as e: print('steps $s [$v] ERR', e)
"
  done
done
