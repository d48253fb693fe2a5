# round 5, call h: does the number of HSA hardware queues (GPU_MAX_HW_QUEUES, default 4) matter for the runner's streams (null, prefill, 2 decode lanes, post)?
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05h; mkdir -p $O
L="--no-alt --no-cpu-baseline --no-extras --no-roofline --no-steady"
for q in default 8 2; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  for rep in 1 2; do
    ( timeout 200 python bench.py --steps 20 --warmup 5 $L > $O/line_q${q}_$rep.json ) 2> $O/err_q${q}_$rep.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05h/line_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], 'from_images', d['from_images']['value'], 'to_rle', d['to_rle']['value'])
    except Exception as e: print(f,'ERR',e)
PY
