# round 5, call a: baseline on this round's box + how the decode-group size interacts with the driver's 20-step window
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05a; mkdir -p $O
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-roofline"
for m in 8 10 5 4 16; do
  ( timeout 150 python bench.py --steps 20 --warmup 5 --merge $m $LIGHT > $O/line_s20_m$m.json ) 2> $O/err_s20_m$m.log
done
( timeout 200 python bench.py --steps 64 --warmup 2 --merge 8 $LIGHT > $O/line_s64_m8.json ) 2> $O/err_s64_m8.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r05a/line_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], d['value'], d['ms_per_step'])
    except Exception as e: print(f, 'ERR', e)
PY
