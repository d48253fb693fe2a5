cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04i; mkdir -p $O
LIGHT="--no-alt --no-cpu-baseline --no-from-images"
( timeout 600 python bench.py --task ovd --steps 48 --warmup 0 $LIGHT > $O/ovd_line.json ) 2> $O/ovd.err
for wt in bf16 fp8 fp8+act; do ( timeout 900 python bench.py --model 7b --task ric --weights $wt --steps 48 --warmup 0 $LIGHT > $O/ric7b_${wt}_line.json ) 2> $O/ric7b_$wt.err; done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04i/*_line.json")):
    try:
        d=json.load(open(f)); print(f.split("/")[-1], d["value"], d["ms_per_step"], d.get("roofline",{}).get("frac"), d.get("roofline_decode",{}).get("frac_alone"), d.get("roofline_decode",{}).get("us_per_step_alone"))
    except Exception as e: print(f, "ERR", e)
PY
