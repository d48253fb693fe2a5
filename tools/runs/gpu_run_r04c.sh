cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04c
mkdir -p $O
( timeout 600 python -m pytest tests/test_e2e_gpu.py -q -m gpu -s --timeout 300 -p no:cacheprovider -k "vit_stream" 2>&1 | tail -15 ) > $O/t_vit_stream.log
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images --no-roofline"
for v in 0 1 0 1; do ( timeout 300 python bench.py --steps 24 --warmup 8 --vit-stream $v $LIGHT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rec vit_stream', $v, d['value'], d['ms_per_step'])" ) >> $O/ab.log 2>> $O/ab.err; done
for v in 0 1; do ( timeout 400 python bench.py --task ovd --steps 32 --warmup 0 --vit-stream $v $LIGHT | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ovd vit_stream', $v, d['value'], d['ms_per_step'])" ) >> $O/ab.log 2>> $O/ab.err; done
tail -3 $O/t_vit_stream.log; cat $O/ab.log; tail -3 $O/ab.err
