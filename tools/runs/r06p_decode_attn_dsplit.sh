# round 6, call 17: decode attention with two blocks per (kv head, sample) (d-tile halves): kernel tests + same-call A/B of the decode kernels at 64 / 8 / 128 rows
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06p; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "decode" > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
for B in 64 8 32 128; do for F in 0 1; do echo "== B=$B PADT_DECODE_ATTN_DSPLIT=$F" >> $O/ab.log; B=$B GRAPH=1 PACK=1 KVP=1 ATTN_ONLY=1 GEMMS_SKIP=1 PADT_DECODE_ATTN_DSPLIT=$F timeout 300 python tools/bench_kernels.py 2>/dev/null | grep -i "attn" >> $O/ab.log; done; done
cat $O/ab.log
