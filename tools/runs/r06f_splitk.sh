# round 6, call 6: split-K hand-off with 16-byte sc1 stores / loads — decode projections at 64 / 8 / 128 rows under different splits; packed-KV kernel tests after the tolerance fix
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r06f; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_kernels_f16_gpu.py -x -q -m gpu -k "decode or qkv_post or rows_do_not or packed or skinny or splitk or split_k" > $O/tests_k.log 2>&1; echo "rc=$?" >> $O/tests_k.log
tail -4 $O/tests_k.log
for B in 64 8 128; do
  for cfg in "SPLIT_DOWN=2 SPLIT_O=1" "SPLIT_DOWN=1 SPLIT_O=1" "SPLIT_DOWN=4 SPLIT_O=2" "SPLIT_DOWN=3 SPLIT_O=2"; do
    echo "== B=$B $cfg" >> $O/splitk.log
    env $cfg B=$B GRAPH=1 PACK=1 GEMMS_ONLY=1 timeout 200 python tools/bench_kernels.py 2>/dev/null | grep -E "^o |^down" >> $O/splitk.log
  done
done
cat $O/splitk.log
