cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04h; mkdir -p $O
( PADT_OPERANDS=bf16 timeout 900 python -m pytest tests/test_real_shape_gpu.py -q -m gpu -s --timeout 800 -p no:cacheprovider -k "full_depth_3b" 2>&1 | grep "^\[\|passed\|failed\|FAILED\|^E " ) > $O/full_depth_bf16.log
cat $O/full_depth_bf16.log | tail -8
