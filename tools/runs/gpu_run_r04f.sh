cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r04f
mkdir -p $O
( timeout 900 python -m pytest tests/test_e2e_gpu.py tests/test_kernels_gpu.py -q -m gpu -s --timeout 600 -p no:cacheprovider -k "massive_activation or gpu_resize_is_byte_exact" 2>&1 | tail -25 ) > $O/t.log
cat $O/t.log | tail -12
