# round 5, call f: reference precision on what the bench times (batch 8, merged runner), all 8 samples against the oracle
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
O=gpurun_out/r05g; mkdir -p $O
( timeout 560 python -m pytest tests/test_reference_mode_gpu.py -x -q -s -m gpu --timeout 540 -p no:cacheprovider -k ovd_geometry 2>&1 | grep -v "^  File\|^Extension" | tail -40 ) > $O/t_ovd.log
cat $O/t_ovd.log
