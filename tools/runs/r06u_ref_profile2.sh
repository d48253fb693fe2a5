# round 6, call 24: kernel trace of the reference-precision runner after split_rows + fused split SwiGLU
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06w; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $REPO/$O/prof_ref -o trace -- python $REPO/tools/profile_reference.py > $REPO/$O/prof_ref.log 2>&1
cd $REPO
DB=$(find $O/prof_ref -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/reference_kernel_stats.md > $O/rocpd_stats.log 2>&1
rm -rf $O/prof_ref
tail -2 $O/prof_ref.log; head -28 $O/reference_kernel_stats.md | cut -c1-150
