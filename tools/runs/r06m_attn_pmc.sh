# round 6, call 14: SQ counters of the ViT full-attention kernel alone (tools/bench_attn.py), two passes
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r06m; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p1 -o a -- python $REPO/tools/bench_attn.py > $O/p1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM --kernel-trace --output-format csv -d $O/p2 -o b -- python $REPO/tools/bench_attn.py > $O/p2.log 2>&1
cd $REPO
python tools/pmc_sum.py $O/p1 attn_varlen > $O/pmc1.md 2>&1; python tools/pmc_sum.py $O/p2 attn_varlen > $O/pmc2.md 2>&1
cat $O/pmc1.md $O/pmc2.md | grep "80, false, 2" ; tail -n 3 $O/p1.log
rm -rf $O/p1 $O/p2
