set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}; OUT=$REPO/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 --no-alt --no-cpu-baseline --no-roofline --no-graph"
rm -rf $OUT/prof_attn
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/prof_attn -o attn -- $SEQ > $OUT/prof_attn.log 2>&1
cd $REPO; python tools/pmc_sum.py $OUT/prof_attn attn_varlen > $OUT/pmc_attn2.md 2>&1; cat $OUT/pmc_attn2.md | head -40; tail -3 $OUT/prof_attn.log
find $OUT/prof_attn -type f -size +20M -delete
