#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel trace of the default bench + the two PMC passes over one sequential step.
# Outputs under gpurun_out/prof_*; summaries are produced with tools/rocpd_stats.py and tools/pmc_sum.py.
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 --no-alt --no-cpu-baseline --no-roofline --no-graph"
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof_trace -o trace -- python $REPO/bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-roofline > $OUT/prof_trace.log 2>&1
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o fetch -- $SEQ > $OUT/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o write -- $SEQ > $OUT/prof_write.log 2>&1
cd $REPO
DB=$(find $OUT/prof_trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
python tools/pmc_sum.py $OUT/prof_fetch gemm_tile > $OUT/pmc_fetch.md 2>&1
python tools/pmc_sum.py $OUT/prof_write gemm_tile > $OUT/pmc_write.md 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -2 $OUT/prof_trace.log; cat $OUT/bench_line.json; head -12 $OUT/kernel_stats.md; cat $OUT/pmc_fetch.md $OUT/pmc_write.md
# keep the merged-back payload small
find $OUT/prof_trace $OUT/prof_fetch $OUT/prof_write -type f -size +20M -delete
