#!/bin/bash
# Run ON THE GPU BOX (gpurun): kernel trace of the default bench, PMC passes over one sequential step (HBM-side bytes; MFMA / LDS
# utilisation of the tile GEMM), and the full bench line.  Outputs under gpurun_out/prof_*; summaries by tools/rocpd_stats.py / tools/pmc_sum.py.
# PMC passes are separate runs with --kernel-trace only (never combined with sys / runtime traces).
set -u
REPO=${GRAFT_REPO_ROOT:-$PWD}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIGHT="--no-alt --no-cpu-baseline --no-extras --no-from-images"
SEQ="python $REPO/bench.py --steps 1 --warmup 0 --depth 1 --merge 1 $LIGHT --no-roofline --no-graph"
# (1) kernel trace of the pipelined bench; the bench line printed INSIDE this run carries roofline.frac measured under the profiler
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof_trace -o trace -- python $REPO/bench.py --steps 20 --warmup 5 $LIGHT > $OUT/prof_trace_line.json 2> $OUT/prof_trace.err
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_fetch -o fetch -- $SEQ > $OUT/prof_fetch.log 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/prof_write -o write -- $SEQ > $OUT/prof_write.log 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/prof_mfma -o mfma -- $SEQ > $OUT/prof_mfma.log 2>&1
timeout 400 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 --kernel-trace --output-format csv -d $OUT/prof_lds -o lds -- $SEQ > $OUT/prof_lds.log 2>&1
cd $REPO
DB=$(find $OUT/prof_trace -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $OUT/kernel_stats.md > /dev/null 2>&1
python tools/pmc_sum.py $OUT/prof_fetch gemm_tile > $OUT/pmc_fetch.md 2>&1
python tools/pmc_sum.py $OUT/prof_write gemm_tile > $OUT/pmc_write.md 2>&1
python tools/pmc_sum.py $OUT/prof_mfma gemm_tile > $OUT/pmc_mfma.md 2>&1
python tools/pmc_sum.py $OUT/prof_lds gemm_tile > $OUT/pmc_lds.md 2>&1
python tools/pmc_traffic_json.py $OUT/prof_fetch $OUT/prof_write $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
# (2) decode-step kernels at the row counts the bench runs (64 rows: REC decode groups of 8 batches; 128 rows: groups of 16), eager launches:
#     fabric-side bytes, L2 hits / misses and L1 -> L2 read requests of gemm_skinny_kernel / decode_attn_rope_kernel / vrt_head_kernel
cd /tmp
for rows in 64 128; do
  m=$((rows / 8))
  DEC="python $REPO/bench.py --steps $m --warmup 0 --depth 1 --merge $m --no-graph $LIGHT --no-roofline"
  timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/prof_dec${rows}_fetch -o f -- $DEC > $OUT/prof_dec${rows}_fetch.log 2>&1
  timeout 500 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/prof_dec${rows}_l2 -o l -- $DEC > $OUT/prof_dec${rows}_l2.log 2>&1
  timeout 500 rocprofv3 --pmc TCP_TCC_READ_REQ_sum --kernel-trace --output-format csv -d $OUT/prof_dec${rows}_tcp -o t -- $DEC > $OUT/prof_dec${rows}_tcp.log 2>&1
done
cd $REPO
for rows in 64 128; do
  for k in fetch l2 tcp; do python tools/pmc_sum.py $OUT/prof_dec${rows}_$k gemm_skinny decode_attn decode_combine vrt_head greedy > $OUT/pmc_dec${rows}_$k.md 2>&1; done
done
timeout 1200 python bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
tail -c 600 $OUT/prof_trace_line.json; head -14 $OUT/kernel_stats.md; cat $OUT/pmc_mfma.md $OUT/pmc_lds.md | head -40
# keep the merged-back payload small
# keep the merged-back payload small (gpurun copies back at most 64 MiB, and refuses when the LOCAL gpurun_out/ already exceeds it)
find $OUT/prof_* -type f -size +4M -delete
du -sh $OUT
