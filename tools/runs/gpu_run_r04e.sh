cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
export O=gpurun_out/r04e
mkdir -p $O
bash tools/collect_profiles.sh > $O/collect.log 2>&1
# summaries only
mkdir -p $O/sum
cp gpurun_out/kernel_stats.md gpurun_out/pmc_*.md gpurun_out/pmc_traffic.json gpurun_out/pmc_traffic.log gpurun_out/bench_line.json gpurun_out/bench_line.err gpurun_out/prof_trace_line.json $O/sum/ 2>/dev/null
for f in gpurun_out/prof_*.log gpurun_out/prof_trace.err; do tail -5 $f > $O/sum/$(basename $f).tail 2>/dev/null; done
rm -rf gpurun_out/prof_* gpurun_out/kernel_stats.md gpurun_out/pmc_* gpurun_out/bench_line.* 
( timeout 2400 python -m pytest tests -q -m gpu -s --timeout 1200 -p no:cacheprovider --durations=12 2>&1 | grep -v "^\[Gloo\]" > $O/suite_full.log; grep "^\[\|passed\|failed\|FAILED\|ERROR\|s call" $O/suite_full.log > $O/gpu_suite.log; tail -40 $O/suite_full.log >> $O/gpu_suite.log; rm -f $O/suite_full.log )
( python __graft_entry__.py smoke 2>&1 | tail -3 ) > $O/smoke.log
du -sh gpurun_out; ls $O/sum; cat $O/sum/pmc_traffic.log; grep "passed\|failed\|FAILED" $O/gpu_suite.log | tail -5; cat $O/smoke.log; tail -c 400 $O/sum/bench_line.json
