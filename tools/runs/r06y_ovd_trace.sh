# round 6, call 26: kernel trace of the OVD workload (configs[3] per-GPU shape, decode groups of 16 batches = 128-row steps)
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
REPO=$GRAFT_REPO_ROOT
O=gpurun_out/r06y; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $REPO/$O/prof -o trace -- python $REPO/bench.py --task ovd --steps 32 --warmup 0 --no-alt --no-cpu-baseline --no-extras --no-from-images --no-steady --no-roofline --no-bf16-twin > $REPO/$O/line_ovd.json 2> $REPO/$O/err.log
cd $REPO
DB=$(find $O/prof -name "*.db" | head -1)
python tools/rocpd_stats.py $DB $O/ovd_kernel_stats.md > $O/rocpd_stats.log 2>&1
rm -rf $O/prof
head -30 $O/ovd_kernel_stats.md | cut -c1-150; python -c "
import json; d=json.load(open('$O/line_ovd.json')); print(d['value'], d['ms_per_step'], d['config'])"
