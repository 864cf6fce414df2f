#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 300 python tools/ablate_spconv.py --config car --ablate > $O/c2_ablate_car.log 2>&1; echo "ablate car rc=$?"
timeout 300 python tools/ablate_spconv.py --config multi > $O/c2_ablate_multi.log 2>&1; echo "ablate multi rc=$?"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c2 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --eager --inflight 1 > $O/c2_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
find /tmp/prof_c2 -type f | head -20 > $O/c2_prof_files.txt
DB=$(find /tmp/prof_c2 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB > $O/c2_kernel_stats.txt 2>&1; fi
for f in $(find /tmp/prof_c2 -name "*kernel_stats.csv" | head -1); do cp $f $O/c2_kernel_stats.csv; done
echo done
