#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf_x; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_x -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train --no-extra > $O/bench_inflight3_under_rocprof.log 2>&1 ); echo rc=$?
DB=$(find /tmp/pf_x -name "*.db" | head -1); python tools/rocprof_summary.py $DB > $O/bench_inflight3_kernel_stats.txt
