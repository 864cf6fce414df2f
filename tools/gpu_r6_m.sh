#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf_t; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_t -- python $R/bench.py --mode train --steps 10 --warmup 4 > $O/m_train_rocprof.log 2>&1 ); echo "rc=$?"
DB=$(find /tmp/pf_t -name "*.db" | head -1); python tools/rocprof_summary.py $DB > $O/m_train_kernel_stats.txt 2>&1; head -45 $O/m_train_kernel_stats.txt | cut -c1-170
