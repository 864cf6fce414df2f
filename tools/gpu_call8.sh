#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino4.py -x -q -m gpu -s > $O/c8_tests_wino4.log 2>&1; echo "wino4 tests rc=$?"
timeout 300 python tools/run_wino4.py > $O/c8_run_wino4.log 2>&1; echo "run rc=$?"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c8 -- python $GRAFT_REPO_ROOT/tools/run_wino4.py > $O/c8_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_c8 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB > $O/c8_kernel_stats.txt 2>&1; fi
tail -12 $O/c8_tests_wino4.log; cat $O/c8_run_wino4.log; head -8 $O/c8_kernel_stats.txt
