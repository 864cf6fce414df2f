#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino4.py -x -q -m gpu > $O/c10_tests_wino4.log 2>&1; echo "wino4 tests rc=$?"
timeout 300 python tools/run_wino4.py > $O/c10_run_wino4.log 2>&1; echo "run rc=$?"
timeout 300 python tools/run_wino4.py --batch 8 --reps 10 > $O/c10_run_wino4_b8.log 2>&1; echo "run b8 rc=$?"
tail -3 $O/c10_tests_wino4.log; cat $O/c10_run_wino4.log $O/c10_run_wino4_b8.log
