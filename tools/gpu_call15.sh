#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino4.py -x -q -m gpu -s > $O/c15_tests_wino4.log 2>&1; echo "wino4 tests rc=$?"
timeout 1800 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -s > $O/c15_tests_pipeline.log 2>&1; echo "pipeline tests rc=$?"
grep "wino4 (" $O/c15_tests_wino4.log; grep "max abs errors\|AssertionError: \|passed\|failed" $O/c15_tests_pipeline.log
