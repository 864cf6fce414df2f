#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "reference_defaults or nms_normal or facade" > $O/c18_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/c18_tests.log
timeout 1800 python -m pytest tests/test_gpu_train.py tests/test_gpu_optim.py -x -q -m gpu > $O/c18_tests_train.log 2>&1; echo "train tests rc=$?"; tail -3 $O/c18_tests_train.log
timeout 600 python bench.py --mode train --steps 20 --warmup 6 > $O/c18_bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/c18_bench_train.log | cut -c1-200
timeout 600 python tools/time_train_phases.py 10 > $O/c18_phases.log 2>&1; echo "phases rc=$?"; tail -1 $O/c18_phases.log
