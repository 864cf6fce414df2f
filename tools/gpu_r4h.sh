#!/bin/bash
# the rest of the default -m gpu suite after the point where the previous full run stopped (-x), timed
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4h; mkdir -p $O
start=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_wino4.py -x -q -m gpu --durations=8 > $O/pytest_gpu_rest.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - start )) s"
tail -25 $O/pytest_gpu_rest.txt | cut -c1-400
