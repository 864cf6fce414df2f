#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1800 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "training_step_vs_oracle" > $O/c19_tests_train.log 2>&1; echo "train tests rc=$?"; tail -3 $O/c19_tests_train.log
