#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
SASSD_FULL_TESTS=1 timeout 1200 python -m pytest tests/test_gpu_train.py::test_training_step_vs_oracle -q -m gpu -s 2>&1 | grep -a "whole-model\|passed\|failed" | cut -c1-250
timeout 1500 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_bf16.py -q -m gpu -x -s 2>&1 | grep -a "passed\|failed\|BN statistics" | cut -c1-200
python tests/analysis/bn_stats_probe.py 2>&1 | grep nchw | head -3
