#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s > $O/c14_tests_pipeline.log 2>&1; echo "pipeline tests rc=$?"
timeout 1500 python -m pytest tests/test_gpu_train.py -x -q -m gpu -k "training_step_vs_oracle or weight_packs" > $O/c14_tests_train.log 2>&1; echo "train tests rc=$?"
grep "max abs errors" $O/c14_tests_pipeline.log; tail -5 $O/c14_tests_pipeline.log; tail -5 $O/c14_tests_train.log
