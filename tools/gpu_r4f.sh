#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
T='tests/test_gpu_train.py::test_training_step_vs_oracle[configs/multi_cfg.py-names1-HALF1-fp32]'
( cd _bisect/99b8c15 && python -m pytest "$T" -q -s 2>&1 | grep "vs oracle (\|passed\|failed" | cut -c1-700 )
