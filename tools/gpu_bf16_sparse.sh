#!/bin/bash
# 64-channel sparse convs on the bf16 MFMA in the bf16 training mode: step parity on the bench workload (K21 golden), all three modes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/bf16sp; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_train.py -q -s -k "k21_vs_oracle" > $O/pytest_k21.txt 2>&1; echo "pytest rc $?"
grep -o "K21 x 2 training step ([^)]*) vs oracle: stored-layer gradient cosine [0-9.]*\|worst rel L2 [0-9.e-]* over [0-9]* tensors, taken together [0-9.e-]* | all [0-9]* parameters: worst norm error [0-9.e-]*\|[0-9]* passed.*\|[0-9]* failed.*" $O/pytest_k21.txt
grep "AssertionError\|assert " $O/pytest_k21.txt | head -5 | cut -c1-300
