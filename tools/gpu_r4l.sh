#!/bin/bash
# round 4, GPU call L: bisect of the fp32 K21 training golden -- 1ff0a0c plus ONE of: the split GEMM (w4), the input-layer kernel (w5), the rulebook (w6)
O=$GRAFT_REPO_ROOT/gpurun_out/r4l; mkdir -p $O
for w in w4 w5 w6; do
  cd "$GRAFT_REPO_ROOT/.bisect/$w" || exit 1
  timeout 900 python -m pytest tests/test_gpu_train.py -q -x -s -k "k21_vs_oracle and fp32" > $O/k21_$w.txt 2>&1; echo "k21 $w rc $?"
  grep -o "worst rel L2 [0-9.e-]* over [0-9]* tensors, taken together [0-9.e-]* | all [0-9]* parameters: worst norm error [0-9.e-]*, worst projection error [0-9.e-]*" $O/k21_$w.txt
done
