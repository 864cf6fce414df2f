#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_fused.py -q -m gpu -x > $O/l_tests1.log 2>&1; echo "tests1 rc=$?"; tail -3 $O/l_tests1.log
timeout 2400 python -m pytest tests/test_gpu_train.py -q -m gpu -x -s > $O/l_tests2.log 2>&1; echo "tests2 rc=$?"; grep -a "bf16 step, every\|trajector\|passed\|failed\|Error\|K21 x 2 training step (bf16)" $O/l_tests2.log | cut -c1-700 | tail -8
for rep in 1 2; do
timeout 300 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train bf16', d['value'], d['ms_per_step'])"
done
