#!/bin/bash
# round 4, GPU call J: geometries of the split-operand Winograd GEMM -- parity, timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4j; mkdir -p $O
timeout 300 python tools/run_wino4.py --reps 30 --cfgs ${CFGS:-1,0,14,15,16,0,15} > $O/run_wino4.txt 2>&1; echo "run_wino4 rc $?"; grep -v "^/opt" $O/run_wino4.txt | tail -22 | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_wino4.py -q -x -s -k "geometries" > $O/pytest.txt 2>&1; echo "pytest rc $?"; grep "passed\|failed\|Error\|geometry 1[56]" $O/pytest.txt | tail -14
