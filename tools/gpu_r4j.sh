#!/bin/bash
# round 4, GPU call J: fp32 products on the bf16 MFMA (split operands) in the Winograd GEMM -- parity of every geometry, timings
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4j; mkdir -p $O
timeout 300 python tools/run_wino4.py --reps 20 --cfgs ${CFGS:-1,5,12,14,19,20,21,26,27} > $O/run_wino4.txt 2>&1; echo "run_wino4 rc $?"; grep -v "^/opt" $O/run_wino4.txt | tail -22 | cut -c1-230
timeout 600 python -m pytest tests/test_gpu_wino4.py -q -x -s -k "geometries and (19 or 20 or 26 or 27)" > $O/pytest.txt 2>&1; echo "pytest rc $?"; grep "passed\|failed\|Error\|geometry 2\|geometry 19" $O/pytest.txt | tail -14
