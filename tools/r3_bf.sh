#!/bin/bash
mkdir -p gpurun_out/r3m
timeout 300 python -m pytest tests/test_gpu_bf16.py -x -q -s 2>&1 | tail -15 > gpurun_out/r3m/t_bf16.log; echo "bf16 tests rc=$?"; tail -5 gpurun_out/r3m/t_bf16.log
timeout 200 python tools/run_bf16_conv.py --ablate > gpurun_out/r3m/bf16_ablate.json 2>gpurun_out/r3m/bf16_ablate.err; echo "timing rc=$?"; grep -E "fwd3x3|ablate" gpurun_out/r3m/bf16_ablate.json
