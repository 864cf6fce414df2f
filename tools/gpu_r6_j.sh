#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
SASSD_FULL_TESTS=1 timeout 1200 python -m pytest tests/test_gpu_train.py::test_training_step_vs_oracle -q -m gpu -s 2>&1 | grep -a "whole-model\|passed\|failed" | cut -c1-260
timeout 1500 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_bf16.py tests/test_gpu_wino4.py tests/test_gpu_pipeline.py -q -m gpu -x -s > $O/j_tests.log 2>&1; echo "tests rc=$?"; grep -a "passed\|failed\|BN statistics\|chain tail" $O/j_tests.log | cut -c1-300 | tail -12
for rep in 1 2; do
  ( cd _bisect/old && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD: fps', d['value'], 'seq', d['fps_sequential'], 'lat', d['latency_ms_sync_per_frame'], 'frame', d['frame_graph_ms'])" )
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW: fps', d['value'], 'seq', d['fps_sequential'], 'lat', d['latency_ms_sync_per_frame'], 'frame', d['frame_graph_ms'], 'bev', d['bev_total_ms'], d['stage_ms'])"
done
timeout 300 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new train', d['value'], d['ms_per_step'])"
