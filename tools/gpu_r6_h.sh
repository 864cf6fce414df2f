#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_sparse_r2.py -q -m gpu -x > $O/h_tests.log 2>&1; echo "tests rc=$?"; grep -a "passed\|failed\|Error" $O/h_tests.log | tail -5
for rep in 1 2; do
  ( cd _bisect/old && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD: fps', d['value'], 'seq', d['fps_sequential'], 'lat', d['latency_ms_sync_per_frame'], 'frame', d['frame_graph_ms'], 'vox', d['stage_ms']['voxelize'])" )
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-train 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW: fps', d['value'], 'seq', d['fps_sequential'], 'lat', d['latency_ms_sync_per_frame'], 'frame', d['frame_graph_ms'], 'stage', d['stage_ms'])"
done
