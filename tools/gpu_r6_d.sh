#!/bin/bash
# round 6, call D: narrow 1x1 convs + conv0 active tiles -- tests, then the default bench in old / new trees on ONE box
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_kernels.py tests/test_gpu_pipeline.py tests/test_gpu_multi.py -q -m gpu -x -s > $O/d_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/d_tests.log; grep -a "active tiles\|max abs errors" $O/d_tests.log | cut -c1-220 | head -20
for rep in 1 2; do
  ( cd _bisect/old && timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('OLD default: fps', d['value'], 'seq', d['fps_sequential'], 'sparse', d['roofline_sparse']['ms'], 'bev', d['bev_total_ms'], 'train', d['train']['value'], d['train']['ms_per_step'], d['train']['trials'])" )
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('NEW default: fps', d['value'], 'seq', d['fps_sequential'], 'sparse', d['roofline_sparse']['ms'], 'bev', d['bev_total_ms'], 'stage', d['stage_ms'], 'train', d['train']['value'], d['train']['ms_per_step'], d['train']['trials'])"
done
