#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wino4.py -q -m gpu -x > $O/k_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/k_tests.log
for rep in 1 2 3; do
  for dbg in 256 0; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-train --wino4-dbg $dbg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('dbg $dbg: fps', d['value'], 'seq', d['fps_sequential'], 'lat', d['latency_ms_sync_per_frame'], 'frame', d['frame_graph_ms'], 'bev', d['bev_total_ms'])"
  done
done
