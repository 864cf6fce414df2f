#!/bin/bash
# round 6, call E: 2-deep prefetch in the bf16 weight gradient, two cross-stream waits, ADVICE tests, trajectory test
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train_fused.py tests/test_gpu_kernels.py -q -m gpu -x -s > $O/e_tests.log 2>&1; echo "tests rc=$?"; grep -a "passed\|failed" $O/e_tests.log | tail -2; grep -a "BN statistics" $O/e_tests.log
( cd _bisect/old && python tools/run_bf16_conv.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('OLD', {k: round(v,4) for k,v in d.items() if 'wgrad' in k})" )
python tools/run_bf16_conv.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NEW', {k: round(v,4) for k,v in d.items() if 'wgrad' in k})"
for sync in 0,1,2,3 1,3 3 0,1,2,3 1,3 3; do
  timeout 600 python bench.py --no-train --no-cpu-baseline --no-extra --rb-sync $sync 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline_sparse']; print('rb-sync $sync: fps', d['value'], 'seq', d['fps_sequential'], 'sparse ms', r['ms'], 'frac', r['frac_of_measured_copy_peak'], 'frame', d['frame_graph_ms'])"
done
timeout 600 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new train', d['value'], d['ms_per_step'], d['trials'])"
timeout 900 python -m pytest tests/test_gpu_train.py -q -m gpu -x -s -k "trajector or bf16_step" > $O/e_tests2.log 2>&1; echo "tests2 rc=$?"; grep -a "passed\|failed\|trajectories after\|bf16 step, every" $O/e_tests2.log | cut -c1-900 | tail -5
