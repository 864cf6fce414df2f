#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
timeout 1500 python -m pytest tests/test_gpu_bf16.py -q -m gpu -x -s > $O/g_tests.log 2>&1; echo "tests rc=$?"; grep -a "passed\|failed\|Error\|error" $O/g_tests.log | tail -5
for rep in 1 2; do
  ( cd _bisect/old && timeout 300 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old train', d['value'], d['ms_per_step'], d['trials'])" )
  timeout 300 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new train', d['value'], d['ms_per_step'], d['trials'], d['loss_terms'])"
done
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_train_fused.py -q -m gpu -x > $O/g_tests2.log 2>&1; echo "tests2 rc=$?"; grep -a "passed\|failed" $O/g_tests2.log | tail -3
for nf in 2 3 4; do
  timeout 600 python bench.py --no-train --no-cpu-baseline --no-extra --inflight $nf 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inflight $nf: fps', d['value'], 'seq', d['fps_sequential'])"
done
