#!/bin/bash
# round 5, fourth GPU call: weight gradients beside the next BatchNorm backward only (fenced), A/B against the one-stream step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_train.py -q -s -k "side_stream" > $O/c4_tests.log 2>&1; echo "test rc=$?"; grep -n "passed\|failed\|side stream" $O/c4_tests.log | cut -c1-300
for flag in "" "--serial-wgrad" "" "--serial-wgrad" "" "--serial-wgrad"; do
  timeout 300 python bench.py --mode train --steps 40 --warmup 8 $flag > $O/c4_bench_train.log 2>&1
  echo "train [$flag] rc=$?"; python - <<PY
import json
l=[x for x in open("$O/c4_bench_train.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("train [$flag]", d["value"], d["ms_per_step"], d["trials"])
PY
done
timeout 300 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/c4_bench_train_waymo.log 2>&1; grep -o '"value": [0-9.]*, "unit": "samples/s"' $O/c4_bench_train_waymo.log
timeout 300 python bench.py --mode train --config waymo --steps 12 --warmup 4 --serial-wgrad > $O/c4_bench_train_waymo_serial.log 2>&1; grep -o '"value": [0-9.]*, "unit": "samples/s"' $O/c4_bench_train_waymo_serial.log
