#!/bin/bash
# round 5, first GPU call: the new training-parity tests + the sparse segment's kernel timeline (eager two-stream, hipGraph,
# single stream)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_train.py -q -s -x > $O/c1_tests_train.log 2>&1; echo "train tests rc=$?"
grep -n "passed\|failed\|Error\|arbiter\|bf16 step\|bf16:" $O/c1_tests_train.log | cut -c1-1800 | tail -30
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py -q -x > $O/c1_tests_sparse.log 2>&1; echo "sparse tests rc=$?"; tail -3 $O/c1_tests_sparse.log
tl() {  # name, args...
  local name=$1; shift
  rm -rf /tmp/tl_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$name -- python $R/tools/run_sparse_only.py "$@" > $O/c1_tl_${name}.log 2>&1 ); echo "timeline $name rc=$?"
  local DB=$(find /tmp/tl_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/sparse_timeline.py $DB $O/c1_timeline_${name}.txt
}
tl car_eager --config car --reps 5
tl car_graph --config car --reps 5 --graph
tl car_serial --config car --reps 5 --no-overlap
tl multi_graph --config multi --reps 3 --graph
tail -4 $O/c1_timeline_car_graph.txt
