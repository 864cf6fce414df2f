#!/bin/bash
# round 5, second GPU call: parity calibration dump, the tests touched by the BN / fill / issue-order changes, sparse timeline,
# default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5; R=$GRAFT_REPO_ROOT
rm -f $O/c2_parity_dump.txt
SASSD_PARITY_DUMP=$O/c2_parity_dump.txt timeout 1500 python -m pytest tests/test_gpu_train.py -q -s > $O/c2_tests_train.log 2>&1; echo "train tests rc=$?"
grep -n "passed\|failed\|^E  \|arbiter\|bf16 step\|bf16:" $O/c2_tests_train.log | cut -c1-1500 | tail -30
timeout 900 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_pipeline.py tests/test_gpu_kernels.py tests/test_gpu_sparse_r2.py -q -x > $O/c2_tests_other.log 2>&1; echo "other tests rc=$?"; tail -3 $O/c2_tests_other.log
tl() {  # name, args...
  local name=$1; shift
  rm -rf /tmp/tl_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/tl_$name -- python $R/tools/run_sparse_only.py "$@" > $O/c2_tl_${name}.log 2>&1 ); echo "timeline $name rc=$?"
  local DB=$(find /tmp/tl_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/sparse_timeline.py $DB $O/c2_timeline_${name}.txt
}
tl car_graph --config car --reps 5 --graph
tl car_eager --config car --reps 5
tail -2 $O/c2_timeline_car_graph.txt
timeout 900 python bench.py > $O/c2_bench_default.log 2>&1; echo "bench rc=$?"; tail -c 1500 $O/c2_bench_default.log
