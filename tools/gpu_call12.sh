#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_wino4.py -x -q -m gpu > $O/c12_tests_wino4.log 2>&1; echo "wino4 tests rc=$?"
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/c12_bench_car.log 2>&1; echo "bench car rc=$?"
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_c12 -- python $GRAFT_REPO_ROOT/bench.py --config multi --steps 20 --warmup 5 --inflight 1 > $O/c12_prof_multi.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_c12 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB > $O/c12_kernel_stats_multi.txt 2>&1; fi
tail -3 $O/c12_tests_wino4.log
