#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "reference_defaults or nms_normal" > $O/c17_tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/c17_tests.log
timeout 600 python bench.py --mode train --steps 20 --warmup 6 > $O/c17_bench_train.log 2>&1; echo "bench train rc=$?"; tail -1 $O/c17_bench_train.log
timeout 600 python tools/time_train_phases.py 10 > $O/c17_phases.log 2>&1; echo "phases rc=$?"; tail -2 $O/c17_phases.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_c17 -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 4 > $O/c17_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_c17 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB > $O/c17_kernel_stats_train.txt 2>&1; fi
head -45 $O/c17_kernel_stats_train.txt | cut -c1-80,100-170
