#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py -x -q -m gpu -k "pyramid" > $O/c4_tests_sparse.log 2>&1; echo "pyramid tests rc=$?"
timeout 300 python tools/ablate_spconv.py --config car > $O/c4_ablate_car.log 2>&1; echo "ablate car rc=$?"
timeout 300 python tools/ablate_spconv.py --config multi > $O/c4_ablate_multi.log 2>&1; echo "ablate multi rc=$?"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --inflight 1 > $O/c4_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_c4 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB > $O/c4_kernel_stats.txt 2>&1; fi
tail -3 $O/c4_tests_sparse.log; grep rulebooks $O/c4_ablate_*.log
