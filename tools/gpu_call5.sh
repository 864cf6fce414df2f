#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py -x -q -m gpu > $O/c5_tests_sparse.log 2>&1; echo "sparse tests rc=$?"
timeout 300 python tools/ablate_spconv.py --config car > $O/c5_ablate_car.log 2>&1; echo "ablate car rc=$?"
timeout 300 python tools/ablate_spconv.py --config multi > $O/c5_ablate_multi.log 2>&1; echo "ablate multi rc=$?"
timeout 300 python tools/ablate_spconv.py --config waymo > $O/c5_ablate_waymo.log 2>&1; echo "ablate waymo rc=$?"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c5 -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --inflight 1 > $O/c5_prof.log 2>&1; echo "prof rc=$?"
cd $GRAFT_REPO_ROOT
DB=$(find /tmp/prof_c5 -name "*.db" | head -1)
if [ -n "$DB" ]; then python tools/rocprof_summary.py $DB > $O/c5_kernel_stats.txt 2>&1; fi
tail -3 $O/c5_tests_sparse.log; grep "rulebooks\|sum of" $O/c5_ablate_*.log
