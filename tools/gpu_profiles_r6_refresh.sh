#!/bin/bash
# round 6: host-side changes after the last kernel change (csrc_hash unchanged: the PMC / stall records of gpu_profiles_r6.sh stay
# valid) -- re-collect what they move: the complete GPU test set, the default line, the driver's 20-step form, the training trace
# and the training lines.  usage (on the GPU box): bash tools/gpu_profiles_r6_refresh.sh   -> gpurun_out/prof/*
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
SASSD_FULL_TESTS=1 timeout 1500 python -m pytest tests -q -m gpu -s > $O/full_tests.log 2>&1; echo "full gpu tests rc=$?"
( echo "# SASSD_FULL_TESTS=1 python -m pytest tests -q -m gpu -s on csrc $(python -c 'import sassd; from sassd import _C; print(_C.csrc_hash())'), commit $(git rev-parse --short HEAD 2>/dev/null || echo "${GRAFT_COMMIT:-unknown}")";
  grep -a "passed\|failed\| error\|vs float64 arbiter\|vs the .* arbiter\|bf16 step, every\|max abs errors\|waymo-scale training step" $O/full_tests.log | cut -c1-2500 ) > $O/full_tests_tail.txt
tail -1 $O/full_tests_tail.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench default rc=$?"
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20steps.log 2>&1 ) 2>&1 | grep real; echo "bench 20 steps rc=$?"
rm -rf /tmp/pf_t; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_t -- python $R/bench.py --mode train --steps 10 --warmup 4 > $O/bench_train_under_rocprof.log 2>&1 ); echo "train trace rc=$?"
DB=$(find /tmp/pf_t -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/bench_train_kernel_stats.txt 2>&1
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_bf16.log 2>&1; echo "train bf16 rc=$?"
timeout 600 python bench.py --mode train --steps 40 --warmup 8 --force-ddp > $O/bench_train_forceddp.log 2>&1; echo "train force-ddp rc=$?"
timeout 600 python bench.py --mode train --precision fp32 --steps 30 --warmup 6 > $O/bench_train_fp32.log 2>&1; echo "train fp32 rc=$?"
timeout 600 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2>&1; echo "train waymo rc=$?"
timeout 300 python bench.py --config multi --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_multi.log 2>&1; echo "multi rc=$?"
timeout 300 python bench.py --config waymo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_waymo.log 2>&1; echo "waymo rc=$?"
rm -rf /tmp/pf_i; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_i -- python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train --no-extra > $O/bench_inflight3_under_rocprof.log 2>&1 ); echo "inference trace rc=$?"
DB=$(find /tmp/pf_i -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/bench_inflight3_kernel_stats.txt 2>&1
grep -o '"value": [0-9.]*' $O/bench_default.log $O/bench_20steps.log $O/bench_train_bf16.log $O/bench_train_waymo.log $O/bench_train_fp32.log $O/bench_train_forceddp.log $O/bench_multi.log $O/bench_waymo.log
