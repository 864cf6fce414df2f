#!/bin/bash
# round-3 evidence: bench JSON lines, rocprofv3 kernel-trace summaries, PMC (FETCH_SIZE / WRITE_SIZE) passes.
# usage (on the GPU box): bash tools/gpu_profiles_r3.sh [quick]      -> gpurun_out/prof/*, then tools/collect_profiles.py r03
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pf_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -- "$@" > $O/${name}_under_rocprof.log 2>&1 ); echo "$name rc=$?"
  local DB=$(find /tmp/pf_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/${name}_kernel_stats.txt 2>&1
}
pmc() {  # name, counter, command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pm_$name; ( cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm_$name -- "$@" > $O/${name}_${ctr}.log 2>&1 ); echo "$name $ctr rc=$?"
  local DB=$(find /tmp/pm_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc.py $DB > $O/${name}_${ctr}.json 2>&1
}
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_default.log 2>&1; echo "bench default rc=$?"
trace bench_inflight3 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train
trace bench_inflight1 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train --inflight 1
trace bench_multi python $R/bench.py --config multi --steps 20 --warmup 5
trace bench_waymo python $R/bench.py --config waymo --steps 20 --warmup 5
trace bench_train python $R/bench.py --mode train --steps 10 --warmup 4
trace bench_train_waymo_trace python $R/bench.py --mode train --config waymo --steps 4 --warmup 2
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_bf16.log 2>&1; echo "train bf16 rc=$?"
timeout 600 python bench.py --mode train --precision fp32 --steps 30 --warmup 6 > $O/bench_train_fp32.log 2>&1; echo "train fp32 rc=$?"
timeout 600 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2>&1; echo "train waymo rc=$?"
timeout 300 python tools/run_bf16_conv.py --ablate > $O/bf16_conv_timing.json 2>/dev/null; echo "bf16 timing rc=$?"
pmc wino4 FETCH_SIZE python $R/tools/run_wino4.py --profile --reps 5
pmc wino4 WRITE_SIZE python $R/tools/run_wino4.py --profile --reps 5
pmc sparse_car FETCH_SIZE python $R/tools/run_sparse_only.py --config car --reps 5
pmc sparse_car WRITE_SIZE python $R/tools/run_sparse_only.py --config car --reps 5
pmc sparse_multi FETCH_SIZE python $R/tools/run_sparse_only.py --config multi --reps 3
pmc sparse_multi WRITE_SIZE python $R/tools/run_sparse_only.py --config multi --reps 3
pmc sparse_waymo FETCH_SIZE python $R/tools/run_sparse_only.py --config waymo --reps 2
pmc sparse_waymo WRITE_SIZE python $R/tools/run_sparse_only.py --config waymo --reps 2
pmc bf16conv FETCH_SIZE python $R/tools/run_bf16_conv.py --iters 5
pmc bf16conv WRITE_SIZE python $R/tools/run_bf16_conv.py --iters 5
ls -la $O | head -60
