#!/bin/bash
# round-6 evidence on FINAL kernel sources (csrc_hash stamps every record): the complete GPU test set incl. the slow-marked
# case, the default bench line, rocprofv3 kernel-trace summaries, PMC passes (FETCH_SIZE / WRITE_SIZE,
# SQ stall counters), per-layer sparse-conv timings, the other configs' bench lines.
# usage (on the GPU box): bash tools/gpu_profiles_r6.sh [notests]   -> gpurun_out/prof/*, copied to profiles/ by collect_profiles.py r06
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
if [ "$1" != "notests" ]; then
  SASSD_FULL_TESTS=1 timeout 1500 python -m pytest tests -q -m gpu -s > $O/full_tests.log 2>&1; echo "full gpu tests rc=$?"
  ( echo "# SASSD_FULL_TESTS=1 python -m pytest tests -q -m gpu -s on csrc $(python -c 'import sassd; from sassd import _C; print(_C.csrc_hash())'), commit $(git rev-parse --short HEAD 2>/dev/null || echo "${GRAFT_COMMIT:-unknown}")";
    grep -a "passed\|failed\| error\|vs float64 arbiter\|vs the .* arbiter\|bf16 step, every\|max abs errors\|waymo-scale training step" $O/full_tests.log | cut -c1-2500 ) > $O/full_tests_tail.txt
  tail -3 $O/full_tests_tail.txt | cut -c1-300
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
fi
trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pf_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -- "$@" > $O/${name}_under_rocprof.log 2>&1 ); echo "$name rc=$?"
  local DB=$(find /tmp/pf_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/${name}_kernel_stats.txt 2>&1
}
pmc() {  # name, counter(s), command...
  local name=$1; local ctr=$2; shift; shift
  local tag=$(echo $ctr | cut -d' ' -f1)
  rm -rf /tmp/pm_$name; ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm_$name -- "$@" > $O/${name}_${tag}.log 2>&1 ); echo "$name $tag rc=$?"
  local DB=$(find /tmp/pm_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc.py $DB > $O/${name}_${tag}.json 2>&1
}
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
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
pmc stall_bf16conv "$SQ" python $R/tools/run_bf16_conv.py --iters 5
pmc stall_sparse_car "$SQ" python $R/tools/run_sparse_only.py --config car --reps 5
pmc stall_sparse_multi "$SQ" python $R/tools/run_sparse_only.py --config multi --reps 3
pmc stall_wino4 "$SQ" python $R/tools/run_wino4.py --profile --reps 5
python tools/collect_profiles.py r06 > $O/collect.log 2>&1; echo "collect rc=$?"
timeout 900 python bench.py > $O/bench_default.log 2>&1; echo "bench default rc=$?"
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_20steps.log 2>&1 ) 2>&1 | grep real; echo "bench 20 steps (the driver's form) rc=$?"
trace bench_inflight3 python $R/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-train --no-extra
trace bench_train python $R/bench.py --mode train --steps 10 --warmup 4
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_bf16.log 2>&1; echo "train bf16 rc=$?"
timeout 600 python bench.py --mode train --steps 40 --warmup 8 --force-ddp > $O/bench_train_forceddp.log 2>&1; echo "train force-ddp rc=$?"
timeout 600 python bench.py --mode train --precision fp32 --steps 30 --warmup 6 > $O/bench_train_fp32.log 2>&1; echo "train fp32 rc=$?"
timeout 600 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2>&1; echo "train waymo rc=$?"
timeout 300 python bench.py --config multi --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_multi.log 2>&1; echo "multi rc=$?"
timeout 300 python bench.py --config waymo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_waymo.log 2>&1; echo "waymo rc=$?"
timeout 300 python tools/run_bf16_conv.py --ablate > $O/bf16_conv_timing.json 2>/dev/null; echo "bf16 timing rc=$?"
timeout 200 python tools/run_conv1x1_bf16.py 30 2>/dev/null | grep "^1x1" > $O/conv1x1_bf16_timing.txt; echo "bf16 1x1 timing rc=$?"
timeout 400 python tools/ablate_spconv.py --config car --ablate 2>&1 | grep -v "^/opt" > $O/spconv_layers_car.txt; echo "layers car rc=$?"
python tools/collect_profiles.py r06 >> $O/collect.log 2>&1; echo "collect rc=$?"
grep -o '"traffic_measured_at": [^,]*' $O/bench_default.log | head -3; grep -o '"value": [0-9.]*' $O/bench_default.log $O/bench_train_bf16.log $O/bench_multi.log $O/bench_waymo.log $O/bench_train_waymo.log $O/bench_train_fp32.log
