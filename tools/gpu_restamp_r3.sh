cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
pmc() {  # name, counter(s), command...
  local name=$1; local ctr=$2; shift; shift
  local tag=$(echo $ctr | cut -d' ' -f1)
  rm -rf /tmp/pm_$name; ( cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm_$name -- "$@" > $O/${name}_${tag}.log 2>&1 ); echo "$name $tag rc=$?"
  local DB=$(find /tmp/pm_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc.py $DB > $O/${name}_${tag}.json 2>&1
}
trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pf_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -- "$@" > $O/${name}_under_rocprof.log 2>&1 ); echo "$name rc=$?"
  local DB=$(find /tmp/pf_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/${name}_kernel_stats.txt 2>&1
}
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
SQ="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
pmc stall_bf16conv "$SQ" python $R/tools/run_bf16_conv.py --iters 5
pmc ldsclk_bf16conv "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" python $R/tools/run_bf16_conv.py --only-fwd --iters 10
pmc ldsclk_bf16conv "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" python $R/tools/run_bf16_conv.py --only-fwd --iters 10
pmc stall_sparse_car "$SQ" python $R/tools/run_sparse_only.py --config car --reps 5
pmc stall_wino4 "$SQ" python $R/tools/run_wino4.py --profile --reps 5
python tools/collect_profiles.py r03 > $O/collect.log 2>&1; echo "collect rc=$?"
trace bench_train python $R/bench.py --mode train --steps 10 --warmup 4
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_bf16.log 2>&1; echo "train bf16 rc=$?"
timeout 600 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2>&1; echo "train waymo rc=$?"
timeout 300 python tools/run_bf16_conv.py --ablate > $O/bf16_conv_timing.json 2>/dev/null; echo "bf16 timing rc=$?"
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_default.log 2>&1; echo "bench default rc=$?"
grep -o '"traffic_measured_at": [^,]*' $O/bench_default.log | head; grep -o '"value": [0-9.]*' $O/bench_default.log $O/bench_train_bf16.log $O/bench_train_waymo.log
