cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16.py tests/test_gpu_train.py -q -s -k "bf16 or disk_to_ap or waymo_scale" > gpurun_out/t_run7.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/t_run7.log
pmc() {  # name, counter, command...
  local name=$1; local ctr=$2; shift; shift
  rm -rf /tmp/pm_$name; ( cd /tmp && timeout 180 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm_$name -- "$@" > $O/${name}_${ctr}.log 2>&1 ); echo "$name $ctr rc=$?"
  local DB=$(find /tmp/pm_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc.py $DB > $O/${name}_${ctr}.json 2>&1
}
trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pf_$name; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -- "$@" > $O/${name}_under_rocprof.log 2>&1 ); echo "$name rc=$?"
  local DB=$(find /tmp/pf_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/${name}_kernel_stats.txt 2>&1
}
timeout 600 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2>&1; echo "train waymo rc=$?"
trace bench_train_waymo_trace python $R/bench.py --mode train --config waymo --steps 4 --warmup 2
trace bench_train python $R/bench.py --mode train --steps 10 --warmup 4
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_bf16.log 2>&1; echo "train bf16 rc=$?"
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
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench_default.log 2>&1; echo "bench default rc=$?"
grep -o '"value": [0-9.]*' $O/bench_train_waymo.log $O/bench_train_bf16.log $O/bench_default.log
