# refresh of the training-step kernel table only (python-side changes leave the PMC stamps valid)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof; R=$GRAFT_REPO_ROOT
rm -rf /tmp/pf_bt; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_bt -- python $R/bench.py --mode train --steps 10 --warmup 4 > $O/bench_train_under_rocprof.log 2>&1 ); echo "trace rc=$?"
DB=$(find /tmp/pf_bt -name "*.db" | head -1)
[ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/bench_train_kernel_stats.txt 2>&1
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_bf16.log 2>&1; echo "train bf16 rc=$?"
grep -o '"value": [0-9.]*' $O/bench_train_bf16.log | head -1; head -3 $O/bench_train_kernel_stats.txt | cut -c1-150
