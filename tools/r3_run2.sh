cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3b
O=gpurun_out/r3b
timeout 600 python -m pytest tests/test_gpu_train_fused.py -q -x -s > $O/t_fused.log 2>&1; echo "fused rc=$?"
timeout 900 python -m pytest tests/test_gpu_train.py -q -s -k "training_step or waymo" > $O/t_train.log 2>&1; echo "train rc=$?"
timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train.log 2> $O/bench_train.err; echo "train bench rc=$?"
timeout 300 python bench.py --mode train --steps 40 --warmup 8 --torch-bn > $O/bench_train_torchbn.log 2> $O/bench_train_torchbn.err; echo "train bench torch-bn rc=$?"
timeout 300 python bench.py --mode train --config waymo --steps 10 --warmup 3 > $O/bench_train_waymo.log 2> $O/bench_train_waymo.err; echo "waymo train rc=$?"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 4 > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1 ); echo "prof rc=$?"
DB=$(find /tmp/pf_train -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB > $O/train_kernel_stats.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_trainw -- python $GRAFT_REPO_ROOT/bench.py --mode train --config waymo --steps 4 --warmup 2 > $GRAFT_REPO_ROOT/$O/prof_train_waymo.log 2>&1 ); echo "prof waymo rc=$?"
DB=$(find /tmp/pf_trainw -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB > $O/train_waymo_kernel_stats.txt 2>&1
tail -n 4 $O/t_fused.log; tail -n 4 $O/t_train.log
grep -o '"value": [0-9.]*' $O/bench_train.log $O/bench_train_torchbn.log $O/bench_train_waymo.log
head -30 $O/train_kernel_stats.txt
