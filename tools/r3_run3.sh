cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
O=gpurun_out/r3c
timeout 600 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_pointops.py tests/test_gpu_wino4.py -q -s > $O/t_kernels.log 2>&1; echo "kernels rc=$?"
timeout 900 python -m pytest tests/test_gpu_train.py -q -s -k "training_step or waymo" > $O/t_train.log 2>&1; echo "train rc=$?"
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -s -k "pipeline_vs_oracle or configured or multi_class" > $O/t_pipe.log 2>&1; echo "pipe rc=$?"
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --config multi --steps 20 --warmup 5 > $O/bench_multi.log 2> $O/bench_multi.err; echo "bench multi rc=$?"
timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train.log 2> $O/bench_train.err; echo "train bench rc=$?"
timeout 300 python bench.py --mode train --config waymo --steps 10 --warmup 3 > $O/bench_train_waymo.log 2> $O/bench_train_waymo.err; echo "waymo train rc=$?"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 4 > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1 ); echo "prof rc=$?"
DB=$(find /tmp/pf_train -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB > $O/train_kernel_stats.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_inf -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/$O/prof_infer.log 2>&1 ); echo "prof infer rc=$?"
DB=$(find /tmp/pf_inf -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB > $O/infer_kernel_stats.txt 2>&1
tail -n 5 $O/t_kernels.log; tail -n 5 $O/t_train.log; tail -n 5 $O/t_pipe.log
grep -o '"value": [0-9.]*' $O/bench_default.log $O/bench_multi.log $O/bench_train.log $O/bench_train_waymo.log
