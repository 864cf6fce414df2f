cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
O=gpurun_out/r3d
timeout 600 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_train_fused.py -q -s -k "chain or bn_relu or bn2d" > $O/t_kernels.log 2>&1; echo "kernels rc=$?"
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -s -k "pipeline_vs_oracle or configured or multi_class or waymo" > $O/t_pipe.log 2>&1; echo "pipe rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
timeout 300 python bench.py --config multi --steps 20 --warmup 5 > $O/bench_multi.log 2> $O/bench_multi.err; echo "bench multi rc=$?"
timeout 300 python bench.py --config waymo --steps 20 --warmup 5 > $O/bench_waymo.log 2> $O/bench_waymo.err; echo "bench waymo rc=$?"
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_inf -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-train > $GRAFT_REPO_ROOT/$O/prof_infer.log 2>&1 ); echo "prof infer rc=$?"
DB=$(find /tmp/pf_inf -name "*.db" | head -1); [ -n "$DB" ] && python tools/rocprof_summary.py $DB > $O/infer_kernel_stats.txt 2>&1
tail -n 6 $O/t_kernels.log; tail -n 8 $O/t_pipe.log | cut -c1-300
grep -o '"value": [0-9.]*' $O/bench_default.log $O/bench_multi.log $O/bench_waymo.log
head -24 $O/infer_kernel_stats.txt | cut -c1-160
