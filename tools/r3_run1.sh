cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
O=gpurun_out/r3a
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -x -k "configured_thresholds" -s > $O/t_thr.log 2>&1; echo "thr rc=$?"
timeout 900 python -m pytest tests/test_gpu_train.py -q -x -k "waymo_vs_oracle" -s > $O/t_waymo.log 2>&1; echo "waymo rc=$?"
timeout 900 python -m pytest tests/test_gpu_train.py -q -k "test_training_step_vs_oracle" -s > $O/t_step.log 2>&1; echo "step rc=$?"
timeout 600 python bench.py --steps 50 --warmup 10 > $O/bench_default.log 2> $O/bench_default.err; echo "bench rc=$?"
timeout 600 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2> $O/bench_train_waymo.err; echo "waymo train rc=$?"
tail -3 $O/t_thr.log $O/t_waymo.log $O/t_step.log
tail -c 1500 $O/bench_train_waymo.log
