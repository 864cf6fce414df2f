cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3k; O=gpurun_out/r3k
timeout 600 python -m pytest tests/test_gpu_bf16.py -q -s > $O/t_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -3 $O/t_bf16.log
timeout 300 python tools/run_bf16_conv.py --ablate > $O/bf16_conv_timing.json 2> $O/bf16_conv_timing.err; echo "timing rc=$?"; grep "fwd3x3_bf16\|kernel_8plus4\|no_mfma\|no_stores\|no_input" $O/bf16_conv_timing.json
timeout 600 python -m pytest tests/test_gpu_train.py -q -s -k "bf16" > $O/t_train_bf16.log 2>&1; echo "bf16 train parity rc=$?"; tail -2 $O/t_train_bf16.log | cut -c1-200
timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train.log 2> $O/bench_train.err; echo "train rc=$?"
timeout 300 python bench.py --mode train --config waymo --steps 12 --warmup 4 > $O/bench_train_waymo.log 2> $O/bench_train_waymo.err; echo "train waymo rc=$?"
grep -o '"value": [0-9.]*' $O/bench_train.log $O/bench_train_waymo.log
