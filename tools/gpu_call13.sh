#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_pipeline.py -x -q -m gpu > $O/c13_tests.log 2>&1; echo "tests rc=$?"
timeout 300 python tools/run_wino4.py > $O/c13_run_wino4.log 2>&1; echo "run rc=$?"
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/c13_bench_car.log 2>&1; echo "bench car rc=$?"
timeout 400 python bench.py --config multi --steps 30 --warmup 5 > $O/c13_bench_multi.log 2>&1; echo "bench multi rc=$?"
tail -3 $O/c13_tests.log; cat $O/c13_run_wino4.log
