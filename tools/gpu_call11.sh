#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/c11_tests.log 2>&1; echo "tests rc=$?"
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/c11_bench_car.log 2>&1; echo "bench car rc=$?"
timeout 400 python bench.py --config multi --steps 30 --warmup 5 > $O/c11_bench_multi.log 2>&1; echo "bench multi rc=$?"
tail -3 $O/c11_tests.log
