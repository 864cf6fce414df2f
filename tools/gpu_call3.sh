#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py -x -q -m gpu > $O/c3_tests_sparse.log 2>&1; echo "sparse tests rc=$?"
timeout 300 python tools/ablate_spconv.py --config car --ablate > $O/c3_ablate_car.log 2>&1; echo "ablate car rc=$?"
timeout 300 python tools/ablate_spconv.py --config multi > $O/c3_ablate_multi.log 2>&1; echo "ablate multi rc=$?"
timeout 300 python tools/ablate_spconv.py --config waymo > $O/c3_ablate_waymo.log 2>&1; echo "ablate waymo rc=$?"
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/c3_bench_car.log 2>&1; echo "bench car rc=$?"
tail -3 $O/c3_tests_sparse.log
