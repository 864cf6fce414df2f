#!/bin/bash
# round-2 GPU call 1: parity of the new sparse path + ablation + baseline benches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py -x -q -m gpu > $O/c1_tests_sparse.log 2>&1; echo "sparse tests rc=$?" | tee -a $O/c1_summary.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "rulebook or spconv or voxel" > $O/c1_tests_kernels.log 2>&1; echo "kernel tests rc=$?" | tee -a $O/c1_summary.log
timeout 600 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $O/c1_tests_pipeline.log 2>&1; echo "pipeline tests rc=$?" | tee -a $O/c1_summary.log
timeout 300 python tools/ablate_spconv.py --config car --ablate > $O/c1_ablate_car.log 2>&1; echo "ablate car rc=$?" | tee -a $O/c1_summary.log
timeout 300 python tools/ablate_spconv.py --config multi > $O/c1_ablate_multi.log 2>&1; echo "ablate multi rc=$?" | tee -a $O/c1_summary.log
timeout 300 python tools/ablate_spconv.py --config waymo > $O/c1_ablate_waymo.log 2>&1; echo "ablate waymo rc=$?" | tee -a $O/c1_summary.log
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/c1_bench_car.log 2>&1; echo "bench car rc=$?" | tee -a $O/c1_summary.log
timeout 400 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --eager > $O/c1_bench_car_eager.log 2>&1; echo "bench car eager rc=$?" | tee -a $O/c1_summary.log
timeout 400 python bench.py --config multi --steps 30 --warmup 5 > $O/c1_bench_multi.log 2>&1; echo "bench multi rc=$?" | tee -a $O/c1_summary.log
timeout 400 python bench.py --config waymo --steps 30 --warmup 5 > $O/c1_bench_waymo.log 2>&1; echo "bench waymo rc=$?" | tee -a $O/c1_summary.log
tail -3 $O/c1_tests_*.log
