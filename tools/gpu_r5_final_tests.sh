#!/bin/bash
# the default GPU suite (what the driver runs at round end) + smoke on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5
timeout 300 python -m pytest tests/test_gpu_sparse_r2.py -q -s -k "float64" > $O/c9_f64.log 2>&1; echo "f64 guard rc=$?"; grep "sparse layers on\|passed\|failed\|Error" $O/c9_f64.log | cut -c1-1500
timeout 1500 python -m pytest tests -q -m gpu > $O/c9_default_tests.log 2>&1; echo "default gpu tests rc=$?"; tail -3 $O/c9_default_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/c9_bench_driver_form.log 2>&1; echo "bench rc=$?"; grep -o '"value": [0-9.]*' $O/c9_bench_driver_form.log | head -3
