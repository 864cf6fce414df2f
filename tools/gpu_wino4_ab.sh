#!/bin/bash
# round 4, GPU call K: split-operand Winograd GEMM as the default -- kernel + pipeline parity, frame rate against the fp32-MFMA kernel
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/wino4_ab; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_pipeline.py -q -x -s > $O/pytest.txt 2>&1; echo "pytest rc $?"; grep "passed\|failed\|Error\|max abs\|box" $O/pytest.txt | tail -30 | cut -c1-220
timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline > $O/bench_car.json 2> $O/bench_car.err; echo "car rc $?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline --wino4-cfg 1 > $O/bench_car_fp32mfma.json 2> $O/bench_car_fp32mfma.err; echo "car fp32 rc $?"
timeout 300 python bench.py --config multi --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_multi.json 2> $O/bench_multi.err; echo "multi rc $?"
timeout 300 python bench.py --config waymo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_waymo.json 2> $O/bench_waymo.err; echo "waymo rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/wino4_ab/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); r = d["roofline"]; rs = d.get("roofline_sparse") or {}
            print(f.split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "gemm ms", r["ms_per_launch"], "frac", r["frac"], "frac_fp32", r.get("frac_of_fp32_mfma_peak"), "layer", r["layer"]["ms"], "sparse", rs.get("ms"))
PY
