#!/bin/bash
# round 4, GPU call D: sparse parity with the new default, bench lines (car / multi / waymo / train) new default vs round-3 geometry
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_sparse_r2.py tests/test_gpu_kernels.py -q -k "spconv" -x > $O/pytest_sparse.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sparse.txt
for cfg in 0; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline --spconv-cfg $cfg > $O/bench_car_cfg$cfg.json 2> $O/bench_car_cfg$cfg.err; echo "bench car cfg $cfg rc $?"
  timeout 300 python bench.py --config multi --steps 30 --warmup 5 --no-cpu-baseline --spconv-cfg $cfg > $O/bench_multi_cfg$cfg.json 2> $O/bench_multi_cfg$cfg.err; echo "bench multi cfg $cfg rc $?"
  timeout 300 python bench.py --mode train --steps 30 --warmup 6 --spconv-cfg $cfg > $O/bench_train_cfg$cfg.json 2> $O/bench_train_cfg$cfg.err; echo "bench train cfg $cfg rc $?"
done
timeout 300 python bench.py --config waymo --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_waymo_cfg0.json 2> $O/bench_waymo_cfg0.err; echo "bench waymo rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4d/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); rs = d.get("roofline_sparse") or {}
            print(f.split("/")[-1], d["metric"][:30], d["value"], "ms/step", d["ms_per_step"], "sparse ms", rs.get("ms"), "frac_copy", rs.get("frac_of_measured_copy_peak"))
PY
