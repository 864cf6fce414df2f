#!/bin/bash
# round 4, GPU call M: frame rate on one more box (box-to-box spread check)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4m; mkdir -p $O
timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline > $O/bench_car.json 2> $O/bench_car.err; echo "car rc $?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline --wino4-cfg 1 > $O/bench_car_fp32.json 2> $O/bench_car_fp32.err; echo "car fp32-MFMA rc $?"
python - <<'PY'
import json
for f in ("bench_car", "bench_car_fp32"):
  for l in open("gpurun_out/r4m/%s.json" % f):
    if l.startswith("{"):
        d = json.loads(l); print(f, d["value"], "ms/step", d["ms_per_step"], "conv7", d["stage_ms"]["bev_conv7"], "bev total", d["bev_total_ms"])
PY
