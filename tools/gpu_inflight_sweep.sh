#!/bin/bash
# frames in flight per GPU: K21 frame rate at --inflight 2..5 on one box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/inflight; mkdir -p $O
for n in 3 2 4 5 3; do
  timeout 200 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline --inflight $n > $O/b$n.json 2> $O/b$n.err
  python - $n <<'PY'
import json, sys
for l in open("gpurun_out/inflight/b%s.json" % sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l); print("inflight", sys.argv[1], d["value"], "frames/s", d["ms_per_step"], "ms", d["trials"]["ms_per_step_min"], d["trials"]["ms_per_step_max"])
PY
done
