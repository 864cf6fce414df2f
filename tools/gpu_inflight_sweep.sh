#!/bin/bash
# frames in flight per GPU: frame rate at several --inflight values on one box
# usage: bash tools/gpu_inflight_sweep.sh [config] [steps] [values...]      (default: car 200 3 2 4 5 3)
cd "$GRAFT_REPO_ROOT" || exit 1
CFG=${1:-car}; STEPS=${2:-200}; shift; shift
VALS=${@:-3 2 4 5 3}
O=gpurun_out/inflight; mkdir -p $O
for n in $VALS; do
  timeout 300 python bench.py --config $CFG --steps $STEPS --warmup 5 --no-train --no-cpu-baseline --inflight $n > $O/${CFG}_b$n.json 2> $O/${CFG}_b$n.err
  python - $CFG $n <<'PY'
import json, sys
for l in open("gpurun_out/inflight/%s_b%s.json" % (sys.argv[1], sys.argv[2])):
    if l.startswith("{"):
        d = json.loads(l); print(sys.argv[1], "inflight", sys.argv[2], d["value"], "frames/s", d["ms_per_step"], "ms", d["trials"]["ms_per_step_min"], d["trials"]["ms_per_step_max"])
PY
done
