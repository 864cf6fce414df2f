#!/bin/bash
# round 6, call A: the two-phase-per-level pyramid and its persistent form -- parity, then the A/B of the sparse segment
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
timeout 900 python -m pytest tests/test_gpu_sparse_r2.py tests/test_gpu_pipeline.py -q -m gpu -x > $O/a_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/a_tests.log
for pyr in levels persistent levels persistent; do
  timeout 600 python bench.py --no-train --no-cpu-baseline --no-extra --pyramid $pyr > $O/a_bench_$pyr.json 2> $O/a_bench_$pyr.err; echo "bench $pyr rc=$?"
  python - <<PY
import json
d=json.load(open("$O/a_bench_$pyr.json"))
r=d["roofline_sparse"]
print("$pyr", "fps", d["value"], "seq", d["fps_sequential"], "sparse ms", r["ms"], "convs", r["ms_convs_only"], "pyr", r["ms_pyramid_only"], "frac", r["frac_of_measured_copy_peak"], "frame", d["frame_graph_ms"], "bev", d["bev_total_ms"])
PY
done
