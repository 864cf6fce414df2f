#!/bin/bash
# round 6, call B: byte-map pyramid (levels / persistent), per-call cfg refactor -- sparse + wino4 + pipeline tests, A/B, the full default line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
timeout 1200 python -m pytest tests/test_gpu_sparse_r2.py tests/test_gpu_pipeline.py tests/test_gpu_wino4.py tests/test_gpu_kernels.py -q -m gpu -x > $O/b_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/b_tests.log
for pyr in levels persistent levels; do
  timeout 600 python bench.py --no-train --no-cpu-baseline --no-extra --pyramid $pyr > $O/b_bench_$pyr.json 2> $O/b_bench_$pyr.err; echo "bench $pyr rc=$?"
  python - <<PY
import json
d=json.load(open("$O/b_bench_$pyr.json"))
r=d["roofline_sparse"]
print("$pyr", "fps", d["value"], "seq", d["fps_sequential"], "sparse ms", r["ms"], "convs", r["ms_convs_only"], "pyr", r["ms_pyramid_only"], "frac", r["frac_of_measured_copy_peak"], "frame", d["frame_graph_ms"], "bev", d["bev_total_ms"])
PY
done
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/b_bench_default.json 2> $O/b_bench_default.err ) 2>&1 | grep real; echo "default bench rc=$?"
python - <<PY
import json
d=json.load(open("$O/b_bench_default.json"))
print("default: fps", d["value"], "train", d.get("train",{}).get("value"), d.get("train",{}).get("ms_per_step"))
for k in ("infer_multi","infer_waymo","train_waymo"):
    r=d.get(k,{})
    print(k, r.get("value"), r.get("unit"), r.get("ms_per_step"), (r.get("roofline_sparse") or {}).get("frac_of_measured_copy_peak"), r.get("wall_s"), r.get("error"))
PY
