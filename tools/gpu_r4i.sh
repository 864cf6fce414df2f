#!/bin/bash
# round 4, GPU call I: rulebook kernels after the interleaved hash entries / three-lookups-per-thread restructure
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4i; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_sparse_r2.py tests/test_gpu_pipeline.py -q -x -k "rulebook or pyramid or pipeline_vs_oracle or facade or multi_class or spconv_layer or input_layer" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
for c in car multi waymo; do timeout 300 python tools/ablate_spconv.py --config $c 2>&1 | grep "^rulebooks" ; done
timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline > $O/bench_car.json 2> $O/bench_car.err; echo "car rc $?"
timeout 300 python bench.py --config multi --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_multi.json 2> $O/bench_multi.err; echo "multi rc $?"
timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4i/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l); rs = d.get("roofline_sparse") or {}
            print(f.split("/")[-1], d["value"], "ms/step", d["ms_per_step"], "sparse ms", rs.get("ms"), "frac_copy", rs.get("frac_of_measured_copy_peak"))
PY
