#!/bin/bash
# round 4, GPU call G: training A/B of the sparse-conv defaults, batch-2 layer timings, changed training kernels' tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4g; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py tests/test_gpu_kernels.py tests/test_gpu_train.py -q -x -k "forward_rulebook or spconv_backward or weight_gradient_formulations or (training_step_vs_oracle and multi and HALF1)" > $O/pytest.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.txt
timeout 300 python tools/ablate_spconv.py --config car --batch 2 > $O/ablate_car_b2.txt 2>&1; echo "ablate b2 rc $?"
grep -v "^/opt" $O/ablate_car_b2.txt | cut -c1-260
for v in "default FOO=1" "r3geom SASSD_SPCONV_DEBUG=655360" "transposed_tables SASSD_SUBM_FWD_TABLE=0"; do
  set -- $v; name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train_$name.json 2> $O/bench_train_$name.err; echo "train $name rc $?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4g/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], d["value"], d["unit"], "ms/step", d["ms_per_step"], d.get("trials"))
PY
