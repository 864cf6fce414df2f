#!/bin/bash
# round 4, GPU call E: new parity / DDP tests, bench with the one-rank RCCL exchange, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4e; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_multi.py tests/test_gpu_train.py tests/test_gpu_train_fused.py tests/test_gpu_kernels.py tests/test_gpu_wino4.py -q -x -k "one_rank or k21_vs_oracle or fused_head or nms or wino4 or spconv_backward or spconv_layer" > $O/pytest_new.txt 2>&1; echo "pytest rc $?"; tail -8 $O/pytest_new.txt; grep "vs oracle" $O/pytest_new.txt
timeout 600 python bench.py --mode train --steps 40 --warmup 8 --force-ddp > $O/bench_train_forceddp.json 2> $O/bench_train_forceddp.err; echo "train force-ddp rc $?"; tail -2 $O/bench_train_forceddp.err
timeout 600 python bench.py --mode train --steps 40 --warmup 8 > $O/bench_train.json 2> $O/bench_train.err; echo "train rc $?"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train > $O/bench_20steps.json 2> $O/bench_20steps.err; echo "bench 20 rc $?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r4e/bench_*.json")):
    for l in open(f):
        if l.startswith("{"):
            d = json.loads(l)
            print(f.split("/")[-1], d["value"], d["unit"], "ms/step", d["ms_per_step"], "trials", d.get("trials"), "allreduce_ms", d.get("allreduce_ms"), d["config"].get("parallelism"))
PY
