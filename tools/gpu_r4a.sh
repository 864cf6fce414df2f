#!/bin/bash
# round 4, GPU call A: MFMA 4x4x1 probe, parity of the balanced sparse kernel, per-layer timings, two bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4a; mkdir -p $O
timeout 120 tools/bin/probe_mfma4x4 > $O/probe.txt 2>&1; echo "probe rc $?" 
timeout 600 python -m pytest tests/test_gpu_sparse_r2.py tests/test_gpu_kernels.py -q -k "spconv or mfma" -x > $O/pytest_sparse.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sparse.txt
timeout 300 python tools/ablate_spconv.py --config car --ablate > $O/ablate_car.txt 2>&1; echo "ablate car rc $?"
timeout 300 python tools/ablate_spconv.py --config multi > $O/ablate_multi.txt 2>&1; echo "ablate multi rc $?"
timeout 300 python tools/ablate_spconv.py --config waymo > $O/ablate_waymo.txt 2>&1; echo "ablate waymo rc $?"
for cfg in 0 10 6; do
  timeout 300 python bench.py --steps 200 --warmup 20 --no-train --no-cpu-baseline --spconv-cfg $cfg > $O/bench_cfg$cfg.json 2> $O/bench_cfg$cfg.err; echo "bench cfg $cfg rc $?"
done
tail -n 40 $O/probe.txt $O/ablate_car.txt
grep -h '"metric"' $O/bench_cfg*.json | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['config'].get('spconv_cfg'), d['value'], d.get('roofline_sparse'))"
