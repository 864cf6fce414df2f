#!/bin/bash
# branch bf16-sparse-trunk: 64-channel sparse convs on the bf16 MFMA in the bf16 training mode -- kernel parity, training step A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/bf16sp; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_sparse_r2.py -q -x -s -k "bf16_mfma_tiles" > $O/pytest.txt 2>&1; echo "pytest rc $?"; grep "bf16 MFMA tiles\|passed\|failed\|Error\|assert" $O/pytest.txt | cut -c1-200 | tail -20
if grep -q " passed" $O/pytest.txt && ! grep -q failed $O/pytest.txt; then
  timeout 300 python bench.py --mode train --steps 40 --warmup 8 > $O/train_fp32sp.json 2> $O/train_fp32sp.err; echo "train (fp32 sparse) rc $?"
  timeout 300 python bench.py --mode train --steps 40 --warmup 8 --sparse-precision bf16 > $O/train_bf16sp.json 2> $O/train_bf16sp.err; echo "train (bf16 sparse) rc $?"
  python - <<'PY'
import json
for f in ("train_fp32sp", "train_bf16sp"):
    for l in open("gpurun_out/bf16sp/%s.json" % f):
        if l.startswith("{"):
            d = json.loads(l); print(f, d["value"], "samples/s", d["ms_per_step"], "ms", d["final_loss"], d["loss_terms"])
PY
fi
