#!/bin/bash
# the driver's round-end GPU tier: default -m gpu suite (timed) + smoke
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4h; mkdir -p $O
start=$(date +%s)
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - start )) s"
tail -40 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
