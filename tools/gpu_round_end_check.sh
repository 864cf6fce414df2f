#!/bin/bash
# the driver's round-end sequence on one box: the default -m gpu suite (timed), then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/round_end; mkdir -p $O
start=$(date +%s)
timeout 1700 python -m pytest ${TESTS:-tests/} -x -q -m gpu --durations=12 > $O/pytest_gpu.txt 2>&1; echo "pytest rc $? in $(( $(date +%s) - start )) s"
tail -22 $O/pytest_gpu.txt | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; echo "smoke rc $?"; tail -3 $O/smoke.txt
