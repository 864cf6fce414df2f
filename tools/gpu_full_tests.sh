#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out
SASSD_FULL_TESTS=1 timeout 3000 python -m pytest tests -q -m gpu -s > $O/full_tests.log 2>&1; echo "full gpu tests rc=$?"
grep "max abs errors\|AssertionError\|Error\|passed\|failed" $O/full_tests.log | tail -30
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
