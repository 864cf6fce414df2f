#!/bin/bash
# the COMPLETE GPU test set (default + the slow-marked case) + smoke, with the printed parity lines kept:
#   gpurun_out/prof/full_tests.log, full_tests_tail.txt (-> profiles/rNN_full_tests_tail.txt by tools/collect_profiles.py)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/prof; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/prof
SASSD_FULL_TESTS=1 timeout 1800 python -m pytest tests -q -m gpu -s > $O/full_tests.log 2>&1; echo "full gpu tests rc=$?"
( echo "# SASSD_FULL_TESTS=1 python -m pytest tests -q -m gpu -s on csrc $(python -c 'import sassd; from sassd import _C; print(_C.csrc_hash())'), commit $(git rev-parse --short HEAD 2>/dev/null || echo "${GRAFT_COMMIT:-unknown}")";
  grep -a "passed\|failed\| error\|vs float64\|vs the .* arbiter\|bf16 step, every\|max abs errors\|waymo-scale training step" $O/full_tests.log | cut -c1-2500 ) > $O/full_tests_tail.txt
tail -2 $O/full_tests_tail.txt | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
