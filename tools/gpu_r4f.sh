#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
echo "== r3 worktree, legacy kernels forced"; ( cd _bisect/99b8c15 && python ../run_legacy.py 256 2>&1 | grep "vs oracle (\|passed\|failed" | cut -c1-600 )
