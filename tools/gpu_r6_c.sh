#!/bin/bash
# round 6, call C: training step A/B against the round-5 tree (_bisect/old), kernel stats of the new tree, trajectory record
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
for rep in 1 2; do
  ( cd _bisect/old && timeout 300 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('old train', d['value'], d['ms_per_step'], d['trials'])" )
  timeout 300 python bench.py --mode train --steps 40 --warmup 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new train', d['value'], d['ms_per_step'], d['trials'])"
done
rm -rf /tmp/pf_train; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_train -- python $GRAFT_REPO_ROOT/bench.py --mode train --steps 20 --warmup 8 > $O/c_train_prof.json 2> $O/c_train_prof.err ); echo "prof rc=$?"
DB=$(find /tmp/pf_train -name "*.db" | head -1)
[ -n "$DB" ] && python tools/rocprof_summary.py $DB > $O/c_train_kernel_stats.txt 2>&1
head -60 $O/c_train_kernel_stats.txt
timeout 600 python tests/analysis/train_trajectory.py --steps 200 --out $O/train_trajectory.json 2>&1 | tail -14
