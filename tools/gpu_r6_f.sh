#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r6; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6
for rep in 1 2; do
( cd _bisect/old && python tools/run_bf16_conv.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('OLD', {k: round(v,4) for k,v in d.items() if 'wgrad' in k})" )
python tools/run_bf16_conv.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('NEW', {k: round(v,4) for k,v in d.items() if 'wgrad' in k})"
done
