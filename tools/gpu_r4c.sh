#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r4c; mkdir -p $O
timeout 400 python tools/ablate_spconv.py --config multi --ablate > $O/ablate_multi.txt 2>&1; echo "ablate multi rc $?"
timeout 400 python tools/ablate_spconv.py --config car --ablate > $O/ablate_car.txt 2>&1; echo "ablate car rc $?"
grep -v "^/opt" $O/ablate_multi.txt | grep -A5 "64->64\|^sum" | cut -c1-200
grep -v "^/opt" $O/ablate_car.txt | grep -A5 "64->64\|^sum"| cut -c1-200
