#!/bin/bash
# round 5, third GPU call: training parity tests (bars re-stated), weight-gradient side stream A/B, pyramid issue order A/B,
# training kernel profile
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r5; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r5; R=$GRAFT_REPO_ROOT
rm -f $O/c3_parity_dump.txt
SASSD_PARITY_DUMP=$O/c3_parity_dump.txt timeout 1500 python -m pytest tests/test_gpu_train.py -q -s -k "vs_oracle or side_stream" > $O/c3_tests_train.log 2>&1; echo "train tests rc=$?"
grep -n "passed\|failed\|^E  \|side stream" $O/c3_tests_train.log | cut -c1-600 | tail -20
for pol in front ahead front ahead; do
  SASSD_PYRAMID_ISSUE=$pol timeout 300 python bench.py --mode infer --steps 200 --warmup 20 --no-train --no-cpu-baseline > $O/c3_bench_infer_$pol.log 2>&1
  echo "infer $pol rc=$?"; python - <<PY
import json
l=[x for x in open("$O/c3_bench_infer_$pol.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("$pol", d["value"], d["ms_per_step"], d["roofline_sparse"]["ms"], d["roofline_sparse"].get("frac_of_measured_copy_peak"), d.get("fps_sequential"), d.get("frame_graph_ms"))
PY
done
for flag in "" "--serial-wgrad" "" "--serial-wgrad"; do
  timeout 300 python bench.py --mode train --steps 40 --warmup 8 $flag > $O/c3_bench_train.log 2>&1
  echo "train [$flag] rc=$?"; python - <<PY
import json
l=[x for x in open("$O/c3_bench_train.log") if x.startswith("{")]
if l:
    d=json.loads(l[-1]); print("train [$flag]", d["value"], d["ms_per_step"], d["trials"])
PY
done
rm -rf /tmp/pf_train; ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pf_train -- python $R/bench.py --mode train --steps 40 --warmup 8 > $O/c3_train_under_rocprof.log 2>&1 ); echo "train profile rc=$?"
DB=$(find /tmp/pf_train -name "*.db" | head -1); [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB > $O/c3_train_kernel_stats.txt 2>&1
head -30 $O/c3_train_kernel_stats.txt | cut -c1-170
