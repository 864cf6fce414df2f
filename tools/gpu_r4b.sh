#!/bin/bash
# round 4, GPU call B: probe with wall clock; in-situ kernel traces of the sparse segment per geometry; stall counters at multi scale
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O
timeout 120 tools/bin/probe_mfma4x4 > $O/probe.txt 2>&1; echo "probe rc $?"
timeout 900 python -m pytest tests/test_gpu_sparse_r2.py -q -k "gather_gemm or many_blocks or reproducible" -x > $O/pytest_sparse.txt 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_sparse.txt
trace() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pf_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/pf_$name -- "$@" > $O/${name}.log 2>&1 ); echo "$name rc=$?"
  local DB=$(find /tmp/pf_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_summary.py $DB | grep -i "spconv\|rb_\|hash_build\|total GPU" | cut -c1-60,100-160 > $O/${name}_kernel_stats.txt 2>&1
}
pmc() {  # name, command...
  local name=$1; shift
  rm -rf /tmp/pm_$name; ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS -d /tmp/pm_$name -- "$@" > $O/${name}_pmc.log 2>&1 ); echo "$name pmc rc=$?"
  local DB=$(find /tmp/pm_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc.py $DB spconv > $O/${name}_pmc.json 2>&1
}
for cfg in 10 9 7 14 11; do trace car_cfg$cfg python $R/tools/run_sparse_only.py --config car --reps 20 --spconv-cfg $cfg; done
for cfg in 10 8 9 14 13 11 2; do trace multi_cfg$cfg python $R/tools/run_sparse_only.py --config multi --reps 5 --spconv-cfg $cfg; done
timeout 400 python tools/ablate_spconv.py --config multi > $O/ablate_multi.txt 2>&1; echo "ablate multi rc $?"
timeout 400 python tools/ablate_spconv.py --config car > $O/ablate_car.txt 2>&1; echo "ablate car rc $?"
pmc multi_cfg10 python $R/tools/run_sparse_only.py --config multi --reps 3 --spconv-cfg 10
pmc multi_cfg14 python $R/tools/run_sparse_only.py --config multi --reps 3 --spconv-cfg 14
pmc car_cfg10 python $R/tools/run_sparse_only.py --config car --reps 5 --spconv-cfg 10
pmc car_cfg9 python $R/tools/run_sparse_only.py --config car --reps 5 --spconv-cfg 9
pmc car_cfg7 python $R/tools/run_sparse_only.py --config car --reps 5 --spconv-cfg 7
tail -20 $O/probe.txt
for f in $O/car_cfg*_kernel_stats.txt $O/multi_cfg*_kernel_stats.txt; do echo "== $f"; cat $f; done
