cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3p; export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r3p; R=$GRAFT_REPO_ROOT
pmc() {  # name, counter(s), command...
  local name=$1; local ctr=$2; shift; shift
  local tag=$(echo $ctr | cut -d' ' -f1)
  rm -rf /tmp/pm_$name; ( cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pm_$name -- "$@" > $O/${name}_${tag}.log 2>&1 ); echo "$name $tag rc=$?"
  local DB=$(find /tmp/pm_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocprof_pmc.py $DB > $O/${name}_${tag}.json 2>&1
}
timeout 300 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
pmc bf "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" python $R/tools/run_bf16_conv.py --only-fwd --iters 10
pmc bf "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" python $R/tools/run_bf16_conv.py --only-fwd --iters 10
timeout 200 python tools/run_bf16_conv.py --ablate > $O/bf16_ablate.json 2>$O/bf16_ablate.err; echo "timing rc=$?"; grep -E "fwd3x3_bf16_ms|hot_in|order|default_again|no_input" $O/bf16_ablate.json
grep -A12 conv2d_bf16 $O/bf_SQ_LDS_BANK_CONFLICT.json | head -30; grep -A12 conv2d_bf16 $O/bf_GRBM_GUI_ACTIVE.json | head -30; tail -3 $O/bf_GRBM_GUI_ACTIVE.log
