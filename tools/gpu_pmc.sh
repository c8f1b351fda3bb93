#!/bin/bash
# PMC passes for the bench command (separate runs per counter group; kernel-trace only besides --pmc).
# Summarised on the box (the rocpd databases are too large to copy back).
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
i=0
ARGS=""
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM" ; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc/g$i -o p -- $CMD 2>&1 | grep -iE "error|invalid|not found" | head -3
  ARGS="$ARGS g$i=/tmp/pmc/g$i/p_results.db"
done
cd $R
python tools/rocpd_summary.py /tmp/pmc/g1/p_results.db --pmc $ARGS 2>&1 | grep -E "^#|spconv_f16x3|vfe_layer|kernel  " | cut -c1-200 > gpurun_out/pmc/summary.txt
wc -l gpurun_out/pmc/summary.txt
