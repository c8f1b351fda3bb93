#!/bin/bash
# round 5, GPU call 3: (a) one-column-block variants of the 256-column layers, A/B on the headline workload, interleaved;
# (b) TA / TCP counters of the production line; (c) the bit-identity test of the variants
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5c3
mkdir -p $OUT
cd $R
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
( timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "one_column_block" 2>&1 | tail -3 ) > $OUT/pytest_onecol.txt
for rep in 1 2; do
  for d in 0 262144 524288; do
    echo "diag $d rep $rep: $(timeout 300 $B --conv-diag $d 2>/dev/null | tail -1 | python tools/r5/line_brief.py)"
  done
done > $OUT/ab_onecol.txt 2>&1
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
for grp in "TA_BUSY_avr TA_UTIL" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"; do
  n=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/prof/$n -o b -- $CMD > /tmp/prof_$n.log 2>&1
  echo "== $grp (rc $?)"
  python $R/tools/rocpd_summary.py /tmp/prof/$n/b_results.db --pmc x=/tmp/prof/$n/b_results.db 2>/dev/null | grep -A200 "PMC pass" | grep "spconv_f16x3_kernel<256\|spconv_f16x3_kernel<128, 8, 2, 8\|spconv_dma_kernel<64, 4\|counter" | cut -c1-200
done > $OUT/pmc_ta.txt 2>&1
cat $OUT/pytest_onecol.txt $OUT/ab_onecol.txt; head -60 $OUT/pmc_ta.txt
