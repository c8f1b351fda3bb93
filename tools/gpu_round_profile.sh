#!/bin/bash
# Runs on the GPU box: the judged bench line + rocprofv3 kernel-trace stats + HBM traffic counters (separate passes: PMC
# runs carry --kernel-trace only), for the headline (--config 2) and a kernel trace of --config 3; everything summarised
# to text under gpurun_out/ (the rocpd databases stay on the box).  Copy the summaries you keep into profiles/.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
python $R/bench.py 2>/dev/null | tail -1 > $R/gpurun_out/bench_line.json
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined"
export MIOPEN_FIND_MODE=FAST   # keeps the naive_conv_* find-mode kernels of the warm-up out of the cfg-3 trace
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace -o bench -- $CMD > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof/fetch -o bench -- $CMD > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof/write -o bench -- $CMD > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES -d /tmp/prof/sq -o bench -- $CMD > /dev/null 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof/trace3 -o bench -- python $R/bench.py --steps 10 --warmup 5 --no-cpu-baseline --config 3 > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py /tmp/prof/trace/bench_results.db --pmc fetch=/tmp/prof/fetch/bench_results.db write=/tmp/prof/write/bench_results.db | cut -c1-260 > gpurun_out/round_profile.txt
python tools/rocpd_summary.py /tmp/prof/trace/bench_results.db --pmc sq=/tmp/prof/sq/bench_results.db 2>/dev/null | grep -A400 "PMC pass" | grep "spconv\|PMC\|kernel " | cut -c1-260 > gpurun_out/pmc_sq.txt
python tools/rocpd_summary.py /tmp/prof/trace3/bench_results.db | cut -c1-200 > gpurun_out/cfg3_kernels.txt
python tools/timeline_gaps.py /tmp/prof/trace/bench_results.db vfe_prep_kernel 4 4 | cut -c1-200 > gpurun_out/timeline_gaps.txt
python tools/timeline_gaps.py /tmp/prof/trace3/bench_results.db vfe_prep_kernel 4 4 | cut -c1-200 > gpurun_out/timeline_gaps_cfg3.txt
head -40 gpurun_out/round_profile.txt
cut -c1-300 gpurun_out/bench_line.json
