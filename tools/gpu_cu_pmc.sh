cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ISF_BENCH_FRAME_CACHE=/tmp/isf_bench_frames
mkdir -p $R/gpurun_out/r04_cu_pmc
for V in 512 6656; do
  CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-cfg3 --no-cfg4 --no-cfg5 --no-pipelined --conv-diag $V"
  timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof/cu_t$V -o bench -- $CMD > /dev/null 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d /tmp/prof/cu_p$V -o bench -- $CMD > /dev/null 2>&1
  cd $R
  python tools/rocpd_summary.py /tmp/prof/cu_t$V/bench_results.db --pmc sq=/tmp/prof/cu_p$V/bench_results.db 2>/dev/null | grep "spconv_cu\|spconv_f16x3_kernel<256\|^kernel\|PMC pass" | cut -c1-200 > gpurun_out/r04_cu_pmc/diag$V.txt
  cd /tmp
done
head -30 $R/gpurun_out/r04_cu_pmc/diag512.txt
