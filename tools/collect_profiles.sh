#!/bin/bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bench JSON lines, rocprofv3 kernel trace of the bench (timed-window stats via tools/trace_stats.py), PMC passes
#   (FETCH_SIZE / WRITE_SIZE in SEPARATE runs, MFMA busy + clock) on the GEMM and attention drivers.
# Everything lands under gpurun_out/prof_r02/; copy what should be judged into profiles/.
R=$PWD; O=$R/gpurun_out/prof_r02; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --attention split --no-cpu-baseline > $O/bench_split_attention.json 2> $O/bench_split_attention.err
python bench.py --attention split --gemm split --no-cpu-baseline --shape-report > $O/bench_split.json 2> $O/bench_split.err
python tools/mb_gsplit.py 2>&1 | grep -v amdgpu > $O/gemm_split_microbench.txt
cd /tmp
HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
    python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $O/bench_rocprof_stats_raw.csv
[ -n "$T" ] && python $R/tools/trace_stats.py $T 3 5 > $O/bench_kernel_stats.csv 2> $O/bench_timed_window.txt
rm -rf $O/trace
HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace2 -- \
    python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing --attention split --gemm split > $O/bench_split_under_rocprof.json 2> /dev/null
T=$(find $O/trace2 -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python $R/tools/trace_stats.py $T 3 5 > $O/bench_split_kernel_stats.csv 2> $O/bench_split_timed_window.txt
rm -rf $O/trace2
for drv in pmc_gemm pmc_attn mb_split; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_${drv}_$ctr -- python $R/tools/$drv.py > /dev/null 2>&1
    C=$(find $O/pmc_${drv}_$ctr -name "*counter_collection.csv" | head -1)
    [ -n "$C" ] && grep -E "Counter_Name|hoisdf" $C > $O/pmc_${drv}_$ctr.csv
    rm -rf $O/pmc_${drv}_$ctr
  done
done
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_busy -- python $R/tools/pmc_gemm.py > /dev/null 2>&1
C=$(find $O/pmc_busy -name "*counter_collection.csv" | head -1)
[ -n "$C" ] && grep -E "Counter_Name|hoisdf" $C > $O/pmc_gemm_mfma_busy.csv
rm -rf $O/pmc_busy
ls -la $O
