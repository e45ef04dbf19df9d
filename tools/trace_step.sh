#!/bin/bash
# rocprofv3 kernel trace of the bench restricted to the timed steps -> gpurun_out/<tag>_kernel_stats.csv (tools/trace_stats.py)
R=$PWD; TAG=${1:-step}; shift; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
    python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --exact-f32 0 --no-kernel-timing "$@" > $O/bench_under_rocprof.json 2> /dev/null
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$S" ] && cp $S $R/gpurun_out/${TAG}_rocprof_stats_raw.csv
[ -n "$T" ] && python $R/tools/trace_stats.py $T 3 5 > $R/gpurun_out/${TAG}_kernel_stats.csv 2> $R/gpurun_out/${TAG}_timed_window.txt
cp $O/bench_under_rocprof.json $R/gpurun_out/${TAG}_bench_under_rocprof.json
rm -rf $O
head -45 $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160; cat $R/gpurun_out/${TAG}_timed_window.txt
