R=$PWD; O=$R/gpurun_out/prof_r03b; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
    python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_under_rocprof.json 2> /dev/null
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python $R/tools/trace_stats.py $T 3 5 > $O/bench_kernel_stats.csv 2> $O/bench_timed_window.txt
rm -rf $O/trace
cat $O/bench_timed_window.txt
