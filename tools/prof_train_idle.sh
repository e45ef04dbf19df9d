#!/bin/bash
# device idle time inside the training step (host-bound stretches): kernel trace of the default bench (two streams), window = the
# last steps delimited by the optimizer kernel (adamw_chunks_kernel, once per step); tools/prof_train_idle.sh <outdir>
R=$PWD; O=$R/$1; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/trace_t
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_t -- \
    python $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --exact-f32 0 --no-kernel-timing > $O/bench_train_under_rocprof.json 2> /dev/null
T=$(find /tmp/trace_t -name "*kernel_trace.csv" | head -1)
[ -n "$T" ] && python $R/tools/trace_window.py $T adamw_chunks_kernel 1 5 > $O/train_kernel_stats_two_streams.csv 2> $O/train_window_two_streams.txt
cat $O/train_window_two_streams.txt
[ -n "$T" ] && python $R/tools/trace_gaps.py $T adamw_chunks_kernel 1 5 > $O/train_gaps.txt 2>&1
head -60 $O/train_gaps.txt
