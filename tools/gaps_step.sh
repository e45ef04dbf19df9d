#!/bin/bash
# device idle time inside the timed steps (two streams, as the bench runs): rocprofv3 kernel trace -> tools/trace_gaps.py
R=$PWD; TAG=${1:-gaps}; shift; O=$R/gpurun_out/trace_$TAG; mkdir -p $O
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- \
    python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 --no-kernel-timing "$@" > $O/bench.json 2> /dev/null
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_gaps.py $T adamw_chunks_kernel 1 4 > $R/gpurun_out/${TAG}_gaps.txt 2>&1
tail -1 $O/bench.json | cut -c1-200 >> $R/gpurun_out/${TAG}_gaps.txt
rm -rf $O
head -30 $R/gpurun_out/${TAG}_gaps.txt
