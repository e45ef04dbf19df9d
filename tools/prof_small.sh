#!/bin/bash
# kernel durations of the small-M linear forms under rocprofv3 (python-loop timings of ~10 us kernels measure the launch path)
R=$PWD; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/trace_s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/trace_s -- python $R/tools/mb_small.py > /dev/null 2>&1
S=$(find /tmp/trace_s -name "*kernel_stats.csv" | head -1)
grep -E "Name|emu_small|gemm_f32" $S | cut -d, -f1-5 | cut -c1-160
