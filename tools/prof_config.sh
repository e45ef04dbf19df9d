#!/bin/bash
# rocprofv3 kernel trace of an inference config of bench.py (run through gpurun from the repo root): tools/prof_config.sh <config> <outdir>
R=$PWD; CFG=$1; O=$R/$2; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for ts in 1 0; do
  rm -rf /tmp/trace_c
  HOISDF_TWO_STREAMS=$ts timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_c -- \
      python $R/bench.py --config $CFG --steps 6 --warmup 3 --no-cpu-baseline --no-kernel-timing > $O/bench_config${CFG}_under_rocprof_streams$ts.json 2> /dev/null
  T=$(find /tmp/trace_c -name "*kernel_trace.csv" | head -1)
  [ -n "$T" ] && python $R/tools/trace_window.py $T lattice_count_kernel 2 5 > $O/config${CFG}_kernel_stats_streams$ts.csv 2> $O/config${CFG}_window_streams$ts.txt
done
cat $O/config${CFG}_window_streams*.txt
head -40 $O/config${CFG}_kernel_stats_streams0.csv
