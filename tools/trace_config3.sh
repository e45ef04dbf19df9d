R=$PWD; O=$R/gpurun_out/c3t; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/bench.py --config 3 --steps 5 --warmup 3 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 --no-kernel-timing > $O/b.json 2>/dev/null
T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_stats.py $T 3 5 vote_loss_fwd_kernel vote_loss_merge_kernel > $R/gpurun_out/c3_kernel_stats.csv 2> $R/gpurun_out/c3_window.txt
rm -rf $O/trace; cat $R/gpurun_out/c3_window.txt
