R=$PWD; O=$R/gpurun_out/prof_attn; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python $R/tools/prof_attn_bwd.py > /dev/null 2>&1
S=$(find $O/trace -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $S | cut -c1-150 | head -12; rm -rf $O/trace
