#!/bin/bash
# Round-5 evidence (run through gpurun from the repo root): bench JSON lines (default line carries the exact-f32 sub-measurement),
# shape report, inference configs, branch mix, rocprofv3 kernel trace of the bench restricted to the timed steps
# (tools/trace_stats.py) + the raw --stats table, configs[4] trace.  PMC passes: tools/pmc_collect.sh + tools/pmc_fold.py.
# Everything lands under gpurun_out/prof_r05/; copy what should be judged into profiles/.
R=$PWD; O=$R/gpurun_out/prof_r05; mkdir -p $O
export TMPDIR=/tmp
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --exact-f32 0 --shape-report > /dev/null 2> $O/shape_report.txt
python bench.py --config 3 --no-cpu-baseline --exact-f32 0 > $O/bench_config3.json 2> /dev/null
python bench.py --config 4 --no-cpu-baseline --exact-f32 0 > $O/bench_config4.json 2> /dev/null
python bench.py --config 4 --f16-attention 0 --no-cpu-baseline --exact-f32 0 > $O/bench_config4_fp32_attention.json 2> /dev/null
HOISDF_ATTN16=f16 python bench.py --config 4 --no-cpu-baseline --exact-f32 0 > $O/bench_config4_round2_f16_kernel.json 2> /dev/null
python bench.py --branch-mix --no-cpu-baseline --exact-f32 0 > $O/bench_branch_mix.json 2> /dev/null
cd /tmp
for cfg in default config4; do
  extra=""; marker=""; [ $cfg = config4 ] && extra="--config 4" && marker="vote_loss_fwd_kernel"
  HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
      python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 --no-kernel-timing $extra > $O/bench_${cfg}_under_rocprof.json 2> /dev/null
  T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && cp $S $O/${cfg}_rocprof_stats_raw.csv
  [ -n "$T" ] && python $R/tools/trace_stats.py $T 3 5 $marker > $O/${cfg}_kernel_stats.csv 2> $O/${cfg}_timed_window.txt
  rm -rf $O/trace
done
cd $R
bash tools/pmc_collect.sh $O/pmc_raw > /dev/null 2>&1
python tools/pmc_fold.py $O/pmc_raw > $O/pmc.json
ls -la $O
