#!/bin/bash
# Round-6 evidence: ONE collection at HEAD on ONE box (run through gpurun from the repo root).  Order matters: the kernel trace of the
# timed steps and the PMC records are made FIRST and copied into profiles/ ON THE BOX, so that the bench lines that follow read THIS
# run's own files (roofline.traffic / mfma_busy from r06_pmc.json, hot_path's owner split from r06_bench_kernel_stats.csv).
# Everything lands under gpurun_out/prof_r06/; copy what should be judged into profiles/ (tools/collect_profiles_r6.sh does it on the
# box, the builder repeats it locally from the merged gpurun_out).
R=$PWD; O=$R/gpurun_out/prof_r06; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for cfg in default config3 config4; do
  extra=""; marker=""
  [ $cfg = config3 ] && extra="--config 3" && marker="vote_loss_fwd_kernel vote_loss_merge_kernel"
  [ $cfg = config4 ] && extra="--config 4" && marker="vote_loss_fwd_kernel vote_loss_merge_kernel"
  HOISDF_TWO_STREAMS=0 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- \
      python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 --no-kernel-timing $extra > $O/bench_${cfg}_under_rocprof.json 2> /dev/null
  T=$(find $O/trace -name "*kernel_trace.csv" | head -1)
  S=$(find $O/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$S" ] && [ $cfg = default ] && cp $S $O/${cfg}_rocprof_stats_raw.csv
  [ -n "$T" ] && python $R/tools/trace_stats.py $T 3 5 $marker > $O/${cfg}_kernel_stats.csv 2> $O/${cfg}_timed_window.txt
  rm -rf $O/trace
done
cp $O/default_kernel_stats.csv $R/profiles/r06_bench_kernel_stats.csv
cd $R
bash tools/pmc_collect.sh $O/pmc_raw > /dev/null 2>&1
python tools/pmc_fold.py $O/pmc_raw > $O/pmc.json
cp $O/pmc.json $R/profiles/r06_pmc.json
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 --shape-report > /dev/null 2> $O/shape_report.txt
python bench.py --config 3 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 > $O/bench_config3.json 2> /dev/null
python bench.py --config 4 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 > $O/bench_config4.json 2> /dev/null
python bench.py --branch-mix --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0 > $O/bench_branch_mix.json 2> /dev/null
# the round-5 tree on the same box (ab/r5tree: git archive of 811360c + its own library), interleaved quick lines
[ -d ab/r5tree ] && bash ab/ab_r5.sh > $O/same_box_ab_vs_round5.txt 2>&1
# this round's encoder-side / inference-side changes against their switches, same box, alternating
Q="--steps 10 --warmup 5 --no-cpu-baseline --exact-f32 0 --bf16x3-leg 0"
val() { python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; }
{
  echo "# bench.py $Q (configs[1]): HOISDF_BN=torch (MIOpen BatchNorm + ATen add / relu) vs the fused BatchNorm passes, alternating"
  for i in 1 2 3; do echo "torch $(HOISDF_BN=torch python bench.py $Q 2>/dev/null | val)"; echo "hip   $(python bench.py $Q 2>/dev/null | val)"; done
  echo "# bench.py --config 3 $Q: round-1 lattice kernels + round-1 gather loop (HOISDF_LATTICE=1 HOISDF_GATHER_FWD=1) vs the chunked lattice passes + gather_fwd4"
  for i in 1 2; do echo "old   $(HOISDF_LATTICE=1 HOISDF_GATHER_FWD=1 python bench.py --config 3 $Q 2>/dev/null | val)"; echo "new   $(python bench.py --config 3 $Q 2>/dev/null | val)"; done
} > $O/same_box_ab_this_round.txt 2>&1
HOISDF_MAG_TRACE=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --bf16x3-leg 0 --exact-f32 0 --no-kernel-timing 2>&1 >/dev/null | grep "hoisdf mag" | sort | uniq -c | sort -rn > $O/library_side_magnitude_passes_3_steps.txt
ls -la $O
