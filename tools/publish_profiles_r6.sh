#!/bin/bash
# copy the merged collection (gpurun_out/prof_r06, made by tools/collect_profiles_r6.sh on the GPU box) into profiles/ under the names
# profiles/README.md indexes - run from the repo root after the gpurun call returned
O=gpurun_out/prof_r06; P=profiles
cp $O/bench_default.json $P/r06_bench_default.json.log
cp $O/bench_default_under_rocprof.json $P/r06_bench_default_under_rocprof.json.log
cp $O/bench_config3.json $P/r06_bench_config3.json.log
cp $O/bench_config4.json $P/r06_bench_config4.json.log
cp $O/bench_branch_mix.json $P/r06_bench_branch_mix.json.log
cp $O/default_kernel_stats.csv $P/r06_bench_kernel_stats.csv
cp $O/default_rocprof_stats_raw.csv $P/r06_bench_rocprof_stats_raw.csv
cp $O/default_timed_window.txt $P/r06_bench_timed_window.txt
cp $O/shape_report.txt $P/r06_bench_shape_report.txt
cp $O/config3_kernel_stats.csv $P/r06_config3_kernel_stats.csv
cp $O/config3_timed_window.txt $P/r06_config3_timed_window.txt
cp $O/config4_kernel_stats.csv $P/r06_config4_kernel_stats.csv
cp $O/config4_timed_window.txt $P/r06_config4_timed_window.txt
cp $O/library_side_magnitude_passes_3_steps.txt $P/r06_library_side_magnitude_passes_3_steps.txt
cp $O/pmc.json $P/r06_pmc.json
rm -rf $P/r06_pmc_raw; cp -r $O/pmc_raw $P/r06_pmc_raw
cp $O/same_box_ab_this_round.txt $P/r06_same_box_ab_this_round.txt
ls -la $P/r06_* | wc -l
