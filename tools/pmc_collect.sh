#!/bin/bash
# PMC passes (rocprofv3 --pmc only: no trace domains) for every case of tools/pmc_case.py, one counter group per run:
#   FETCH_SIZE | WRITE_SIZE | GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES | SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
# usage (repo root, on the GPU box): tools/pmc_collect.sh <outdir>;  then tools/pmc_fold.py <outdir> > profiles/rNN_pmc.json
R=$PWD; O=$1; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
for case in $(python $R/tools/pmc_case.py --list); do
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    rm -rf /tmp/pmc_run
    timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_run -- python $R/tools/pmc_case.py $case > /dev/null 2>&1
    C=$(find /tmp/pmc_run -name "*counter_collection.csv" | head -1)
    [ -n "$C" ] && grep -E "Counter_Name|hoisdf" $C > $O/${case}.pass$i.csv
  done
done
ls $O | wc -l
