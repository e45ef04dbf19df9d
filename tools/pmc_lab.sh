#!/bin/bash
# PMC passes over one launch family of a ubench lab binary: pmc_lab.sh <out-prefix> <binary> <args...>
# (separate rocprofv3 runs per counter group; --pmc only, no trace domains)
OUT=$1; shift
export TMPDIR=/tmp
cd /tmp
for grp in "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rm -rf /tmp/pmc_$tag
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag -- "$@" > /dev/null 2>&1
  C=$(find /tmp/pmc_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$C" ] && python3 - "$C" >> $OUT <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].split("(")[0][-60:]
    if "prep_weight" in n: continue
    acc[(n, r["Counter_Name"])].append(float(r["Counter_Value"]))
for (n, c), v in sorted(acc.items()):
    print(f"{n:60s} {c:28s} {sum(v)/len(v):16.0f}  (x{len(v)})")
PY
done
cat $OUT
