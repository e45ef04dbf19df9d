#!/bin/bash
# rocprofv3 kernel statistics of tools/mb_one.py for a list of "M N K" shapes: bash tools/prof_one.sh out.txt "65536 256 256" ...
R=$PWD; OUT=$R/$1; shift
export TMPDIR=/tmp
: > $OUT
cd /tmp
for shp in "$@"; do
  rm -rf /tmp/p1
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python $R/tools/mb_one.py $shp 20 ${WHAT:-fwd,dx,dw} > /dev/null 2>&1
  S=$(find /tmp/p1 -name "*kernel_stats.csv" | head -1)
  echo "== $shp (HOISDF_EMU_FORM=${HOISDF_EMU_FORM:-default})" >> $OUT
  python3 - "$S" >> $OUT <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%6s calls  avg %9.1f us  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:90]))
PY
done
cat $OUT
