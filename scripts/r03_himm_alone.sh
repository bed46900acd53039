#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03/himm_alone
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/scripts/himm_alone.py 200
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o h --output-format csv -- python3 $ROOT/scripts/himm_alone.py 100 > $OUT/log.txt 2>&1
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
for r in rows[:14]:
    print("%-44s calls %5s avg %8.1f us  total %6.2f%%" % (r["Name"].replace("(anonymous namespace)::","")[:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
