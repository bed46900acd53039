#!/bin/bash
# developer run: the map-update chain by itself under rocprofv3, on the whole chip and restricted to the 32 CUs the
# search streams leave free (RNA_ENGINE_CU_MASK=32) -- what the chain costs in CU time, not in waiting
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for mask in 0 32; do
  OUT=$ROOT/gpurun_out/r04/himm_alone_$mask
  rm -rf $OUT; mkdir -p $OUT
  export RNA_ENGINE_CU_MASK=$mask
  [ -n "$RNA_LIB" ] && export RNA_LIB
  python3 $ROOT/scripts/himm_alone.py 200
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o h --output-format csv -- python3 $ROOT/scripts/himm_alone.py 100 > $OUT/log.txt 2>&1
  F=$(find $OUT -name "*kernel_stats.csv" | head -1)
  echo "== RNA_ENGINE_CU_MASK=$mask ${RNA_LIB}"
  python3 - <<PY
import csv
rows=list(csv.DictReader(open("$F")))
for r in rows[:14]:
    print("%-44s calls %5s avg %8.1f us  total %6.2f%%" % (r["Name"].replace("(anonymous namespace)::","")[:44], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
done
