#!/bin/bash
# developer probe (round 5): one 256-query batch alone on the bench map for several builds of the library: launch times
# (scripts/astar_stats.py) and the SQ instruction / wait counters of the search kernel.
# usage: bash scripts/r05_sq.sh out_name lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
OUT=$ROOT/gpurun_out/r05/$NAME
mkdir -p $OUT
: > $OUT.txt
for lib in "$@"; do
  echo "== $lib" >> $OUT.txt
  RNA_LIB=$lib REPS=5 timeout 200 python3 $ROOT/scripts/astar_stats.py 4096 256 96000 2>&1 | grep "search ms\|settled mean" | cut -c1-260 >> $OUT.txt
  i=0
  for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    i=$((i+1))
    RNA_LIB=$lib REPS=1 timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/${lib}_p$i -o r --output-format csv -- python3 $ROOT/scripts/astar_stats.py 4096 256 96000 > $OUT/${lib}_log$i.txt 2>&1
  done
  python3 - <<PY >> $OUT.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/${lib}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "search" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s launches=%d avg=%.6g" % (c, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/${lib}_p*
done
cat $OUT.txt
