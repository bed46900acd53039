#!/bin/bash
# developer probe: SQ instruction / busy counters of the A* search kernel for one 256-query batch on 4096^2 (one --pmc set
# per pass; gpurun_out/pmc_sq/summary.txt).  usage: bash scripts/pmc_astar_sq.sh [bucket_width]
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
BW=${1:-24000}
export REPS=1
OUT=$ROOT/gpurun_out/pmc_sq
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r --output-format csv -- python3 $ROOT/scripts/astar_stats.py 4096 256 $BW > $OUT/log$i.txt 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "search" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s launches=%d avg=%.6g" % (c, len(v), sum(v) / len(v)))
PY
# the settled cells of the batch the counters belong to (scripts/astar_stats.py prints mean and maximum per query)
python3 - <<PY >> $OUT/summary.txt
import re
m = re.search(r"settled mean/max=(\d+)/", open("$OUT/log1.txt").read())
if m: print("   %-28s %d  (256 queries x the mean the same run reports)" % ("SETTLED_CELLS_OF_THE_BATCH", int(m.group(1)) * 256))
PY
cat $OUT/summary.txt
