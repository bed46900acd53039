#!/bin/bash
# developer probe: SQ instruction counters of rrt_kernel, per launch of scripts/rrt_stats.py with RRT_ONLY_ABORTED=1 (launches 1-2: config
# 4's 512 queries; 3: the 13 queries that exhaust the sample budget, alone; 4: those with a one-node tree; 5: the others)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_rrt
rm -rf $OUT; mkdir -p $OUT
export RRT_ONLY_ABORTED=1
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r --output-format csv -- python3 $ROOT/scripts/rrt_stats.py 2048 512 > $OUT/log$i.txt 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "rrt_kernel" in r["Kernel_Name"]:
            n[r["Counter_Name"]] += 1
            acc[n[r["Counter_Name"]]][r["Counter_Name"]] = float(r["Counter_Value"])
for launch in sorted(acc):
    print("launch %d  " % launch + "  ".join("%s %.4g" % (c[3:], v) for c, v in sorted(acc[launch].items())))
PY
cat $OUT/summary.txt; grep "aborted alone\|queries, trees" $OUT/log1.txt
