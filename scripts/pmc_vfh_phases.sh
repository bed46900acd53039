#!/bin/bash
# developer probe: the instruction counters and the time of vfh_step_kernel at 16 384 poses cut short at each of its phase marks
# (RNA_VFH_EXIT) (a -DRNA_VFH_SKIPS build: `make -C ros_navigation_amd/csrc variant NAME=vskip EXTRA=-DRNA_VFH_SKIPS`).
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_vfh_phases
rm -rf $OUT; mkdir -p $OUT
export RNA_LIB=$ROOT/ros_navigation_amd/librna_vskip.so
for sk in 1 2 3 4 5 6 7 8 9 0; do
  export RNA_VFH_EXIT=$sk
  VFH_PROBE_SIZES=1,1024,16384 python3 $ROOT/scripts/vfh_probe.py 2>/dev/null | tr '\n' ' ' > $OUT/time_$sk.txt
  VFH_PROBE_SIZES=16384 VFH_PROBE_REPS=2 timeout 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/p$sk -o r --output-format csv -- python3 $ROOT/scripts/vfh_probe.py > $OUT/log$sk.txt 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
print("every thread leaves at mark n: 1 state loads + ranges preset, 2 + submap geometry, 3 + window cells, 4 + obstacle ranges, 5 + cell magnitudes, 6 + sector sums, 7 + binary/masked histograms, 8 + Select_Direction, 9 + Cant_Turn_To_Goal, 0 whole kernel; per wavefront (32 768 per launch)")
for sk in (1, 2, 3, 4, 5, 6, 7, 8, 9, 0):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/p%d/*counter_collection.csv" % sk):
        for r in csv.DictReader(open(f)):
            if "vfh_step_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    t = open("$OUT/time_%d.txt" % sk).read().strip()
    print("exit %2d  " % sk + "  ".join("%s %7.1f" % (c[9:], sum(v) / len(v) / 32768) for c, v in sorted(acc.items())) + "   | " + t)
PY
cat $OUT/summary.txt
