#!/bin/bash
# developer probe: SQ instruction / busy counters of vfh_step_kernel at 16 384 poses on a 1024^2 map (one --pmc set per pass;
# gpurun_out/pmc_vfh/summary.txt).  usage: bash scripts/pmc_vfh_sq.sh
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_vfh
mkdir -p $OUT
export VFH_PROBE_SIZES=16384 VFH_PROBE_REPS=3
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r --output-format csv -- python3 $ROOT/scripts/vfh_probe.py > $OUT/log$i.txt 2>&1
done
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "vfh_step_kernel" if "vfh_step_kernel" in r["Kernel_Name"] else "other"
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "vfh_step" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s launches=%d avg=%.6g" % (c, len(v), sum(v) / len(v)))
PY
cat $OUT/summary.txt
