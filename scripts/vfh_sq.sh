#!/bin/bash
# developer probe (round 5): SQ instruction / wait counters of vfh_step_kernel for 16 384 poses on 1024^2
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r05/vfh_sq
mkdir -p $OUT
cat > /tmp/vfh16k.py <<'PY'
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import ros_navigation_amd as R
n = 1024; L = n * 0.05
e = R.Engine(L, L, 0.05)
e.upload(R.capi.LAYER_MASTER, R.synth.occupancy_sparse(n, n, seed=1))
m = 16384
poses = R.synth.poses(m, L, L, seed=1)
e.vfh_init(m)
for _ in range(5): e.vfh_step(poses)
PY
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o r --output-format csv -- python3 /tmp/vfh16k.py > $OUT/log$i.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "vfh_step_kernel" if "vfh_step" in r["Kernel_Name"] else r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "vfh_step" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s launches=%d avg=%.6g" % (c, len(v), sum(v) / len(v)))
PY
rm -rf $OUT/p*
