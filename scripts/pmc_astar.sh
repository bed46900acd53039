#!/bin/bash
# developer probe: hardware counters of the A* search kernel (one --pmc set per pass)
# usage: bash scripts/pmc_astar.sh <tile|persist|frontier> <bucket_width>
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
export RNA_ASTAR_KERNEL=$1 RNA_ASTAR_PIPELINE=1 REPS=1
i=0
mkdir -p $ROOT/gpurun_out/pmc_$1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $ROOT/gpurun_out/pmc_$1/p$i -o r --output-format csv -- python3 $ROOT/scripts/astar_stats.py 4096 256 $2 > $ROOT/gpurun_out/pmc_$1/log$i.txt 2>&1
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$ROOT/gpurun_out/pmc_$1/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "search" in k or "persist_kernel" in k:
        print(k)
        for c, v in sorted(d.items()):
            print("   %-28s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
PY
