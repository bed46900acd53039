#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r03/trace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT -o bench -- python3 $ROOT/bench.py --steps 12 --no-cpu > $OUT/log.txt 2>&1
DB=$(find $OUT -name "*.db" | head -1)
python3 $ROOT/scripts/timeline2.py $DB | tee $OUT/timeline.txt
cp $DB /tmp/last.db 2>/dev/null; rm -f $DB
