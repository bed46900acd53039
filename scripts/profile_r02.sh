#!/bin/bash
# Round-2 profiles (run on the GPU box through gpurun): kernel trace of the default bench, the two HBM-traffic PMC passes
# of the same command, a kernel trace of the per-row measurements.  Summaries: python scripts/rocpd_summary.py gpurun_out/prof_r02 profiles r02
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py --steps 60 --no-cpu > $OUT/bench_under_rocprof.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 8 --warmup 4 --no-cpu > $OUT/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 8 --warmup 4 --no-cpu > $OUT/pmc_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_rows -o rows -- python3 $ROOT/scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.log
find $OUT -name "*.db" | head
tail -1 $OUT/bench_under_rocprof.log | cut -c1-300
