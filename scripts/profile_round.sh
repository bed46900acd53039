#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/prof_$1 (run on the GPU box), then summarise with
#   python scripts/rocpd_summary.py gpurun_out/prof_$1 profiles $1
# usage: bash scripts/profile_round.sh r01
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# per-kernel time of the default bench command (pipeline depth 4) and of the unpipelined one
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py > $OUT/bench_under_rocprof.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_p1 -o bench_p1 -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu --pipeline 1 > $OUT/bench_p1_under_rocprof.log 2>&1
# per-kernel time of the other SURVEY 8 rows (config 2 VFH, 4096^2 HIMM, config 3 A* alone, config 4 RRT share)
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_rows -o rows -- python3 $ROOT/scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.err
# HBM traffic counters: one counter per pass, kernel-trace only
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --pipeline 1 > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --pipeline 1 > $OUT/pmc_write.log 2>&1
find $OUT -name "*.db" | head
