#!/bin/bash
# Round-5 profiles (run on the GPU box through gpurun): the default bench, a kernel trace of the same command, the two
# HBM-traffic PMC passes, the SQ instruction counters of one search batch, the issue-rate microbenchmark, the per-row
# measurements.  Summaries: python scripts/rocpd_summary.py gpurun_out/prof_r05 profiles r05
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r05
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $ROOT/bench.py --steps 20 --warmup 5 --check-paths > $OUT/bench_driver_command.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py --no-cpu > $OUT/bench_under_rocprof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_rows -o rows -- python3 $ROOT/scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.log
python3 $ROOT/scripts/timeline2.py $OUT/trace/*/*bench_results.db > $OUT/engine_timeline.txt 2>&1 || python3 $ROOT/scripts/timeline2.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/engine_timeline.txt 2>&1
bash $ROOT/scripts/pmc_astar_sq.sh 96000 > $OUT/sq_counters.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $ROOT/scripts/ubench_valu.hip -o /tmp/ubench_valu && timeout 120 /tmp/ubench_valu > $OUT/ubench_valu.txt 2>&1
# the search kernel's phase timers and job counts (a -DRNA_TSA_STATS build next to the product library) under the bench's load
if [ -f $ROOT/ros_navigation_amd/librna_stats.so ]; then
  (cd $ROOT && RNA_LIB=librna_stats.so timeout 300 python3 bench.py --no-cpu --steps 10 --warmup 1 2>&1 | grep "tsa stats" | tail -8 | cut -c1-900) > $OUT/search_job_stats.txt
fi
cd $ROOT && python3 scripts/rocpd_summary.py gpurun_out/prof_r05 gpurun_out/prof_r05/summaries r05 > $OUT/summary_stdout.txt 2>&1
# the trace databases are large: keep the summaries only
find $OUT -name "*.db" -size +20M -delete
tail -c 600 $OUT/bench_default.json | head -c 300; echo; head -12 $OUT/summaries/r05_kernel_stats_trace.txt
