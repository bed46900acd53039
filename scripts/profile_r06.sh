#!/bin/bash
# Round-6 profiles (run on the GPU box through gpurun): the default bench, the driver's command, a kernel trace of the default
# bench, the two HBM-traffic PMC passes, the SQ instruction counters of one search batch, the per-row measurements, the engine
# stream's timeline.  Every summary that bench.py quotes carries the sha256[:12] of ros_navigation_amd/csrc/astar_tile.hip it
# was taken at (`kernel_source_sha`): the bench line then says by itself whether the profile describes the kernel that ran.
# Summaries: gpurun_out/prof_r06/summaries/ -> copied to profiles/ by hand.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r06
rm -rf $OUT; mkdir -p $OUT
SHA=$(sha256sum $ROOT/ros_navigation_amd/csrc/astar_tile.hip | cut -c1-12)
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python3 $ROOT/bench.py --steps 20 --warmup 5 > $OUT/bench_driver_command.json 2> /dev/null
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o bench -- python3 $ROOT/bench.py --no-cpu --no-check-paths > $OUT/bench_under_rocprof.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-check-paths > $OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-check-paths > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace_rows -o rows -- python3 $ROOT/scripts/bench_rows.py > $OUT/rows.json 2> $OUT/rows.log
python3 $ROOT/scripts/timeline2.py $OUT/trace/*/*bench_results.db > $OUT/engine_timeline.txt 2>&1 || python3 $ROOT/scripts/timeline2.py $(find $OUT/trace -name "*.db" | head -1) > $OUT/engine_timeline.txt 2>&1
bash $ROOT/scripts/pmc_astar_sq.sh 128000 > $OUT/sq_counters.txt 2>&1
echo "   kernel_source_sha            $SHA" >> $OUT/sq_counters.txt
cd $ROOT && python3 scripts/rocpd_summary.py gpurun_out/prof_r06 gpurun_out/prof_r06/summaries r06 > $OUT/summary_stdout.txt 2>&1
python3 - <<PY
import json
p = "$OUT/summaries/r06_pmc_summary.json"
try:
    d = json.load(open(p)); d["kernel_source_sha"] = "$SHA"; json.dump(d, open(p, "w"), indent=1)
except Exception as ex:
    print("pmc summary not stamped:", ex)
PY
cp $OUT/sq_counters.txt $OUT/summaries/r06_search_sq_counters.txt
cp $OUT/engine_timeline.txt $OUT/summaries/r06_engine_timeline.txt
cp $OUT/bench_default.json $OUT/summaries/r06_bench_default.json
cp $OUT/bench_driver_command.json $OUT/summaries/r06_bench_driver_command.json
cp $OUT/rows.json $OUT/summaries/r06_rows.json
# the trace databases are large: keep the summaries only
find $OUT -name "*.db" -size +20M -delete
tail -c 700 $OUT/bench_default.json | head -c 300; echo; head -12 $OUT/summaries/r06_kernel_stats_trace.txt; tail -20 $OUT/sq_counters.txt
