#!/bin/bash
# developer run (round 6): the searches alone (RNA_BENCH_ONLY_ASTAR=1: no map update, no VFH+ -- NOT the metric) against the full loop
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-only_astar}.txt; : > $OUT
run() { # lib depth only
  env RNA_LIB=$1 RNA_BENCH_ONLY_ASTAR=$3 timeout 400 python bench.py --no-cpu --no-check-paths --pipeline $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2 only_astar $3', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'host %.2f' % d['config']['host_cores_used'], 'jobs %.0f' % w.get('jobs_per_search',0))" >> $OUT
}
run librna.so 18 1
run librna.so 18 0
run librna.so 20 1
run librna_w4q256.so 20 1
run librna_w4q256.so 20 0
cat $OUT
