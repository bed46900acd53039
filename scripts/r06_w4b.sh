#!/bin/bash
# developer run (round 6): is W = 4 bound by the stages in flight?  512 queries per pass (NOT the metric's pass: an experiment)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-w4b}.txt; : > $OUT
run() { # lib depth queries
  env RNA_LIB=$1 timeout 400 python bench.py --no-cpu --no-check-paths --pipeline $2 --queries $3 --steps ${4:-30} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2 queries $3', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f noop %.3f reruns %.3f' % (w.get('jobs_per_search',0), w.get('noop_job_frac',0), w.get('bucket_reruns_per_search',0)), d['config']['astar_allocated'])" >> $OUT
}
run librna_w4q256.so 18 512
run librna.so 18 512
run librna_w4q256.so 10 512
run librna.so 10 512
run librna_w4full.so 18 512
cat $OUT
