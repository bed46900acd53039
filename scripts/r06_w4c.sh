#!/bin/bash
# developer run (round 6): capacity of W = 4 with 6 / 7 workgroups per CU, with enough queries in flight (512 per launch: an experiment)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-w4c}.txt; : > $OUT
run() { # lib depth queries
  env RNA_LIB=$1 timeout 400 python bench.py --no-cpu --no-check-paths --pipeline $2 --queries $3 --steps ${4:-30} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2 queries $3', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f noop %.3f reruns %.3f' % (w.get('jobs_per_search',0), w.get('noop_job_frac',0), w.get('bucket_reruns_per_search',0)), d['config']['astar_allocated'])" >> $OUT
}
run librna_w4n2304.so 12 512
run librna_w4n1632.so 12 512
run librna_w4n2304.so 20 256
run librna_w4n1632.so 20 256
run librna_w4q256.so 12 512
run librna.so 12 512
cat $OUT
