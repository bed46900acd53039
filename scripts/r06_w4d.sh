#!/bin/bash
# developer run (round 6): W = 4 with more queries in flight than the metric's passes allow (512 per launch, half the pages per query)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-w4d}.txt; : > $OUT
run() { # lib depth queries cap
  env RNA_LIB=$1 RNA_ASTAR_PAGE_CAP=$4 timeout 400 python bench.py --no-cpu --no-check-paths --pipeline $2 --queries $3 --steps 30 2>/tmp/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2 queries $3 cap $4', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f reruns %.3f' % (w.get('jobs_per_search',0), w.get('bucket_reruns_per_search',0)), d['config']['astar_allocated'])" >> $OUT
  tail -1 /tmp/err.txt | grep -v amdgpu | cut -c1-200 >> $OUT
}
run librna_w4q256.so 20 512 4096
run librna.so 20 512 4096
run librna_w4q256.so 16 512 4096
run librna.so 16 512 4096
cat $OUT
