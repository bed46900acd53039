#!/bin/bash
# developer run (round 6): the capacity of W = 4 (five workgroups per CU) with so many queries in flight that no stage limit binds
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-w4e}.txt; : > $OUT
run() { # lib depth queries cap only_astar
  env RNA_LIB=$1 RNA_ASTAR_PAGE_CAP=$4 RNA_BENCH_ONLY_ASTAR=$5 timeout 500 python bench.py --no-cpu --no-check-paths --pipeline $2 --queries $3 --steps 16 2>/tmp/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2 queries $3 cap $4 only_astar $5', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f reruns %.3f' % (w.get('jobs_per_search',0), w.get('bucket_reruns_per_search',0)), d['config']['astar_allocated'])" >> $OUT
  tail -1 /tmp/err.txt | grep -v amdgpu | cut -c1-200 >> $OUT
}
run librna_w4q256.so 20 1024 2048 1
run librna.so 20 1024 2048 1
run librna_w4q256.so 20 1024 2048 0
run librna.so 20 1024 2048 0
cat $OUT
