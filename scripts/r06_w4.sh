#!/bin/bash
# developer run (round 6): four wavefronts per query -- how many workgroups share a CU (LDS pad), stages in flight
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-w4}.txt; : > $OUT
run() { # lib pad depth
  env RNA_LIB=$1 RNA_TSA_LDS_PAD=$2 timeout 300 python bench.py --no-cpu --no-check-paths --pipeline $3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 pad $2 depth $3', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f noop %.3f reruns %.3f' % (w.get('jobs_per_search',0), w.get('noop_job_frac',0), w.get('bucket_reruns_per_search',0)))" >> $OUT
}
run librna_w4full.so 0 18
run librna_w4full.so 2048 18
run librna_w4full.so 0 20
run librna_w4q256.so 0 18
run librna_w4q256.so 0 20
run librna.so 0 18
cat $OUT
