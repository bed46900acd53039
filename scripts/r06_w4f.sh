#!/bin/bash
# developer run (round 6): W = 4 at seven workgroups per CU with the full node pool -- possible on a 2048^2 map, whose tile bit sets are a quarter of the bench's
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/${1:-w4f}.txt; : > $OUT
run() { # lib depth grid pad
  env RNA_LIB=$1 RNA_TSA_LDS_PAD=$4 timeout 400 python bench.py --no-cpu --no-check-paths --pipeline $2 --grid $3 --steps 60 2>/tmp/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2 grid $3 pad $4', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f reruns %.3f' % (w.get('jobs_per_search',0), w.get('bucket_reruns_per_search',0)))" >> $OUT
  tail -1 /tmp/err.txt | grep -v amdgpu | cut -c1-200 >> $OUT
}
for i in 1 2; do
run librna.so 18 2048 0
run librna_w4q256.so 18 2048 0
run librna_w4q256.so 20 2048 0
run librna_w4q256.so 20 2048 1000
run librna_w4q256.so 20 2048 4000
done
cat $OUT
