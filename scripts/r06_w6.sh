cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/w6_five_per_cu.txt; : > $OUT
run() { env RNA_LIB=$1 timeout 400 python bench.py --no-cpu --no-check-paths --pipeline $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); w=d['roofline'].get('work_inflation') or {}
print('$1 depth $2', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'launch ms %.1f' % d['roofline']['avg_launch_ms'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'jobs %.0f noop %.3f reruns %.3f' % (w.get('jobs_per_search',0), w.get('noop_job_frac',0), w.get('bucket_reruns_per_search',0)))" >> $OUT; }
for i in 1 2; do
run librna.so 18
run librna_w6q256.so 18
run librna_w6q256.so 20
run librna_w6q512.so 20
done
cat $OUT
