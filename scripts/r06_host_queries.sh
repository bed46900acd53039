cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/host_queries.txt; : > $OUT
run() { name=$1; shift
  env "$@" RNA_HOST_TRACE=1 timeout 300 python bench.py --no-cpu --no-check-paths $ARGS 2>/tmp/err.txt | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$name', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'host %.2f' % d['config']['host_cores_used'], 'overlap %.1f' % d['roofline']['overlapped_launches'])" >> $OUT
  grep "host trace" /tmp/err.txt | tail -1 | cut -c1-220 >> $OUT
}
for i in 1 2; do
ARGS="" run per_look_3 A=1
ARGS="" run per_look_18 RNA_ASTAR_QUERIES_PER_LOOK=18
ARGS="" run per_look_1 RNA_ASTAR_QUERIES_PER_LOOK=1
ARGS="--pipeline 20" run per_look_3_depth20 A=1
done
cat $OUT
