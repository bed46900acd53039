cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/sweep_depth.txt; : > $OUT
for i in 1 2 3; do for d in ${DEPTHS:-18 20}; do
  for mode in "default:" "driver:--steps 20 --warmup 5"; do
    timeout 300 python bench.py --no-cpu --no-check-paths --pipeline $d ${mode#*:} 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('depth $d ${mode%%:*}', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'overlap %.1f' % d['roofline']['overlapped_launches'], d['config']['astar_allocated'])" >> $OUT
  done; done; done; cat $OUT
