cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/r06; OUT=gpurun_out/r06/sweep_depth_hwq.txt; : > $OUT
export RNA_LIB=$PWD/ros_navigation_amd/librna_d24.so
for i in 1 2; do for q in 8 24; do for d in 20 22 24; do
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --no-cpu --no-check-paths --pipeline $d 2>/dev/null | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); print('hwq $q depth $d', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'overlap %.1f' % d['roofline']['overlapped_launches'], d['config']['astar_allocated'])
except Exception as ex: print('hwq $q depth $d FAILED', ex)" >> $OUT
done; done; done; cat $OUT
