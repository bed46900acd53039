#!/bin/bash
# developer run (round 6): optionally the A* parity tests of the product library, then the default bench for several builds of
# the library in turn, twice around (scripts/r04_ab.sh), each line with the job counts the run itself observed
# usage: bash scripts/r06_ab.sh out_name lib1.so lib2.so ...      (R06_TESTS=1 runs the A* parity tests first)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06
NAME=$1; shift
if [ "${R06_TESTS:-0}" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "astar" > gpurun_out/r06/${NAME}_tests.txt 2>&1
  tail -n 4 gpurun_out/r06/${NAME}_tests.txt
fi
OUT=gpurun_out/r06/${NAME}.txt
: > $OUT
for i in $(seq 1 ${R06_ROUNDS:-2}); do
  for l in "$@"; do
    env $R06_ENV RNA_LIB=$l timeout 300 python bench.py --no-cpu --no-check-paths ${R06_ARGS} 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']; w=d['roofline'].get('work_inflation') or {}
    print('$l', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3), 'host cores %.2f' % d['config']['host_cores_used'],
          '| jobs/search %.0f jobs/tile %.2f noop %.3f sticky %.3f rows/job %.1f tiles %.0f reruns %.3f idle %s' % tuple(w.get(x, 0) for x in ('jobs_per_search','jobs_per_touched_tile','noop_job_frac','sticky_turn_frac','rows_written_per_job','tiles_touched_per_search','bucket_reruns_per_search','idle_frac_developer_build')))
except Exception as ex:
    print('$l FAILED', ex)
" >> $OUT
    tail -n 3 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
  done
done
cat $OUT
