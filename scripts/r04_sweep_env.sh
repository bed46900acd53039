#!/bin/bash
# developer run: the default bench under several values of one environment variable, twice around
# usage: bash scripts/r04_sweep_env.sh out_name VAR v1 v2 ...   (extra bench arguments through R04_ARGS)
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/$1; VAR=$2; shift 2
: > $OUT
for i in 1 2; do
  for v in "$@"; do
    env $VAR=$v timeout 300 python bench.py --no-cpu $R04_ARGS 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$VAR=$v', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3), {n: round(v,3) for n,v in k.items()})
except Exception as ex:
    print('$VAR=$v FAILED', ex)
" >> $OUT
    tail -n 3 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
  done
done
cat $OUT
