#!/bin/bash
# developer run: the default bench over a grid of environment settings ("A=1 B=2" per line of arguments), once each
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/$1; shift
: > $OUT
for cfg in "$@"; do
  env $cfg timeout 300 python bench.py --no-cpu $R04_ARGS 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$cfg', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3), {n: round(v,3) for n,v in k.items() if not n.startswith('astar')})
except Exception as ex:
    print('$cfg FAILED', ex)
" >> $OUT
  tail -n 3 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
done
cat $OUT
