#!/bin/bash
# developer run: the default bench for several builds of the library in turn, twice around, on one box
# usage: bash scripts/r04_ab.sh out_name lib1.so lib2.so ...      (extra env through R04_ENV="A=1 B=2")
cd ${GRAFT_REPO_ROOT:-/root/repo}
OUT=gpurun_out/$1; shift
: > $OUT
for i in 1 2; do
  for l in "$@"; do
    env $R04_ENV RNA_LIB=$l timeout 300 python bench.py --no-cpu 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$l', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3), 'alloc', d['config']['astar_allocated'])
except Exception as ex:
    print('$l FAILED', ex)
" >> $OUT
    tail -n 3 /tmp/err.txt | grep -v amdgpu.ids >> $OUT
  done
done
cat $OUT
