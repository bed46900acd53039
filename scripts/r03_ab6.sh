#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -x -q -k "not fuzz" 2>&1 | tail -3
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search launch %.1f ms' % d['kernel_ms_per_pass_timed_region']['astar_search'], flush=True)"; }
for rep in 1 2; do
  one "prev full" RNA_LIB=librna_prev.so
  one "new full" RNA_LIB=librna.so
  one "new vfh inline" RNA_LIB=librna.so RNA_VFH_INLINE=1
  one "new skip24" RNA_LIB=librna.so RNA_SEARCH_CU_SKIP=24
done
