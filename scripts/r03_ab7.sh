#!/bin/bash
timeout 1500 python -m pytest tests -m gpu -x -q -k "himm or tiled or loop or update or compose or smoke" 2>&1 | tail -3
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_pass']; print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'engine(bracketed) %.2f' % sum(v for n,v in k.items() if n not in ('astar_search','astar_reset','astar_init','vfh_step')), {n:round(v,2) for n,v in k.items() if n.startswith('himm') or n.startswith('comp')}, flush=True)"; }
for rep in 1 2; do
  one "prev full" RNA_LIB=librna_prev.so
  one "new full" RNA_LIB=librna.so
done
timeout 120 python scripts/fuzz_himm_vfh.py 60 91 2>&1 | tail -1
