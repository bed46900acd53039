#!/bin/bash
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search launch %.1f ms' % d['kernel_ms_per_pass_timed_region']['astar_search'], flush=True)"; }
for rep in 1 2; do
  for pf in 0 8 32 64 256; do
    one "prio_first=$pf full" RNA_TSA_PRIO_FIRST=$pf
  done
done
