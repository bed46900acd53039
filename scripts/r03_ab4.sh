#!/bin/bash
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search launch %.1f ms' % d['kernel_ms_per_pass_timed_region']['astar_search'], flush=True)"; }
for rep in 1 2; do
  one "new full" X=1
  one "new only" RNA_BENCH_ONLY_ASTAR=1
  one "new d12" RNA_ASTAR_PIPELINE=12
  one "new d14" RNA_ASTAR_PIPELINE=14
done
timeout 300 python bench.py --no-cpu --steps 20 --warmup 5 | cut -c1-100
