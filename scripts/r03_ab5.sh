#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "astar or smoke" 2>&1 | tail -2
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search launch %.1f ms' % d['kernel_ms_per_pass_timed_region']['astar_search'], flush=True)"; }
for rep in 1 2; do
  one "prev full" RNA_LIB=librna_prev.so
  one "new full" RNA_LIB=librna.so
  one "new only" RNA_LIB=librna.so RNA_BENCH_ONLY_ASTAR=1
done
RNA_LIB=librna_stats.so RNA_BENCH_ONLY_ASTAR=1 timeout 300 python bench.py --no-cpu --steps 20 2>&1 | grep "tsa stats" | cut -c1-600
