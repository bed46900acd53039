#!/bin/bash
# A/B on one box: RNA_LIB variants, search-only and full loop, alternating
mkdir -p gpurun_out/r03
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search launch %.1f ms' % d['kernel_ms_per_pass_timed_region']['astar_search'], flush=True)"; }
for rep in 1 2; do
  for lib in ${LIBS:-librna_prev.so librna.so}; do
    one "$lib only" RNA_LIB=$lib RNA_BENCH_ONLY_ASTAR=1
    one "$lib full" RNA_LIB=$lib
  done
done
