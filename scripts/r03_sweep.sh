#!/bin/bash
one() { tag=$1; shift; env "$@" timeout 300 python bench.py --no-cpu --steps 30 ${EXTRA} 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$tag', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search launch %.1f ms' % d['kernel_ms_per_pass_timed_region']['astar_search'], flush=True)"; }
for bw in 96000 144000 192000 400000; do EXTRA="--bucket-width $bw" one "bucket $bw" X=1; done
EXTRA="" one "skip16" RNA_SEARCH_CU_SKIP=16
EXTRA="" one "skip24" RNA_SEARCH_CU_SKIP=24
EXTRA="" one "skip32" RNA_SEARCH_CU_SKIP=32
EXTRA="" one "skip40" RNA_SEARCH_CU_SKIP=40
EXTRA="--pipeline 14" one "depth 14" X=1
EXTRA="--pipeline 12" one "depth 12" X=1
