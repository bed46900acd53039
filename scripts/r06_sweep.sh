#!/bin/bash
# developer run (round 6): the default bench under several settings of one environment variable or bench argument, twice around
# usage: bash scripts/r06_sweep.sh out_name VAR v1 v2 ...     (VAR = an environment variable, or ARG:--bucket-width for a bench argument)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r06
NAME=$1; VAR=$2; shift 2
OUT=gpurun_out/r06/${NAME}.txt
: > $OUT
for i in 1 2; do
  for v in "$@"; do
    if [[ $VAR == ARG:* ]]; then ENVV=""; ARGS="${VAR#ARG:} $v"; else ENVV="$VAR=$v"; ARGS=""; fi
    env $ENVV timeout 300 python bench.py --no-cpu --no-check-paths $ARGS 2>/tmp/err.txt | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']
    print('$VAR=$v', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], 'search ms %.2f' % k['astar_search'], 'overlap %.1f' % d['roofline']['overlapped_launches'], 'engine ms/pass', round(sum(x for n,x in k.items() if not n.startswith('astar') and n != 'vfh_step'),3))
except Exception as ex:
    print('$VAR=$v FAILED', ex)
" >> $OUT
  done
done
cat $OUT
