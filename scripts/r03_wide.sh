#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for w in "" 1; do
  for d in 13 8; do
    RNA_ASTAR_PIPELINE=$d ${w:+RNA_TSA_WIDE=1} python bench.py --no-cpu --pipeline $d 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wide=${w:-0} depth=$d', round(d['value']), 'ms/step', round(d['ms_per_step'],2))"
  done
done
