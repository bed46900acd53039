#!/bin/bash
# developer run: the default bench alternating between librna.so and another build (RNA_LIB), on one box
# usage: bash scripts/ab_lib.sh librna_other.so [rounds]
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${2:-3}); do
  for l in librna.so $1; do
    RNA_LIB=$l python bench.py --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']; print('$l', round(d['value']), 'engine ms/pass', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3))"
  done
done
