#!/bin/bash
# developer run: pipeline depth 13 against 16 with the default bench and with the driver's shorter command
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2; do for p in 13 16; do
  python bench.py --no-cpu --pipeline $p 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); k=d['kernel_ms_per_pass']; print('depth $p default:', round(d['value']), 'ms/pass %.3f'%d['config']['ms_per_pass'], 'search ms %.2f'%k['astar_search'], 'engine', round(sum(v for n,v in k.items() if not n.startswith('astar') and n != 'vfh_step'),3))"
  python bench.py --no-cpu --steps 20 --warmup 5 --pipeline $p 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('depth $p steps 20 :', round(d['value']), 'ms/pass %.3f'%d['config']['ms_per_pass'])"
done; done
