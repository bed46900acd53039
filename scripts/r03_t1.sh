#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "page or retried or astar" 2>&1 | tail -5
timeout 300 python bench.py --no-cpu --steps 30 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('full', round(d['value']), 'ms/pass %.3f' % d['config']['ms_per_pass'], d['config']['astar_allocated'], flush=True)"
