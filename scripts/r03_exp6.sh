#!/bin/bash
mkdir -p gpurun_out/r03
timeout 900 python -m pytest tests -m gpu -x -q -k "not bench and not tiled" 2>&1 | tail -3
timeout 300 python bench.py --no-cpu --steps 30 > gpurun_out/r03/e6.json 2> gpurun_out/r03/e6.err; python - <<PY
import json
d=json.loads(open("gpurun_out/r03/e6.json").readline())
k=d["kernel_ms_per_pass"]
print(round(d["value"]), "ms/pass %.3f"%d["config"]["ms_per_pass"], "search %.1f"%d["kernel_ms_per_pass_timed_region"]["astar_search"], "engine(profiled turn) %.2f"%sum(v for n,v in k.items() if n not in ("astar_search","astar_reset")), {n:round(v,2) for n,v in k.items()})
PY
RNA_LIB=librna_stats.so RNA_BENCH_ONLY_ASTAR=1 timeout 300 python bench.py --no-cpu --steps 20 2>&1 | grep "tsa stats\|value" | cut -c1-400
RNA_BENCH_ONLY_ASTAR=1 timeout 300 python bench.py --no-cpu --steps 30 2>&1 | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('search only', round(d['value']), d['config']['ms_per_pass'])"
